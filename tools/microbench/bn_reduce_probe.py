#!/usr/bin/env python3
"""BatchNorm backward reduce vs apply on the shapes where the reduce costs 3x the apply in the step: how much of the reduce is its
atomic tail?  Runs each op back to back; under rocprofv3 --kernel-trace the deterministic pass shows col_reduce_kernel WITHOUT atomics
(per-workgroup partial stores) beside the default one.  usage: python tools/microbench/bn_reduce_probe.py [det]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cavp_amd import _lib, ops, train_ops as T  # noqa: E402

dev = torch.device("cuda:0")
det = len(sys.argv) > 1 and sys.argv[1] == "det"
if det:
    _lib.set_deterministic(True, dev)
SHAPES = [(32 * 28 * 28, 128), (32 * 56 * 56, 64), (32 * 14 * 14, 1024), (32 * 14 * 14, 256), (32 * 56 * 56, 256), (32 * 28 * 28, 512)]
for rows, c in SHAPES:
    dy = torch.randn((rows, c), device=dev).to(torch.bfloat16)
    z = torch.randn((rows, c), device=dev).to(torch.bfloat16)
    dz = torch.empty_like(z)
    mean, rstd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    sc, sh, gamma = torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.ones(c, device=dev)
    s0, s1 = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    def red():
        T.bn_act_bwd_reduce(dy, None, z, mean, rstd, ops.ACT_RELU, s0, s1, fwd_scale=sc, fwd_shift=sh)
    def app():
        T.bn_act_bwd_apply(dy, None, z, mean, rstd, gamma, s0, s1, ops.ACT_RELU, dz, fwd_scale=sc, fwd_shift=sh)
    res = []
    for f in (red, app):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"rows {rows:7d} C {c:5d}  ({rows * c * 2 / 1e6:6.1f} MB per tensor)  reduce {res[0]:6.1f} us  apply {res[1]:6.1f} us   {'deterministic (partials + finish launch)' if det else 'default (atomics)'}", flush=True)
