#!/usr/bin/env python3
"""The head / token / projector weight-gradient group of one C1' training step (B = 32: 2B x 56 x 56 = 200704 pixel rows;
row 191 of profiles/r04_layers_train_bf16.txt without the audio encoder's jobs) as ONE cavp_conv2d_wgrad_group call, per
setting of cavp_set_wgrad_big.  GPU box only.  usage: python tools/bench_wgrad_group.py [--reps 10] [--modes 0:1,0:0,1:1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cavp_amd import _lib, train_ops as T  # noqa: E402

# (N, H, W, Cin, Cout, k, pad)
JOBS = [
    ("cls_1x1_256_8", 64, 56, 56, 256, 8, 1, 0),
    ("head1_3x3_256_256", 64, 56, 56, 256, 256, 3, 1),
    ("head0_3x3_304_256", 64, 56, 56, 304, 256, 3, 1),
    ("ca_fc2_1216_304", 64, 1, 3136, 1216, 304, 1, 0),
    ("ca_fc1_304_1216", 64, 1, 3136, 304, 1216, 1, 0),
    ("ca_proj_304_304", 64, 1, 3136, 304, 304, 1, 0),
    ("tok_a_304_304", 64, 1, 3136, 304, 304, 1, 0),
    ("tok_b_304_304", 32, 1, 3136, 304, 304, 1, 0),
    ("tok_c_304_304", 32, 1, 3136, 304, 304, 1, 0),
    ("proj_fc2_256_304", 32, 1, 3136, 256, 304, 1, 0),
    ("proj_fc1_304_256", 32, 1, 3136, 304, 256, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--modes", default="1:1,0:2,0:1")
    ap.add_argument("--single", action="store_true", help="also time every job as its own launch")
    ap.add_argument("--lib", default="", help="load this build of the library (cavp_amd/libcavp_hip_profile.so: CAVP_WGRAD_* knobs)")
    a = ap.parse_args()
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    lib = _lib.load()
    dev = "cuda:0"
    jobs, flops = [], 0.0
    for name, n, h, w, cin, cout, k, p in JOBS:
        x = torch.randn((n, h, w, cin), device=dev).to(torch.bfloat16)
        dy = torch.randn((n, h, w, cout), device=dev).to(torch.bfloat16)
        dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device=dev)
        jobs.append(dict(x=x, dy=dy, dw=dw, kh=k, kw=k, stride=1, pad=p, dil=1, overwrite=True))
        flops += 2.0 * n * h * w * cin * cout * k * k

    def timed(f):
        f(); f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3

    ref = None
    for m in a.modes.split(","):
        mode, stagger = (int(v) for v in m.split(":"))
        assert lib.cavp_set_wgrad_big(mode, stagger) == 0
        us = timed(lambda: T.conv2d_wgrad_group(jobs))
        out = [j["dw"].clone() for j in jobs]
        if ref is None:
            ref = out
        err = max(float((o - r).abs().max() / r.abs().max()) for o, r in zip(out, ref))
        print(f"big mode {mode} schedule {stagger}: group {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s   max rel diff vs first mode {err:.2e}", flush=True)
        if a.single:
            for (name, n, h, w, cin, cout, k, p), j in zip(JOBS, jobs):
                u = timed(lambda: T.conv2d_wgrad(j["x"], j["dy"], j["dw"], kh=k, kw=k, stride=1, pad=p, dil=1, overwrite=True))
                print(f"    {name:24s} {u:8.1f} us  {2.0 * n * h * w * cin * cout * k * k / u / 1e6:7.1f} TF/s", flush=True)
    lib.cavp_set_wgrad_big(0, 2)


if __name__ == "__main__":
    main()
