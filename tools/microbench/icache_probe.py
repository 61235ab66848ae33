#!/usr/bin/env python3
"""Does alternating between DIFFERENT kernel instantiations cost time on the same (hot) data?  One small conv shape, launched
back to back (a) on one igemm tile variant, (b) cycling through several variants (different code, same operands), (c) with an
unrelated small kernel (torch fill of 4 KiB) between launches.  GPU box only.  usage: python tools/microbench/icache_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cavp_amd import ops  # noqa: E402

dev = "cuda:0"
dt = torch.bfloat16
SHAPES = {"l3_1x1_1024_256@14": (32, 14, 14, 1024, 256, 1, 1, 0, 1), "l3_3x3_256_256@14": (32, 14, 14, 256, 256, 3, 1, 1, 1),
          "l2_1x1_512_128@28": (32, 28, 28, 512, 128, 1, 1, 0, 1)}
VARS = [3, 1, 2, 4, 11, 13, 14]


def timed(fns, reps=40):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3


for name, (n, h, w, cin, cout, k, s, p, d) in SHAPES.items():
    ho, wo = (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1
    x = torch.randn((n, h, w, cin), device=dev).to(dt)
    wt = (torch.randn((cout, k, k, cin), device=dev) * (cin * k * k) ** -0.5).to(dt)
    y = torch.empty((n, ho, wo, cout), device=dev, dtype=dt)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    small = torch.empty(1024, device=dev)
    mk = lambda v: (lambda: ops.conv2d(x, wt, y, kh=k, kw=k, stride=s, pad=p, dil=d, scale=sc, shift=sh, act=ops.ACT_RELU, tile=v))
    per = {v: timed([mk(v)]) for v in VARS}
    mean_single = sum(per.values()) / len(per)
    cyc = timed([mk(v) for v in VARS])
    fill = timed([lambda: small.fill_(1.0)])
    with_fill = timed([mk(3), lambda: small.fill_(1.0)]) * 2 - fill
    print(f"{name:22s} per variant " + " ".join(f"{v}:{per[v]:.1f}" for v in VARS) + f" | mean {mean_single:.1f} us, cycling through all {cyc:.1f} us"
          f" | variant 3 with a fill kernel between launches {with_fill:.1f} us (fill alone {fill:.1f})", flush=True)
