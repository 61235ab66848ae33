#!/usr/bin/env python3
"""Device time of the train-mode BatchNorm forward, fused (cavp_bn_apply_tiles) vs two launches (cavp_bn_finalize_tiles +
cavp_scale_shift_act), per call inside a hipGraph of 20 back-to-back calls (no host launch time in the figure).  GPU box only.
usage: python tools/bench_bn_apply.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cavp_amd import ops, train_ops as T  # noqa: E402

DEV = "cuda:0"


def graph_us(fn, n=20, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


def main():
    print(f"{'rows x C':>16s} {'tiles x rpt':>12s} {'fused us':>9s} {'finalize us':>12s} {'apply us':>9s} {'two launches us':>16s}")
    for rows, c, rpt, res in ((6272, 256, 128, False), (6272, 256, 64, False), (6272, 512, 128, False), (6272, 1024, 128, True),
                              (6272, 2048, 128, True), (6272, 1024, 64, True), (32, 256, 128, False)):
        x = torch.randn(rows, c, device=DEV).to(torch.bfloat16)
        r = torch.randn(rows, c, device=DEV).to(torch.bfloat16) if res else None
        y = torch.empty_like(x)
        tiles = (rows + rpt - 1) // rpt
        ts = torch.rand(tiles, c, 2, device=DEV)
        g, b = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        sc, sh, mu, rs = (torch.empty(c, device=DEV) for _ in range(4))
        fused = graph_us(lambda: T.bn_apply_tiles(ts, tiles, rpt, rows, g, b, 1e-5, 0.1, rm, rv, sc, sh, mu, rs, x, y, ops.ACT_RELU, residual=r))
        fin = graph_us(lambda: T.bn_finalize_tiles(ts, tiles, rpt, rows, g, b, 1e-5, 0.1, rm, rv, sc, sh, mu, rs))
        app = graph_us(lambda: T.scale_shift_act(x, sc, sh, y, ops.ACT_RELU, residual=r))

        def two():
            T.bn_finalize_tiles(ts, tiles, rpt, rows, g, b, 1e-5, 0.1, rm, rv, sc, sh, mu, rs)
            T.scale_shift_act(x, sc, sh, y, ops.ACT_RELU, residual=r)
        both = graph_us(two)
        print(f"{rows:>8d} x {c:<5d} {tiles:>5d} x {rpt:<4d} {fused:9.2f} {fin:12.2f} {app:9.2f} {both:16.2f}", flush=True)


if __name__ == "__main__":
    main()
