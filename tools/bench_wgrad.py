#!/usr/bin/env python3
"""Micro-benchmark of cavp_conv2d_wgrad_nhwc on the CAVP train-step layer shapes (B=32 images / 64 fused rows).
GPU box only.  usage: python tools/bench_wgrad.py [--dtype bf16|f32] [--shapes name,...] [--splitk 0]
CAVP_WGRAD_DBG=<bits> (profiling only) removes pieces of the kernel: 1 loads out of range, 2 no MFMAs, 4 no DMA, 8 no epilogue."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cavp_amd import train_ops as T  # noqa: E402

SHAPES = {
    # name: (N, H, W, Cin, Cout, k, stride, pad, dil)
    "head0_3x3_304_256@56": (64, 56, 56, 304, 256, 3, 1, 1, 1),
    "head1_3x3_256_256@56": (64, 56, 56, 256, 256, 3, 1, 1, 1),
    "ca_fc1_304_1216": (64, 1, 3136, 304, 1216, 1, 1, 0, 1),
    "ca_fc2_1216_304": (64, 1, 3136, 1216, 304, 1, 1, 0, 1),
    "ca_q_304_304": (64, 1, 3136, 304, 304, 1, 1, 0, 1),
    "stem1_3x3_64_64@112": (32, 112, 112, 64, 64, 3, 1, 1, 1),
    "stem2_3x3_64_128@112": (32, 112, 112, 64, 128, 3, 1, 1, 1),
    "l1_1x1_64_256@56": (32, 56, 56, 64, 256, 1, 1, 0, 1),
    "l1_1x1_256_64@56": (32, 56, 56, 256, 64, 1, 1, 0, 1),
    "l1_3x3_64_64@56": (32, 56, 56, 64, 64, 3, 1, 1, 1),
    "l2_3x3_128_128@28": (32, 28, 28, 128, 128, 3, 1, 1, 1),
    "l2_1x1_128_512@28": (32, 28, 28, 128, 512, 1, 1, 0, 1),
    "l3_3x3_256_256@14": (32, 14, 14, 256, 256, 3, 1, 1, 1),
    "l3_1x1_1024_256@14": (32, 14, 14, 1024, 256, 1, 1, 0, 1),
    "l4_3x3_512_512@14d2": (32, 14, 14, 512, 512, 3, 1, 2, 2),
    "l4_1x1_512_2048@14": (32, 14, 14, 512, 2048, 1, 1, 0, 1),
    "aspp_3x3_2048_256@14d6": (32, 14, 14, 2048, 256, 3, 1, 6, 6),
    "a_fc0_12288_4096": (64, 1, 1, 12288, 4096, 1, 1, 0, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--splitk", default="0")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--big", default="0:2", help="cavp_set_wgrad_big mode:pipelined (mode 0 auto, 1 never, 2 always)")
    ap.add_argument("--lib", default="", help="load this build of the library (cavp_amd/libcavp_hip_profile.so: CAVP_WGRAD_DBG knobs)")
    a = ap.parse_args()
    from cavp_amd import _lib
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    assert _lib.load().cavp_set_wgrad_big(*(int(v) for v in a.big.split(":"))) == 0
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda:0"
    sks = [int(v) for v in a.splitk.split(",")]
    names = [n for n in SHAPES if not a.shapes or any(s in n for s in a.shapes.split(","))]
    print(f"{'shape':28s} " + " ".join(f"{'splitk=' + str(v):>24s}" for v in sks))
    tot = [0.0] * len(sks)
    for name in names:
        n, h, w, cin, cout, k, s, p, d = SHAPES[name]
        ho, wo = (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1
        x = torch.randn((n, h, w, cin), device=dev).to(dt)
        dy = torch.randn((n, ho, wo, cout), device=dev).to(dt)
        dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device=dev)
        flops = 2.0 * n * ho * wo * cout * cin * k * k
        cells = []
        for j, sk in enumerate(sks):
            f = lambda: T.conv2d_wgrad(x, dy, dw, kh=k, kw=k, stride=s, pad=p, dil=d, splitk=sk)
            f(); f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            tot[j] += us
            cells.append(f"{us:8.1f}us {flops / us / 1e6:7.1f}TF")
        print(f"{name:28s} " + " ".join(f"{c:>24s}" for c in cells), flush=True)
        if int(os.environ.get("CAVP_WGRAD_DBG", "0")) & 16:   # profile build: s_memtime timeline of workgroup 0 / wave 0 (conv_wgrad_big.hip)
            import ctypes
            buf = (ctypes.c_ulonglong * 16)()
            lib = _lib.load()
            if lib.cavp_prof_wgrad_timeline(buf) == 0 and buf[0]:
                n = buf[0]
                lab = ["wait A + co half 0", "cluster 0 (8 MFMA + 8 reads)", "wait co half 1 + stage s+1", "barrier", "cluster 1 (8 MFMA + 16 reads + DMA issue)"]
                print(f"    timeline (s_memtime ticks = 100 MHz? see notes; workgroup 0 / wave 0): stages {n}, entry->loop {buf[1]}, loop {buf[2]} "
                      f"({buf[2] / n:.1f} per stage), loop end->last store issued {buf[3]}")
                for i, l in enumerate(lab):
                    print(f"      {l:44s} {buf[4 + i] / max(n - 1, 1):9.1f} per stage  ({100.0 * buf[4 + i] / max(buf[2], 1):5.1f} % of the loop)")
    print(f"{'sum':28s} " + " ".join(f"{t:8.1f}us{'':>14s}" for t in tot))


if __name__ == "__main__":
    main()
