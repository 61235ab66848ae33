#!/usr/bin/env python3
"""Micro-benchmark of cavp_conv2d_nhwc on the CAVP layer shapes (B=32, C1').  GPU box only.
usage: python tools/bench_conv.py [--dtype bf16|f32] [--variants 0,101,1001,...] [--shapes name,...]
A variant is the `tile` knob of cavp_conv_desc: tile id + 100*(profiling: 1 = no operand loads, 2 = no MFMAs) + 1000*(direct epilogue); 0 = auto."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cavp_amd import ops  # noqa: E402

SHAPES = {
    # name: (N, H, W, Cin, Cout, k, stride, pad, dil, residual)
    "head0_3x3_304_256@56": (32, 56, 56, 304, 256, 3, 1, 1, 1, False),
    "stem2_3x3_64_128@112": (32, 112, 112, 64, 128, 3, 1, 1, 1, False),
    "l1_1x1_64_256@56+res": (32, 56, 56, 64, 256, 1, 1, 0, 1, True),
    "l1_1x1_256_64@56": (32, 56, 56, 256, 64, 1, 1, 0, 1, False),
    "l1_3x3_64_64@56": (32, 56, 56, 64, 64, 3, 1, 1, 1, False),
    "l2_1x1_128_512@28+res": (32, 28, 28, 128, 512, 1, 1, 0, 1, True),
    "l3_3x3_256_256@14": (32, 14, 14, 256, 256, 3, 1, 1, 1, False),
    "l3_1x1_1024_256@14": (32, 14, 14, 1024, 256, 1, 1, 0, 1, False),
    "l4_3x3_512_512@14d2": (32, 14, 14, 512, 512, 3, 1, 2, 2, False),
    "l4_1x1_512_2048@14+res": (32, 14, 14, 512, 2048, 1, 1, 0, 1, True),
    "aspp_3x3_2048_256@14d6": (32, 14, 14, 2048, 256, 3, 1, 6, 6, False),
    "ca_fc1_304_1216@3136": (32, 1, 3136, 304, 1216, 1, 1, 0, 1, False),
    "ca_fc2_1216_304@3136+res": (32, 1, 3136, 1216, 304, 1, 1, 0, 1, True),
    "ca_q_304_304@3136": (32, 1, 3136, 304, 304, 1, 1, 0, 1, False),
    "a_fc0_12288_4096@1": (32, 1, 1, 12288, 4096, 1, 1, 0, 1, False),
    # train-step batch (2B = 64 fused rows) of the decoder head / token layers, and a few more backbone shapes
    "T_head0_3x3_304_256@56": (64, 56, 56, 304, 256, 3, 1, 1, 1, False),
    "T_head0dg_3x3_256_304@56": (64, 56, 56, 256, 304, 3, 1, 1, 1, False),
    "T_head1_3x3_256_256@56": (64, 56, 56, 256, 256, 3, 1, 1, 1, False),
    "T_ca_fc1_304_1216": (64, 1, 3136, 304, 1216, 1, 1, 0, 1, False),
    "T_ca_fc2_1216_304": (64, 1, 3136, 1216, 304, 1, 1, 0, 1, True),
    "T_ca_q_304_304": (64, 1, 3136, 304, 304, 1, 1, 0, 1, False),
    "T_stem1_3x3_64_64@112": (32, 112, 112, 64, 64, 3, 1, 1, 1, False),
    "T_stem2dg_3x3_128_64@112": (32, 112, 112, 128, 64, 3, 1, 1, 1, False),
    "T_l2_3x3_128_128@28": (32, 28, 28, 128, 128, 3, 1, 1, 1, False),
    "T_l2_1x1_512_128@28": (32, 28, 28, 512, 128, 1, 1, 0, 1, False),
    "T_l3_1x1_256_1024@14": (32, 14, 14, 256, 1024, 1, 1, 0, 1, True),
    "T_l4_1x1_2048_512@14": (32, 14, 14, 2048, 512, 1, 1, 0, 1, False),
    "T_aspp_1x1_2048_256@14": (32, 14, 14, 2048, 256, 1, 1, 0, 1, False),
    "T_cls_1x1_256_8@56": (64, 56, 56, 256, 8, 1, 1, 0, 1, False),
    "T_clsdg_1x1_8_256@56": (64, 56, 56, 8, 256, 1, 1, 0, 1, False),
    # PVTv2-B5 (config #4, B = 8 -> 16 frames of 512 x 512): token linears of a stage-3 / stage-2 / stage-1 block
    "P3_q_320_320": (16, 1, 1024, 320, 320, 1, 1, 0, 1, False),
    "P3_fc1_320_1280": (16, 1, 1024, 320, 1280, 1, 1, 0, 1, False),
    "P3_fc2_1280_320": (16, 1, 1024, 1280, 320, 1, 1, 0, 1, True),
    "P3_kv_320_640": (16, 1, 256, 320, 640, 1, 1, 0, 1, False),
    "P3_sr_1280_320": (16, 1, 256, 1280, 320, 1, 1, 0, 1, False),
    "P3_srconv_k2s2_320": (16, 32, 32, 320, 320, 2, 2, 0, 1, False),
    "P2_srconv_k4s4_128": (16, 64, 64, 128, 128, 4, 4, 0, 1, False),
    "P1_srconv_k8s8_64": (16, 128, 128, 64, 64, 8, 8, 0, 1, False),
    "P2_sr_2048_128": (16, 1, 256, 2048, 128, 1, 1, 0, 1, False),
    "P1_sr_4096_64": (16, 1, 256, 4096, 64, 1, 1, 0, 1, False),
    "P2_q_128_128": (16, 1, 4096, 128, 128, 1, 1, 0, 1, False),
    "P2_fc1_128_1024": (16, 1, 4096, 128, 1024, 1, 1, 0, 1, False),
    "P2_fc2_1024_128": (16, 1, 4096, 1024, 128, 1, 1, 0, 1, True),
    "P1_fc1_64_512": (16, 1, 16384, 64, 512, 1, 1, 0, 1, False),
    "P1_fc2_512_64": (16, 1, 16384, 512, 64, 1, 1, 0, 1, True),
    "P4_fc1_512_2048": (16, 1, 256, 512, 2048, 1, 1, 0, 1, False),
    "P4_fc2_2048_512": (16, 1, 256, 2048, 512, 1, 1, 0, 1, True),
    # skinny-K pointwise layers of the backbone (round 6: 1.2 .. 1.9 TB/s in the step): forward expansions / reductions and the
    # stride-1 data gradients of the same layers (a dgrad of Cout -> Cin is a forward conv Cin <- Cout)
    "S_l1_64_256@56": (32, 56, 56, 64, 256, 1, 1, 0, 1, False),
    "S_l1_128_256@56": (32, 56, 56, 128, 256, 1, 1, 0, 1, False),
    "S_l1_128_64@56": (32, 56, 56, 128, 64, 1, 1, 0, 1, False),
    "S_l1_256_64@56": (32, 56, 56, 256, 64, 1, 1, 0, 1, False),
    "S_l1_256_128@56": (32, 56, 56, 256, 128, 1, 1, 0, 1, False),
    "S_l2_128_512@28": (32, 28, 28, 128, 512, 1, 1, 0, 1, False),
    "S_l2_512_128@28": (32, 28, 28, 512, 128, 1, 1, 0, 1, False),
    "S_l2_512_256@28": (32, 28, 28, 512, 256, 1, 1, 0, 1, False),
    "S_clsdg_8_256@56": (64, 56, 56, 8, 256, 1, 1, 0, 1, False),
    "S_red_48_256@56": (32, 56, 56, 48, 256, 1, 1, 0, 1, False),
    # K sweep at a fixed output (anatomy of the per-tile fixed costs)
    "K_64_1216": (32, 1, 3136, 64, 1216, 1, 1, 0, 1, False),
    "K_128_1216": (32, 1, 3136, 128, 1216, 1, 1, 0, 1, False),
    "K_64_256": (32, 1, 3136, 64, 256, 1, 1, 0, 1, False),
    "K_64_256hw": (32, 56, 56, 64, 256, 1, 1, 0, 1, False),
    "K_64_1280": (32, 1, 3136, 64, 1280, 1, 1, 0, 1, False),
    "K_64_640": (32, 1, 3136, 64, 640, 1, 1, 0, 1, False),
    "K_640_1216": (32, 1, 3136, 640, 1216, 1, 1, 0, 1, False),
    "K_1280_1216": (32, 1, 3136, 1280, 1216, 1, 1, 0, 1, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--aux", type=int, default=0, help="1: fc1-style epilogue (GELU + gelu' stored, no scale/shift); 2: fc2-dgrad style (result x aux)")
    ap.add_argument("--ldpad", type=int, default=0, help="round the channel stride of x / y / residual up to a multiple of this many elements")
    ap.add_argument("--plain", type=int, default=0, help="1: no scale / shift / residual / activation (the training forward's conv); 2: the same + fused BatchNorm tile statistics")
    ap.add_argument("--lib", default="", help="load this build of the library (cavp_amd/libcavp_hip_profile.so: CAVP_IGEMM_* knobs; "
                                              "CAVP_IGEMM_DBG=256 prints the s_memtime timeline of workgroup 0 / wave 0 and the plan)")
    a = ap.parse_args()
    from cavp_amd import _lib
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda:0"
    variants = [int(v) for v in a.variants.split(",")]
    names = [n for n in SHAPES if not a.shapes or any(s in n for s in a.shapes.split(","))]
    print(f"{'shape':28s} " + " ".join(f"{'v' + str(v):>22s}" for v in variants))
    for name in names:
        n, h, w, cin, cout, k, s, p, d, res = SHAPES[name]
        ho, wo = (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1
        def padded(shape):   # [..., C] view of a [..., ld] buffer
            c = shape[-1]
            ld = (c + a.ldpad - 1) // a.ldpad * a.ldpad if a.ldpad else c
            return torch.randn(shape[:-1] + (ld,), device=dev).to(dt)[..., :c]
        x = padded((n, h, w, cin))
        wt = (torch.randn((cout, k, k, cin), device=dev) * (cin * k * k) ** -0.5).to(dt)
        y = padded((n, ho, wo, cout))
        r = padded((n, ho, wo, cout)) if res else None
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        flops = 2.0 * n * ho * wo * cout * cin * k * k
        nbytes = (x.numel() + wt.numel() + y.numel() * (2 if res else 1)) * x.element_size()
        cells, ref = [], None
        for v in variants:
            try:
                if a.aux:
                    auxt = torch.randn_like(y) if a.aux == 2 else torch.empty_like(y)
                    f = lambda: ops.conv2d(x, wt, y, kh=k, kw=k, stride=s, pad=p, dil=d, shift=sh,
                                           act=ops.ACT_GELU if a.aux == 1 else ops.ACT_NONE, tile=v, aux=auxt, aux_mode=a.aux)
                elif a.plain:
                    f = lambda: ops.conv2d(x, wt, y, kh=k, kw=k, stride=s, pad=p, dil=d, tile=v, want_tile_stats=a.plain == 2)
                else:
                    f = lambda: ops.conv2d(x, wt, y, kh=k, kw=k, stride=s, pad=p, dil=d, scale=sc, shift=sh, residual=r,
                                           act=ops.ACT_RELU, tile=v)
                f(); f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    f()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / a.reps * 1e3
                yy = y.float()
                if ref is None:
                    ref = yy.clone()
                    ok = ""
                else:
                    err = float((yy - ref).abs().max())
                    ok = "" if err <= 2e-2 * max(1.0, float(ref.abs().max())) else f" !ERR{err:.2g}"
                cells.append(f"{us:7.1f}us {flops / us / 1e6:6.1f}TF {nbytes / us / 1e3:5.0f}GB{ok}")
            except Exception as ex:  # noqa: BLE001
                cells.append(f"{'fail: ' + str(ex)[:14]:>22s}")
        print(f"{name:28s} " + " ".join(f"{c:>22s}" for c in cells), flush=True)
        if int(os.environ.get("CAVP_IGEMM_DBG", "0")) & 256:
            import ctypes
            buf = (ctypes.c_ulonglong * 8)()
            lib = _lib.load()
            if lib.cavp_prof_igemm_timeline(buf) == 0 and buf[0]:
                t, it = buf[0], max(buf[1], 1)
                print(f"    timeline (shader cycles, physical workgroup 0 / wave 0, last launch): {t} tile(s), {buf[1]} K iterations, kernel {buf[7]}; "
                      f"entry->first tile {buf[2]}; per tile: set-up {buf[3] / t:.0f}, first stage landed {buf[4] / t:.0f}, "
                      f"K loop after that {buf[5] / t:.0f} ({buf[5] / it:.0f} per K iteration), epilogue {buf[6] / t:.0f}")


if __name__ == "__main__":
    main()
