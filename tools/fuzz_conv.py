#!/usr/bin/env python3
"""Randomised shapes through cavp_conv2d_nhwc (auto tile choice), its data gradient and its weight gradient against PyTorch on the
CPU (bf16-rounded operands, f32 / f64 accumulation).  GPU box only.  usage: python tools/fuzz_conv.py [--cases 120] [--seed 0]"""
import argparse
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cavp_amd import ops, train_ops as T  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--large", action="store_true", help="shapes that reach the 256x256 tile / split-K / deep-ring plans")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    bad = 0
    for case in range(a.cases):
        dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
        ve = 8 if dt == torch.bfloat16 else 4
        k = rng.choice([1, 1, 3, 3, 3])
        dil = rng.choice([1, 1, 2, 6]) if k == 3 else 1
        stride = rng.choice([1, 1, 2])
        pad = dil * (k // 2) if rng.random() < 0.8 else 0
        n = rng.choice([1, 2, 3, 5])
        h, w = rng.randint(max(1, dil * (k - 1) + 1 - 2 * pad), 40), rng.randint(max(1, dil * (k - 1) + 1 - 2 * pad), 40)
        cin = ve * rng.choice([1, 2, 3, 8, 19, 38, 64])
        cout = ve * rng.choice([1, 2, 5, 6, 32, 38, 40, 64])
        if a.large:
            n = rng.choice([2, 4, 8])
            h, w = rng.randint(14, 64), rng.randint(14, 64)
            cin = rng.choice([64, 128, 256, 304, 512, 1024])
            cout = rng.choice([64, 256, 304, 512, 1216])
            if k == 3 and cin * cout > 300000:
                cin = 256
        ho, wo = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
        if ho <= 0 or wo <= 0:
            continue
        act = rng.choice([ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LEAKY])
        use_res = rng.random() < 0.3
        g = torch.Generator().manual_seed(1000 + case)
        q = lambda t: t.to(dt).float()
        x = q(torch.randn((n, cin, h, w), generator=g))
        wt = q(torch.randn((cout, cin, k, k), generator=g) * (cin * k * k) ** -0.5)
        sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        res = q(torch.randn((n, cout, ho, wo), generator=g)) if use_res else None
        ref = F.conv2d(x.double(), wt.double(), None, stride, pad, dil) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
        if use_res:
            ref = ref + res.double()
        ref = {ops.ACT_NONE: ref, ops.ACT_RELU: ref.relu(), ops.ACT_LEAKY: F.leaky_relu(ref, 0.01)}[act].float()
        xd = x.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)
        wp = ops.pack_weight(wt.to(DEV), dt)
        out = torch.empty((n, ho, wo, cout), dtype=dt, device=DEV)
        desc = f"case {case}: {str(dt)[6:]} N{n} {h}x{w} {cin}->{cout} k{k} s{stride} p{pad} d{dil} act{act} res{int(use_res)}"
        try:
            ops.conv2d(xd, wp, out, kh=k, kw=k, stride=stride, pad=pad, dil=dil, scale=sc.to(DEV), shift=sh.to(DEV),
                       residual=res.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV) if use_res else None, act=act)
            tol = (2e-2 if dt == torch.bfloat16 else 2e-4) * max(1.0, float(ref.abs().max()))
            err = float((out.float().cpu().permute(0, 3, 1, 2) - ref).abs().max())
            ok_f = err <= tol
            # data gradient and weight gradient of the plain conv
            gy = q(torch.randn((n, cout, ho, wo), generator=g))
            xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
            F.conv2d(xr, wr, None, stride, pad, dil).backward(gy.double())
            gyd = gy.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)
            ok_d = ok_w = True
            e_d = e_w = 0.0
            if pad <= dil * (k - 1):
                dx = torch.empty((n, h, w, cin), dtype=dt, device=DEV)
                T.conv2d_dgrad(gyd, T.pack_weight_dgrad(wt.to(DEV), dt), dx, kh=k, kw=k, stride=stride, pad=pad, dil=dil)
                e_d = float((dx.float().cpu().permute(0, 3, 1, 2) - xr.grad.float()).abs().max())
                ok_d = e_d <= (3e-2 if dt == torch.bfloat16 else 3e-4) * max(1.0, float(xr.grad.abs().max()))
            dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device=DEV)
            T.conv2d_wgrad(xd, gyd, dw, kh=k, kw=k, stride=stride, pad=pad, dil=dil)
            e_w = float((dw.cpu().permute(0, 3, 1, 2) - wr.grad.float()).abs().max())
            ok_w = e_w <= (2e-3 if dt == torch.bfloat16 else 2e-4) * max(1.0, float(wr.grad.abs().max()))
            if not (ok_f and ok_d and ok_w):
                bad += 1
                print("FAIL", desc, f"fwd {err:.3g} dgrad {e_d:.3g} wgrad {e_w:.3g}", flush=True)
        except Exception as ex:  # noqa: BLE001
            msg = str(ex)
            if "UNSUPPORTED" in msg.upper() or "unsupported" in msg:
                print("unsupported", desc, msg[:80])
            else:
                bad += 1
                print("ERROR", desc, msg[:160], flush=True)
    print(f"fuzz: {a.cases} cases, {bad} bad")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
