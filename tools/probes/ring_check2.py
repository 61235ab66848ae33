"""epilogue features on igemm tile 11 (4-stage ring) vs tile 3: python tools/probes/ring_check2.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from cavp_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
def rel(a, b):
    return float((a.float() - b.float()).norm() / max(float(b.float().norm()), 1e-30))
for dt in (torch.float32, torch.bfloat16):
    for (cin, cout, k, hw, n) in ((304, 256, 1, 12, 6), (256, 8, 1, 12, 6), (64, 64, 3, 12, 2), (304, 1216, 1, 12, 6)):
        x = torch.randn(n, hw, hw, cin, device=dev).to(dt)
        w = (torch.randn(cout, k, k, cin, device=dev) * (cin * k * k) ** -0.5).to(dt)
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        resd = torch.randn(n, hw, hw, cout, device=dev).to(dt)
        half = torch.randn(n // 2, hw, hw, cout, device=dev).to(dt)
        nb = torch.randn(n, cout, device=dev)
        cases = {
            "scale+shift+relu+res": dict(scale=sc, shift=sh, residual=resd, act=ops.ACT_RELU),
            "periodic res": dict(shift=sh, residual=half, res_rows=(n // 2) * hw * hw),
            "nbias+leaky": dict(nbias=nb, shift=sh, act=ops.ACT_LEAKY),
            "gelu+aux": dict(shift=sh, act=ops.ACT_GELU, aux_mode=1),
            "mul aux": dict(aux_mode=2),
            "stats": dict(want_tile_stats=True),
        }
        for name, kw in cases.items():
            outs = []
            for tile in (3, 11):
                out = torch.empty((n, hw, hw, cout), dtype=dt, device=dev)
                kw2 = dict(kw)
                aux = None
                if kw.get("aux_mode") == 1:
                    aux = torch.empty_like(out); kw2["aux"] = aux
                if kw.get("aux_mode") == 2:
                    aux = resd; kw2["aux"] = aux
                try:
                    r = ops.conv2d(x, w, out, kh=k, kw=k, pad=k // 2, tile=tile, **kw2)
                    torch.cuda.synchronize()
                    st = r[1] if isinstance(r, tuple) else None
                    outs.append((out, aux if kw.get("aux_mode") == 1 else None, st))
                except Exception as ex:
                    outs.append(str(ex)[:60])
            if isinstance(outs[0], str) or isinstance(outs[1], str):
                print(str(dt)[6:], (cin, cout, k), name, outs if isinstance(outs[0], str) else outs[1]); continue
            msg = f"out {rel(outs[1][0], outs[0][0]):.2e}"
            if outs[0][1] is not None: msg += f" aux {rel(outs[1][1], outs[0][1]):.2e}"
            if outs[0][2] is not None or outs[1][2] is not None: msg += f" stats {None if outs[0][2] is None else outs[0][2][1:]} vs {None if outs[1][2] is None else outs[1][2][1:]}"
            print(str(dt)[6:], (cin, cout, k), name, msg)
