"""igemm tile 11 (64x64, 4-stage ring) against torch for short K loops: python tools/probes/ring_check.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from cavp_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    for (cin, cout, k, hw, n) in ((64, 64, 3, 12, 2), (304, 256, 1, 12, 3), (320, 640, 1, 16, 8), (128, 128, 3, 14, 4), (256, 64, 1, 9, 2),
                                  (512, 128, 1, 14, 2), (64, 256, 1, 14, 6)):
        x = torch.randn(n, cin, hw, hw)
        w = torch.randn(cout, cin, k, k) * (cin * k * k) ** -0.5
        xq, wq = x.to(dt).float(), w.to(dt).float()
        ref = F.conv2d(xq, wq, None, 1, k // 2)
        xd = xq.permute(0, 2, 3, 1).contiguous().to(dev, dt)
        wd = wq.permute(0, 2, 3, 1).contiguous().to(dev, dt)
        res = []
        for tile in (0, 3, 11):
            out = torch.empty((n, hw, hw, cout), dtype=dt, device=dev)
            try:
                ops.conv2d(xd, wd, out, kh=k, kw=k, pad=k // 2, tile=tile)
                torch.cuda.synchronize()
                e = float((out.float().cpu().permute(0, 3, 1, 2) - ref).norm() / ref.norm())
            except Exception as ex:
                e = str(ex)[:40]
            res.append((tile, e))
        iters = k * k * ((cin + (63 if dt == torch.bfloat16 else 31)) // (64 if dt == torch.bfloat16 else 32))
        print(str(dt)[6:], (cin, cout, k, hw, n), "iters", iters, res)
