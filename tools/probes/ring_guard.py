"""Does igemm tile 11 (4-stage ring) write outside its output?  Output placed inside a sentinel-filled buffer."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from cavp_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    for (cin, cout, hw, n) in ((256, 8, 12, 6), (304, 304, 12, 6), (304, 304, 1, 6), (256, 304, 12, 6)):
        x = torch.randn(n, hw, hw, cin, device=dev).to(dt)
        w = (torch.randn(cout, 1, 1, cin, device=dev) * cin ** -0.5).to(dt)
        numel = n * hw * hw * cout
        for tile in (3, 11):
            big = torch.full((numel + 2 * 65536,), 777.0, device=dev, dtype=dt)
            out = big[65536:65536 + numel].view(n, hw, hw, cout)
            ops.conv2d(x, w, out, tile=tile)
            torch.cuda.synchronize()
            lo_bad = int((big[:65536] != 777.0).sum()); hi_bad = int((big[65536 + numel:] != 777.0).sum())
            print(str(dt)[6:], (cin, cout, hw, n), "tile", tile, "stray writes below/above:", lo_bad, hi_bad,
                  "first above" if hi_bad else "", (big[65536 + numel:] != 777.0).nonzero()[:4].flatten().tolist() if hi_bad else "")
