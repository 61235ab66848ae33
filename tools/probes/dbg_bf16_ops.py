import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from cavp_amd import ops, train_ops as T
DEV = "cuda:0"
def q(t, dt): return t.to(dt).float()
def nhwc(x, dt): return x.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)
def cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30)), float((a - b).abs().max()), float(b.abs().max())
for dt in (torch.bfloat16, torch.float32):
    for (n, h, w, cin, cout, k, p) in [(16, 16, 16, 256, 256, 3, 1), (2, 16, 16, 256, 256, 3, 1), (16, 16, 16, 256, 256, 1, 0),
                                      (16, 16, 16, 128, 128, 3, 1), (16, 16, 16, 256, 128, 3, 1), (16, 16, 16, 128, 256, 3, 1), (4, 16, 16, 256, 256, 3, 1)]:
        g = torch.Generator().manual_seed(0)
        x = q(torch.randn(n, cin, h, w, generator=g), dt).requires_grad_(True)
        wt = q(torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5, dt).requires_grad_(True)
        y = F.conv2d(x, wt, None, 1, p)
        dy = q(torch.randn(y.shape, generator=g), dt)
        y.backward(dy)
        dx = torch.empty((n, h, w, cin), dtype=dt, device=DEV)
        T.conv2d_dgrad(nhwc(dy, dt), T.pack_weight_dgrad(wt.detach().to(DEV), dt), dx, kh=k, kw=k, stride=1, pad=p, dil=1)
        dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device=DEV)
        T.conv2d_wgrad(nhwc(x.detach(), dt), nhwc(dy, dt), dw, kh=k, kw=k, stride=1, pad=p, dil=1)
        print(dt, (n, h, w, cin, cout, k), "dgrad cos/err/max", cos(dx.permute(0, 3, 1, 2), x.grad), "wgrad", cos(dw.permute(0, 3, 1, 2), wt.grad))
