import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_train_model import _build, DEV
from cavp_amd.synth import synth_inputs
from cavp_amd import train_ops as T
cfg = dict(C=3, B=8, hw=(64, 64), lds=[False, False, False])
B = cfg["B"]
image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=5)
taps = {}
for dt in (torch.float32, torch.bfloat16):
    m, sd = _build(cfg, dt)
    m._keep_train_pass = True
    out, fus, pack = m(image.to(DEV), audio.to(DEV), None, False)
    loss, dl = T.ce_loss(out.detach(), label.to(DEV), B)
    out.backward(dl)
    torch.cuda.synchronize()
    tp = m._last_train_pass
    taps[dt] = {k: (v.t.float().cpu(), None if v.g is None else v.g.float().cpu()) for k, v in tp.named.items()}
def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))
for k in taps[torch.float32]:
    a, ag = taps[torch.float32][k]; b, bg = taps[torch.bfloat16][k]
    print(f"{k:8s} act cos {cos(a, b):.5f}  grad cos {cos(ag, bg) if ag is not None and bg is not None else float('nan'):.5f}  |g| {float(ag.norm()) if ag is not None else 0:.4g} {float(bg.norm()) if bg is not None else 0:.4g}")
