import sys, types, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_train_model import _build, _oracle_grads, DEV
from cavp_amd.synth import synth_inputs
from cavp_amd import train_ops as T
cfg = dict(C=3, B=8, hw=(64, 64), lds=[False, False, False])
m, sd = _build(cfg, torch.bfloat16)
B = cfg["B"]
image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=5)
out, fus, pack = m(image.to(DEV), audio.to(DEV), None, False)
loss, dl = T.ce_loss(out.detach(), label.to(DEV), B)
out.backward(dl)
ref_out, ref_loss, ref_g = _oracle_grads(sd, cfg, image, audio, label)
print("out err", float((out.detach().cpu() - ref_out).abs().max()), float(ref_out.abs().max()))
params = dict(m.named_parameters())
for k, g in ref_g.items():
    a, b = params[k].grad.detach().double().cpu().flatten(), g.double().flatten()
    if float(b.norm()) < 1e-8: continue
    cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
    print(f"{cos:8.4f} {float(a.norm()):10.4g} {float(b.norm()):10.4g} {k}")
