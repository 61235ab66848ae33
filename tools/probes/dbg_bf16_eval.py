import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_train_model import _build, DEV
from cavp_amd.synth import synth_inputs
cfg = dict(C=3, B=8, hw=(64, 64), lds=[False, False, False])
image, audio, label = synth_inputs(8, cfg["hw"], audio_batch=8, num_classes=3, seed=5)
taps = {}
for dt in (torch.float32, torch.bfloat16):
    m, sd = _build(cfg, dt)
    m.eval()
    t = {}
    with torch.no_grad():
        out, fus, pack = m._forward_hip(image.to(DEV), audio.to(DEV), duplicate_visual=False, taps=t)
    t["out"] = out; t["fus"] = fus
    taps[dt] = {k: v.float().cpu() for k, v in t.items()}
for k in taps[torch.float32]:
    a, b = taps[torch.float32][k].double().flatten(), taps[torch.bfloat16][k].double().flatten()
    print(f"{k:14s} cos {float((a@b)/(a.norm()*b.norm())):.6f} relerr {float((a-b).norm()/a.norm()):.4f}")
