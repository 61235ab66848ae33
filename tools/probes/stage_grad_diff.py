"""Per-parameter gradient error of the chained stage methods vs the oracle (f32): python tools/probes/stage_grad_diff.py"""
import sys, os, types
sys.path.insert(0, os.getcwd())
import torch
from cavp_amd.synth import synth_inputs, synth_state_dict
from cavp_amd import _lib
from oracle import cavp_oracle as O
DEV = "cuda:0"
CFG = dict(C=3, B=2, hw=(64, 64), lds=[False, False, False])
from models.cavp_model import CAVP
a = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=CFG["lds"], audio_backbone="vgg", num_classes=CFG["C"], batch_size=CFG["B"], local_rank=DEV)
m = CAVP(50, None, num_classes=CFG["C"], audio_backbone_pretrain_path=None, visual_backbone=50, args=a)
sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
m.load_state_dict(sd, strict=True); m.to(DEV); m.train()
_lib.set_deterministic(True, torch.device(DEV))
from cavp_amd import train as _tr
_TPS = []
_orig_bwd = _tr.TrainPass.backward
def _bwd(self):
    _TPS.append(self)
    named = dict(self.named)
    _orig_bwd(self)
    self._dump = {k: (v.g.detach().float().cpu().clone() if v.g is not None else None) for k, v in named.items()}
    self._dumpt = {k: v.t.detach().float().cpu().clone() for k, v in named.items()}
_tr.TrainPass.backward = _bwd
B, hw = 3, (12, 12)
g = torch.Generator().manual_seed(11)
fea_v = torch.randn((B, 304) + hw, generator=g) * 0.5
audio = synth_inputs(B, CFG["hw"], audio_batch=B, num_classes=CFG["C"], seed=8)[1]
perm = torch.tensor([2, 0, 1])
w_out = torch.randn((2 * B, CFG["C"], 48, 48), generator=g)
w_vis = torch.randn((2 * B, 304) + hw, generator=g) * 0.1
sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
rv = fea_v.clone().requires_grad_(True)
ra1 = O.audio_forward(audio, sdg); ra = torch.cat((ra1, ra1[perm]), 0)
rf, rp = O.forward_fusion(torch.cat((rv, rv), 0), ra, sdg)
ro = O.forward_cls(rf, sdg, (48, 48), train=True)
((ro * w_out).sum() + (rp["visual"] * w_vis).sum() + rf.square().mean()).backward()
info = {"shuffle_idx": perm.to(DEV), "mod_idx_map": {}, "image_label": torch.zeros((B, CFG["C"]), device=DEV)}
xv = fea_v.to(DEV).requires_grad_(True)
fa = m.forward_audio(audio.to(DEV), info, ow_flag=False)
fus, pack = m.forward_fusion(torch.cat((xv, xv), 0), fa)
out = m.forward_cls(fus, (48, 48))
((out * w_out.to(DEV)).sum() + (pack["visual"] * w_vis.to(DEV)).sum() + fus.square().mean()).backward()
def rel(a, b):
    return float((a.detach().cpu().double() - b.detach().double()).norm() / max(float(b.detach().double().norm()), 1e-30))
print("fa", rel(fa, ra), "fus", rel(fus, rf), "out", rel(out, ro), "vis", rel(pack["visual"], rp["visual"]), "dx", rel(xv.grad, rv.grad))
rows = []
for k, p in m.named_parameters():
    r = sdg[k].grad if isinstance(sdg.get(k), torch.Tensor) else None
    if r is None or p.grad is None or float(r.norm()) == 0: continue
    rows.append((rel(p.grad, r), k))
for e, k in sorted(rows, reverse=True)[:6] + [r for r in rows if "segment.upsample" in r[1]]:
    print(f"{e:.2e} {k}")

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
torch.save({"g": [tp._dump for tp in _TPS], "t": [tp._dumpt for tp in _TPS]}, f"gpurun_out/r03/stage_dump_{tag}.pt")
