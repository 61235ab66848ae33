import sys, types, collections
sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity
from bench import build_model, model_cfg
from cavp_amd.synth import synth_inputs
dev = torch.device("cuda:0")
cfg = model_cfg("c1p")
B = 32
m, sd = build_model(cfg, B, torch.bfloat16, dev)
m.train()
image, audio, label = [t.to(dev) for t in synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=1)]
for _ in range(2):
    m.train_step(image, audio, label)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    m.train_step(image, audio, label)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::cat", "aten::zeros", "aten::index_select",
                   "aten::add_", "aten::mul", "aten::_foreach_add_", "aten::stack", "aten::contiguous", "aten::flip"):
        st = [s for s in (ev.stack or []) if "cavp_amd" in s]
        cnt[(ev.name, st[0] if st else "?", st[1] if len(st) > 1 else "")] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(v, k)
