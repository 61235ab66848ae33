"""Where do the device-to-device copies / fills of a training step come from?  (torch.profiler, eager step of bench.py's model)
usage: python tools/probes/find_copies.py [c4|c1p] [batch]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cavp_amd.synth import synth_inputs

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = bench.model_cfg(cfgname)
dev = torch.device("cuda:0")
model, _ = bench.build_model(cfg, B, torch.bfloat16, dev)
model.train()
image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=100)
image, audio, label = image.to(dev), audio.to(dev), label.to(dev)
with torch.no_grad():
    for _ in range(2):
        model.train_step(image, audio, label)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        model.train_step(image, audio, label)
        torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    n = ev.name
    if n.startswith("aten::copy_") or n.startswith("aten::fill_") or n.startswith("aten::zero_") or n in ("aten::clone", "aten::contiguous", "aten::cat", "aten::index_select", "aten::flip", "aten::stack"):
        st = [s for s in (ev.stack or []) if "cavp_amd" in s or "bench.py" in s]
        cnt[(n, (st[0] if st else "?") + " " + str(ev.input_shapes)[:120])] += 1
for (n, st), c in cnt.most_common(40):
    print(f"{c:5d}  {n:22s} {st}")
