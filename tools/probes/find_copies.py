"""Where do the device-to-device copies / fills of a training step come from?  (torch.profiler, eager step of bench.py's model)
usage: python tools/probes/find_copies.py [c4|c1p] [batch]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cavp_amd.synth import synth_inputs

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")


def c5_copies():
    """config #5 step (model on the graphed autograd node + torch CE + ContrastLoss): device-to-host / host-to-device copies of one
    steady-state step.   python tools/probes/find_copies.py c5copies"""
    import torch.nn.functional as F
    from cavp_amd.contrast import ContrastLoss
    B = 30
    cfg = bench.model_cfg("c5")
    model, _ = bench.build_model(cfg, B, torch.bfloat16, dev)
    model.train()
    model.enable_graphed_autograd()
    image, audio, label = [t.to(dev) for t in synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=100)]
    label_shuf = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=900)[2].to(dev)
    crit = ContrastLoss(temperature=0.1, ignore_idx=255, max_views=512)

    def step():
        model.zero_grad(set_to_none=True)
        out, fus, _ = model(image, audio, None, False)
        loss = F.cross_entropy(out[:B] + out[B:] * 0.0, label, ignore_index=255) + crit(fus[:B], label, fus[B:], label_shuf)
        loss.backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    c = collections.Counter()
    for ev in prof.events():
        if "Memcpy" in ev.name or "memcpy" in ev.name:
            c[ev.name] += 1
    print("copies in one steady-state config-#5 step:", dict(c))
    print("device-to-host copies:", sum(v for k, v in c.items() if "DtoH" in k or "Device -> Host" in k))


if cfgname == "c5copies":
    c5_copies()
else:
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

