"""cProfile of the config-#5 eager step (bench.py's run_step): python tools/probes/c5_hostprof.py"""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
import bench
from cavp_amd.synth import synth_inputs
from cavp_amd.contrast import ContrastLoss
B = 30
cfg = bench.model_cfg("c5") if "c5" in bench.model_cfg.__code__.co_consts else bench.model_cfg("c1p")
dev = torch.device("cuda:0")
model, _ = bench.build_model(cfg, B, torch.bfloat16, dev)
model.train()
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
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()     # host time to ISSUE 5 steps
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e3 * (t1 - t0) / 5:.1f} ms/step, complete {1e3 * (t2 - t0) / 5:.1f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
