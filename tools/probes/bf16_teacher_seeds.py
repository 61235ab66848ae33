#!/usr/bin/env python3
"""How much does the worst per-parameter gradient cosine of tests/test_gpu_bf16_parity.py::test_bf16_teacher_forced_layer_by_layer move with
the input seed - and with the tree (usage: python tools/probes/bf16_teacher_seeds.py [tree_dir] [seeds...]; tree_dir = the directory whose
cavp_amd/ is imported, default the repo root; e.g. ab_base)?  GPU box only."""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tree = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 and os.path.isdir(sys.argv[1]) else REPO
seeds = [int(s) for s in sys.argv[1:] if s.isdigit()] or [11, 12, 13, 14, 15, 16]
sys.path.insert(0, tree)
import torch  # noqa: E402

spec = importlib.util.spec_from_file_location("bf16p", os.path.join(REPO, "tests", "test_gpu_bf16_parity.py"))
P = importlib.util.module_from_spec(spec)
spec.loader.exec_module(P)
from cavp_amd.synth import synth_inputs  # noqa: E402

cfg = dict(C=3, B=4, hw=(96, 96), lds=[False, False, False])
for seed in seeds:
    image, audio, label = [t.to("cuda:0") for t in synth_inputs(cfg["B"], cfg["hw"], audio_batch=2 * cfg["B"], num_classes=cfg["C"], seed=seed)]
    rec, report = [], []
    l32, _, g32 = P._train_pass(P._build(cfg, torch.float32), image, audio, label, cfg["B"], cfg["C"], record=rec)
    l16, _, g16 = P._train_pass(P._build(cfg, torch.bfloat16), image, audio, label, cfg["B"], cfg["C"], teacher=rec, report=report)
    stats = []
    for k in g32:
        a, b = g16[k], g32[k]
        if float(b.norm()) == 0.0:
            continue
        stats.append((float((a @ b) / (a.norm() * b.norm())), float(a.norm() / b.norm()), k))
    stats.sort()
    cs = [s[0] for s in stats]
    print(f"{os.path.basename(tree) or 'repo'} seed {seed}: worst 3 {[(round(c, 4), k) for c, _, k in stats[:3]]} median {cs[len(cs) // 2]:.4f} "
          f"below 0.98: {sum(c < 0.98 for c in cs)} of {len(cs)}", flush=True)
