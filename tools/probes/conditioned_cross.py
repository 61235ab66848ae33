#!/usr/bin/env python3
"""Is a move of tests/test_gpu_conditioned_parity.py's bf16 figures the KERNELS' doing or the conditioned WEIGHTS'?  The test first trains the
synthetic weights for 100 f32 steps with the tree's own kernels, so any change of an f32 summation order gives it different weights to be measured on.
  condition <tree> <out.pt>   100 deterministic f32 steps with <tree>/cavp_amd -> state_dict file
  compare   <tree> <sd.pt>    the test's bf16 (and f32) step-vs-oracle figures of <tree>/cavp_amd on those weights
GPU box only (`tools/ab_r06.sh cross` drives the 2 x 2 cross)."""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode, tree, path = sys.argv[1], os.path.abspath(sys.argv[2]), sys.argv[3]
sys.path.insert(0, tree)
if tree != REPO:
    sys.path.insert(1, REPO)   # oracle/ (identical in both trees) and tests/
import torch  # noqa: E402

spec = importlib.util.spec_from_file_location("condp", os.path.join(REPO, "tests", "test_gpu_conditioned_parity.py"))
P = importlib.util.module_from_spec(spec)
spec.loader.exec_module(P)
import cavp_amd  # noqa: E402
assert os.path.abspath(os.path.dirname(cavp_amd.__file__)).startswith(tree), cavp_amd.__file__

if mode == "condition":
    sd, first, last = P.conditioned.__wrapped__() if hasattr(P.conditioned, "__wrapped__") else P.conditioned.__pytest_wrapped__.obj()
    torch.save(sd, path)
    print(f"conditioned with {os.path.basename(tree)}: loss {first:.4f} -> {last:.4f}", flush=True)
else:
    sd = torch.load(path)
    cond = (sd, 0.0, 0.0)
    ostep = (P.oracle_step.__pytest_wrapped__.obj if hasattr(P.oracle_step, "__pytest_wrapped__") else P.oracle_step.__wrapped__)(cond)
    for dt in (torch.float32, torch.bfloat16):
        r = P._compare(sd, dt, ostep)
        print(f"weights {os.path.basename(path)}, kernels {os.path.basename(tree)}, {dt}: " +
              ", ".join(f"{k} {r[k]:.5f}" for k in ("logits_rel", "whole_cos", "cos_med", "cos_p05")), flush=True)
