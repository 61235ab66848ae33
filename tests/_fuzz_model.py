#!/usr/bin/env python3
"""Random model-level configurations (batch, image extent incl. odd sizes, classes, dilation flags): MI355X eval forward and training
step (f32) against the CPU oracle.  GPU box only.  usage: python tests/_fuzz_model.py [--cases 6] [--seed 0]"""
import argparse
import os
import random
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cavp_amd.synth import synth_inputs, synth_state_dict  # noqa: E402
from oracle import cavp_oracle as O  # noqa: E402  (checker only)

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=6)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    from cavp_amd.cavp_model import CAVP
    from cavp_amd import train_ops as T
    rng = random.Random(a.seed)
    bad = 0
    for case in range(a.cases):
        B = rng.choice([2, 3, 4])
        hw = (rng.choice([33, 48, 64, 70, 96]), rng.choice([40, 64, 81, 96, 112]))
        C = rng.choice([2, 3, 7, 22, 71])
        lds = rng.choice([[False, False, False], [False, True, True], [False, False, True]])
        args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=lds, audio_backbone="vgg", num_classes=C,
                                     batch_size=B, local_rank="cpu")
        m = CAVP(50, None, num_classes=C, args=args)
        sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
        m.load_state_dict(sd, strict=True)
        m.to(DEV)
        image, audio, label = synth_inputs(B, hw, audio_batch=2 * B, num_classes=C, seed=100 + case)
        desc = f"case {case}: B{B} {hw} C{C} lds{[int(v) for v in lds]}"
        try:
            m.eval()
            with torch.no_grad():
                out = m(image.to(DEV), audio[:B].to(DEV), eval_mode=True)[0].cpu()
                ref = O.cavp_forward(sd, image, audio[:B], lds, eval_mode=True)[0]
            e_eval = float((out - ref).abs().max())
            m.train()
            o2, _, _ = m(image.to(DEV), audio.to(DEV), None, False)
            loss, dl = T.ce_loss(o2.detach(), label.to(DEV), B)
            o2.backward(dl)
            params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
            sd2 = dict(sd)
            sd2.update(params)
            ro = O.cavp_forward(sd2, image, audio, lds, eval_mode=False)[0]
            rl = O.ce_loss_train(ro, label, B)
            rl.backward()
            e_loss = abs(float(loss.item()) - float(rl.item()))
            mine = dict(m.named_parameters())
            cos = []
            for k, p in params.items():
                if p.grad is None or float(p.grad.norm()) < 1e-9:
                    continue
                g1, g2 = mine[k].grad.detach().double().cpu().flatten(), p.grad.double().flatten()
                cos.append(float(g1 @ g2 / (g1.norm() * g2.norm() + 1e-30)))
            ok = e_eval <= 1e-3 and e_loss <= 2e-4 * max(1.0, abs(float(rl.item()))) and min(cos) >= 0.97
            print(("ok  " if ok else "FAIL"), desc, f"eval {e_eval:.2e} loss {e_loss:.2e} grad cosine min {min(cos):.4f} median {sorted(cos)[len(cos) // 2]:.6f}",
                  flush=True)
            bad += 0 if ok else 1
        except Exception as ex:  # noqa: BLE001
            bad += 1
            print("ERROR", desc, str(ex)[:200], flush=True)
        del m
        torch.cuda.empty_cache()
    print(f"fuzz_model: {a.cases} cases, {bad} bad")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
