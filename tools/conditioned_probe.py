#!/usr/bin/env python3
"""How far is the bf16 training step from the CPU oracle (f32, pinned to the reference) END TO END, and on which weights?

With the synthetic random weights the batch-statistics BatchNorm trunk amplifies any perturbation (bf16 storage rounding
included) layer by layer - DESIGN.md 6c - so an un-forced bf16-vs-oracle comparison says nothing about the kernels.  This
probe measures the same comparison on CONDITIONED weights: the synthetic set trained for N f32 steps on one synthetic batch with
the repo's own fused optimiser (CAVP.train_step + FusedSGDAdam), then frozen.  It prints, per variant, the relative L2 error of
the train-mode logits and the per-parameter gradient cosine of the bf16 HIP step against the oracle's autograd on those weights;
tests/test_gpu_conditioned_parity.py pins the variant that is used.

GPU box only:  python tools/conditioned_probe.py [--steps 60] [--lr 0.01] [--batch 8] [--hw 96]"""
import argparse
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(sd, cfg, dtype, dev):
    from cavp_amd.cavp_model import CAVP
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=cfg["lds"], audio_backbone="vgg",
                                 num_classes=cfg["C"], batch_size=cfg["B"], local_rank="cpu")
    m = CAVP(50, None, num_classes=cfg["C"], args=args)
    if sd is None:
        from cavp_amd.synth import synth_state_dict
        sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    m.train().to(dev).set_compute_dtype(dtype)
    return m, sd


def blob_batch(cfg, seed):
    from cavp_amd.synth import learnable_inputs
    return learnable_inputs(cfg["B"], cfg["hw"], cfg["C"], seed)


def condition(cfg, steps, lr, dev, damp=1.0, seed=3):
    """Synthetic weights -> `steps` f32 training steps on ONE synthetic batch (fused step + fused SGD / Adam) -> CPU state_dict."""
    from cavp_amd.optim import FusedSGDAdam
    from cavp_amd.synth import synth_inputs
    m, sd = build(None, cfg, torch.float32, dev)
    if damp != 1.0:   # zero_init_residual, relaxed: the last BatchNorm of every bottleneck starts small
        with torch.no_grad():
            for k, p in m.named_parameters():
                if ".bn3.weight" in k:
                    p.mul_(damp)
    opt = None
    losses = []
    for it in range(steps):
        image, audio, label = [t.to(dev) for t in blob_batch(cfg, seed + it % 4)]   # four batches in turn
        loss = m.train_step(image, audio, label)
        if opt is None:
            opt = FusedSGDAdam(m, m._grad_arena, lr, momentum=0.9, weight_decay=1e-4)
        opt.step(lr)
        losses.append(float(loss.item()))
    torch.cuda.synchronize()
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, losses


def compare(sd, cfg, dev, seed=11):
    """bf16 HIP train step vs the oracle's f32 autograd on the weights `sd`: (logits rel L2, loss pair, gradient cosines)."""
    from cavp_amd.synth import synth_inputs
    from oracle import cavp_oracle as O
    image, audio, label = blob_batch(cfg, seed)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    out, _, _ = O.cavp_forward(sd2, image, audio, cfg["lds"], eval_mode=False)
    o_loss = O.ce_loss_train(out, label, cfg["B"])
    o_loss.backward()
    res = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        m, _ = build(sd, cfg, dt, dev)
        loss = m.train_step(image.to(dev), audio.to(dev), label.to(dev), want_pred=True)
        torch.cuda.synchronize()
        pred = m._last_outputs[0].float().cpu()
        ref = out.detach()
        rel = float((pred - ref).norm() / ref.norm())
        cos, nr = [], []
        dot = na = nb = 0.0
        for k, p in m.named_parameters():
            if p.grad is None or params[k].grad is None:
                continue
            a, b = p.grad.double().cpu().flatten(), params[k].grad.double().flatten()
            if float(b.norm()) < 1e-12:
                continue
            dot, na, nb = dot + float(a @ b), na + float(a @ a), nb + float(b @ b)
            cos.append((float((a @ b) / (a.norm() * b.norm() + 1e-300)), k))
            nr.append(float(a.norm() / b.norm()))
        cos.sort()
        nr.sort()
        res[name] = dict(whole_cos=dot / (na * nb) ** 0.5, logits_rel=rel, logit_range=(float(ref.min()), float(ref.max()), float(ref.std())), loss=float(loss.item()), oracle_loss=float(o_loss.item()), cos_min=cos[0], cos_p05=cos[len(cos) // 20][0],
                         cos_med=cos[len(cos) // 2][0], norm_ratio=(nr[0], nr[len(nr) // 2], nr[-1]))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--hw", type=int, default=96)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = dict(C=3, B=a.batch, hw=(a.hw, a.hw), lds=[False, False, False])
    from cavp_amd.synth import synth_state_dict
    m0, sd0 = build(None, cfg, torch.float32, dev)
    variants = [("synthetic (untrained)", {k: v.cpu().clone() for k, v in sd0.items()}, None)]
    for steps, damp in ((a.steps, 1.0), (2 * a.steps, 1.0), (a.steps, 0.2)):
        sd, losses = condition(cfg, steps, a.lr, dev, damp=damp)
        variants.append((f"{steps} f32 steps at lr {a.lr}, bn3 gamma x{damp}", sd, losses))
    for name, sd, losses in variants:
        r = compare(sd, cfg, dev)
        if losses:
            print(f"== {name}: training loss {losses[0]:.4f} -> {losses[-1]:.4f}")
        else:
            print(f"== {name}")
        for dt, v in r.items():
            print(f"   {dt}: oracle logits min/max/std {v['logit_range'][0]:.2f}/{v['logit_range'][1]:.2f}/{v['logit_range'][2]:.2f}  logits rel L2 {v['logits_rel']:.3e}  loss {v['loss']:.5f} (oracle {v['oracle_loss']:.5f})  whole-gradient cosine {v['whole_cos']:.5f}  per-parameter cosine min {v['cos_min'][0]:.4f} "
                  f"({v['cos_min'][1]}) p05 {v['cos_p05']:.4f} median {v['cos_med']:.4f}  norm ratio min/med/max {v['norm_ratio'][0]:.3f}/{v['norm_ratio'][1]:.3f}/{v['norm_ratio'][2]:.3f}")


if __name__ == "__main__":
    main()
