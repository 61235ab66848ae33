"""Optimiser harness: parameter grouping / schedule against the fixture generated from the reference's own
group_weight + WarmUpPolyLR (tools/make_golden.py::run_optstep), the fused HIP optimiser against torch.optim, and two full
train + update steps against the reference's updated sentinel weights."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cavp_amd.synth import synth_state_dict  # noqa: E402

DEV = "cuda:0"
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build_model(num_classes, lds, batch):
    from cavp_amd.cavp_model import CAVP
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=lds, audio_backbone="vgg",
                                 num_classes=num_classes, batch_size=batch, local_rank="cpu")
    return CAVP(50, None, num_classes=num_classes, args=args)


def load_synth_weights(m, seed=1):
    m.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed), strict=True)


def _golden():
    with open(os.path.join(GOLDEN_DIR, "optim_groups.json")) as f:
        groups = json.load(f)
    return groups, np.load(os.path.join(GOLDEN_DIR, "optstep.npz"), allow_pickle=True)


def test_param_groups_and_schedule_match_reference():
    from cavp_amd.optim import set_group_lr, warmup_poly_lr
    ref, z = _golden()
    m = build_model(num_classes=int(z["cfg/CBHW"][0]), lds=[False, False, False], batch=int(z["cfg/CBHW"][1]))
    names = {id(p): k for k, p in m.named_parameters()}
    groups = set_group_lr(m, 1.0)
    assert [[names[id(p)] for p in g["params"]] for g in groups] == ref["groups"]
    assert [g["lr"] for g in groups] == ref["lr_mult"]
    hyp_lr, power, mom, wd = [float(v) for v in z["cfg/hyp"]]
    assert [g.get("weight_decay", wd) for g in groups] == ref["weight_decay"]
    assert [k for k, _ in m.audio_backbone.named_parameters()] == ref["audio"]
    covered = {id(p) for g in groups for p in g["params"]} | {id(p) for p in m.audio_backbone.parameters()}
    assert covered == {id(p) for p in m.parameters()}          # the two optimisers partition the model
    sched = warmup_poly_lr(hyp_lr, power, int(z["cfg/total_iters"][0]), 0)
    np.testing.assert_allclose([sched(i) for i in range(len(z["lr"]))], z["lr"], rtol=1e-12)
    w = warmup_poly_lr(1e-2, 0.9, 100, warmup_steps=10)
    assert w(0) == 0.0 and abs(w(5) - 5e-3) < 1e-15 and w(99) >= 1e-8 and w(10) == pytest.approx(1e-2 * 0.9 ** 0.9)


@pytest.mark.gpu
def test_fused_optimizer_vs_torch_optim():
    from cavp_amd.optim import FusedSGDAdam, set_group_lr
    from cavp_amd.train import GradArena
    m = build_model(num_classes=3, lds=[False, False, False], batch=2)
    load_synth_weights(m, seed=1)
    m = m.to(DEV)
    arena = GradArena(list(m.parameters()), DEV)
    ref = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in m.named_parameters()}
    names = {id(p): k for k, p in m.named_parameters()}
    lr0, mom, wd = 1e-2, 0.9, 1e-3
    groups = [dict({kk: vv for kk, vv in g.items() if kk != "params"}, params=[ref[names[id(p)]] for p in g["params"]],
                   lr=g["lr"] * lr0) for g in set_group_lr(m, 1.0)]
    opt_v = torch.optim.SGD(groups, lr=lr0, momentum=mom, weight_decay=wd)
    opt_a = torch.optim.Adam([ref["audio_backbone." + k] for k, _ in m.audio_backbone.named_parameters()], lr=lr0)
    fused = FusedSGDAdam(m, arena, lr0, momentum=mom, weight_decay=wd)
    gen = torch.Generator().manual_seed(5)
    for it, lr in enumerate([1e-2, 7e-3, 3e-3]):
        g_flat = torch.randn(arena.flat.numel(), generator=gen) * 0.1
        arena.flat.copy_(g_flat.to(DEV))
        never = {id(p) for p in m.params_without_grad()}   # pos_embed_*, cls_head: torch leaves .grad None and skips them
        for k, p in m.named_parameters():
            ref[k].grad = None if id(p) in never else arena.views[id(p)].detach().cpu().clone()
        for i, g in enumerate(opt_v.param_groups):
            g["lr"] = lr * (1.0 if i < 4 else 10.0)
        opt_v.step()
        opt_a.step()
        fused.step(lr)
        torch.cuda.synchronize()
        worst = 0.0
        for k, p in m.named_parameters():
            d = float((p.detach().cpu() - ref[k].detach()).abs().max())
            worst = max(worst, d / (1e-6 + float(ref[k].detach().abs().max())))
        assert worst <= 2e-6, (it, worst)


@pytest.mark.gpu
def test_two_train_steps_vs_reference_weights(deterministic):
    from cavp_amd.optim import FusedSGDAdam, warmup_poly_lr
    from cavp_amd.synth import synth_inputs
    ref, z = _golden()
    C, B, H, W = [int(v) for v in z["cfg/CBHW"]]
    hyp_lr, power, mom, wd = [float(v) for v in z["cfg/hyp"]]
    m = build_model(num_classes=C, lds=[False, False, False], batch=B)
    load_synth_weights(m, seed=1)
    m = m.train().to(DEV)
    image, audio, label = synth_inputs(B, (H, W), audio_batch=2 * B, num_classes=C, seed=4)
    image, audio, label = image.to(DEV), audio.to(DEV), label.to(DEV)
    sched = warmup_poly_lr(hyp_lr, power, int(z["cfg/total_iters"][0]), 0)
    params = dict(m.named_parameters())
    sent = [k[len("w0/"):] for k in z.files if k.startswith("w0/")]

    def samp(t):
        t = t.detach().float().cpu().flatten()
        stride = max(1, t.numel() // 4096)
        return t[::stride][:4096].numpy()

    for k in sent:
        np.testing.assert_allclose(samp(params[k]), z["w0/" + k], rtol=0, atol=1e-7)
    opt = None
    prev = {k: z["w0/" + k].astype(np.float64) for k in sent}
    mine = {k: z["w0/" + k].astype(np.float64) for k in sent}
    for it in range(len(z["loss"])):
        loss = m.train_step(image, audio, label)
        if opt is None:
            opt = FusedSGDAdam(m, m._grad_arena, hyp_lr, momentum=mom, weight_decay=wd)
        opt.step(sched(it))
        torch.cuda.synchronize()
        got = float(loss.item())
        # step 1 runs on weights that already differ (Adam's first step is +-lr * sign(g): a gradient element that rounds
        # to the other side of zero moves a weight by 2 lr), and B = 4 batch-statistics BN amplifies it: 1 run in ~8 lands
        # 6-7 % away from the reference's second loss, so only step 0 is held tight
        assert abs(got - float(z["loss"][it])) <= (1e-4 if it == 0 else 0.15) * abs(float(z["loss"][it])), (it, got)
        rep = []
        for k in sent:
            cur = samp(params[k]).astype(np.float64)
            want = z[f"w{it + 1}/" + k].astype(np.float64)
            d_got, d_ref = cur - mine[k], want - prev[k]     # each side's own update of this step
            prev[k], mine[k] = want, cur
            n_ref = np.linalg.norm(d_ref)
            if n_ref < 1e-12:
                continue
            cos = float(d_got @ d_ref / (np.linalg.norm(d_got) * n_ref + 1e-30))
            ratio = float(np.linalg.norm(d_got) / n_ref)
            rep.append((k, cos, ratio))
            adam = k.startswith("audio_backbone.")
            if it == 0:
                # first step: grouping, lr multipliers, weight decay, SGD first-step buffer and Adam's bias-corrected
                # sign step, straight against the reference (the gradient itself is pinned to ~1e-3 by c1p_train)
                assert cos >= (0.93 if adam else 0.985), (it, k, cos, ratio)
                assert abs(ratio - 1.0) <= 0.05, (it, k, cos, ratio)
            else:
                # second step: the gradient is re-evaluated at weights that already differ in the last bits and by
                # +-2 lr wherever Adam's sign step met a near-zero gradient; with batch-statistics BN at B = 4 that is
                # enough to move the gradients of the early layers (the momentum / bias-correction / schedule arithmetic
                # itself is pinned exactly by test_fused_optimizer_vs_torch_optim).  The test runs in deterministic mode
                # (fixed-order reductions), so the figures are reproducible run to run - measured: worst sentinel cosine
                # 0.663, norm ratios 0.93 .. 1.13 (round 2, with the default f32 atomics, saw 0.44 .. 0.99 across runs and
                # could only keep a 0.3 band)
                assert cos >= 0.6 and 0.85 <= ratio <= 1.2, (it, k, cos, ratio)
        print(f"step {it}: loss {got:.5f} (reference {float(z['loss'][it]):.5f}); weight-update cosine min "
              f"{min(r[1] for r in rep):.4f}, norm ratio in [{min(r[2] for r in rep):.3f}, {max(r[2] for r in rep):.3f}]")
