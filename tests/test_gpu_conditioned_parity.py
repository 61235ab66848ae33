"""End-to-end parity of the bf16 training step - the dtype the headline frames/s is measured in - against the CPU ORACLE
(f32, pinned to the reference by tests/golden) on CONDITIONED weights.

On the synthetic random weights the batch-statistics BatchNorm trunk amplifies every perturbation layer by layer (bf16 storage
rounding included: the un-forced bf16 logits sit 60 % away from the oracle's, tools/conditioned_probe.py), so the model-level
bf16 tests on those weights can only pin loss and gradient-norm statistics.  Here the synthetic set is first trained for 100
f32 steps on a learnable synthetic task with the repo's own fused step + fused SGD / Adam (cavp_amd.optim.FusedSGDAdam), then
frozen; on these weights the bf16 step is held to the oracle's logits, loss and gradient direction.  Measured (MI355X, round
3): logits 3e-3 .. 5e-3 relative L2 (the f32 path: 5e-7), loss within 1e-4 .. 9e-4, whole-gradient cosine 0.979, per-parameter
cosine median 0.958 / 5th percentile 0.89 (f32 path: >= 0.9999): at a trained point the gradient is a small residual of
near-cancelling terms, so bf16 storage rounding shows in its direction before it shows in the logits."""
import types

import pytest
import torch

from cavp_amd.synth import learnable_inputs, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = dict(C=3, B=8, hw=(96, 96), lds=[False, False, False])


def _build(sd, dtype):
    from cavp_amd.cavp_model import CAVP
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=CFG["lds"], audio_backbone="vgg",
                                 num_classes=CFG["C"], batch_size=CFG["B"], local_rank="cpu")
    m = CAVP(50, None, num_classes=CFG["C"], args=args)
    if sd is None:
        sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    m.train().to(DEV).set_compute_dtype(dtype)
    return m


@pytest.fixture(scope="module")
def conditioned():
    """(state_dict on the CPU, first loss, last loss) after 100 f32 training steps over four learnable batches."""
    from cavp_amd import _lib
    from cavp_amd.optim import FusedSGDAdam
    m = _build(None, torch.float32)
    batches = [[t.to(DEV) for t in learnable_inputs(CFG["B"], CFG["hw"], CFG["C"], seed=3 + i)] for i in range(4)]
    opt, losses = None, []
    # deterministic mode (fixed-order reductions) for the training run: the weights the tests below are measured on are then the
    # same on every run - with the default f32 atomics 100 steps of rounding noise moved the bf16 figures by +-0.02 from run to run
    _lib.set_deterministic(True, torch.device(DEV))
    try:
        for it in range(100):
            image, audio, label = batches[it % 4]
            loss = m.train_step(image, audio, label)
            if opt is None:
                opt = FusedSGDAdam(m, m._grad_arena, 1e-2, momentum=0.9, weight_decay=1e-4)
            opt.step(1e-2)
            losses.append(float(loss.item()))
        torch.cuda.synchronize()
    finally:
        _lib.set_deterministic(False)
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, losses[0], losses[-1]


@pytest.fixture(scope="module")
def oracle_step(conditioned):
    from oracle import cavp_oracle as O
    sd = conditioned[0]
    image, audio, label = learnable_inputs(CFG["B"], CFG["hw"], CFG["C"], seed=11)   # a batch the training run never saw
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    out, _, _ = O.cavp_forward(sd2, image, audio, CFG["lds"], eval_mode=False)
    loss = O.ce_loss_train(out, label, CFG["B"])
    loss.backward()
    return (image, audio, label), out.detach(), float(loss.item()), {k: p.grad for k, p in params.items() if p.grad is not None}


def _compare(sd, dtype, oracle_step):
    (image, audio, label), ref, ref_loss, ref_g = oracle_step
    m = _build(sd, dtype)
    loss = m.train_step(image.to(DEV), audio.to(DEV), label.to(DEV), want_pred=True)
    torch.cuda.synchronize()
    pred = m._last_outputs[0].float().cpu()
    rel = float((pred - ref).norm() / ref.norm())
    dot = na = nb = 0.0
    cos = []
    big = max(float(g.norm()) for g in ref_g.values())
    for k, p in m.named_parameters():
        if p.grad is None or k not in ref_g:
            continue
        a, b = p.grad.double().cpu().flatten(), ref_g[k].double().flatten()
        dot, na, nb = dot + float(a @ b), na + float(a @ a), nb + float(b @ b)
        if float(b.norm()) >= 1e-3 * big:   # (directions of gradients that are ~0 in the oracle are noise on both sides)
            cos.append(float((a @ b) / (a.norm() * b.norm())))
    cos.sort()
    return dict(logits_rel=rel, loss=float(loss.item()), ref_loss=ref_loss, whole_cos=dot / (na * nb) ** 0.5,
                cos_med=cos[len(cos) // 2], cos_p05=cos[len(cos) // 20], n=len(cos), logit_std=float(ref.std()))


def test_training_conditions_the_weights(conditioned):
    _, first, last = conditioned
    assert last < 0.5 * first, (first, last)   # measured 2.39 -> ~0.2: the task is learnable and the fused step + optimiser learn it


def test_f32_step_vs_oracle_on_conditioned_weights(conditioned, oracle_step):
    r = _compare(conditioned[0], torch.float32, oracle_step)
    print("f32 :", r)
    assert r["logit_std"] > 1.0                                   # a real prediction, not a collapsed constant
    assert r["logits_rel"] <= 1e-4 and abs(r["loss"] - r["ref_loss"]) <= 1e-4 * max(1.0, r["ref_loss"])
    assert r["whole_cos"] >= 0.9999 and r["cos_p05"] >= 0.999


def test_bf16_step_vs_oracle_on_conditioned_weights(conditioned, oracle_step):
    r = _compare(conditioned[0], torch.bfloat16, oracle_step)
    print("bf16:", r)
    assert r["logits_rel"] <= 2e-2, r                             # measured 3e-3 .. 5e-3 (VERDICT r02 target: <= 2 %)
    assert abs(r["loss"] - r["ref_loss"]) <= 5e-3 * max(1.0, r["ref_loss"]), r
    # The figures follow the conditioned WEIGHTS, which 100 f32 training steps with the tree's own kernels produce: any change of an f32
    # summation order gives this test another instance.  profiles/r06_conditioned_cross.txt (2 x 2 cross of weights x kernels, rounds
    # 5 / 6): whole-gradient cosine 0.982 on one weight set and 0.950 on the other with EITHER kernel build (the builds agree to <= 3e-3
    # on the same weights), median 0.966 / 0.958, p05 0.91 / 0.90.  (Round 3 .. 5 instance: 0.979 / 0.958 / 0.891, bound 0.96.)
    assert r["whole_cos"] >= 0.93, r
    assert r["cos_med"] >= 0.93 and r["cos_p05"] >= 0.80, r


@pytest.fixture(scope="module")
def oracle_step_b32(conditioned):
    """ONE oracle training step (forward + autograd backward, f32, CPU) at the benchmarked shape - B = 32 images + 64 audio clips,
    224 x 224 - on the conditioned weights: ~10 .. 50 s of CPU time and ~13 GB, paid once per suite.  Round-5 review: the benchmarked
    dtype's gradients at the benchmarked shape must be held to the ORACLE, not to another HIP path."""
    from oracle import cavp_oracle as O
    B, hw = 32, (224, 224)
    sd = conditioned[0]
    image, audio, label = learnable_inputs(B, hw, CFG["C"], seed=17)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    out, _, _ = O.cavp_forward(sd2, image, audio, CFG["lds"], eval_mode=False)
    loss = O.ce_loss_train(out, label, B)
    loss.backward()
    grads = {k: p.grad.detach().double().flatten() for k, p in params.items() if p.grad is not None}
    return (image, audio, label), out.detach(), float(loss.item()), grads


def _grad_stats(g_ref, g):
    keys = [k for k in g_ref if k in g]
    dot = sum(float(g_ref[k] @ g[k]) for k in keys)
    na, nb = sum(float(g_ref[k] @ g_ref[k]) for k in keys), sum(float(g[k] @ g[k]) for k in keys)
    cos = sorted(float((g_ref[k] @ g[k]) / (g_ref[k].norm() * g[k].norm())) for k in keys
                 if float(g_ref[k].norm()) > 1e-12 * na ** 0.5 and g_ref[k].numel() >= 16 and float(g[k].norm()) > 0)
    return dict(whole=dot / (na * nb) ** 0.5, ratio=(nb / na) ** 0.5, med=cos[len(cos) // 2], p05=cos[len(cos) // 20], n=len(cos), keys=len(keys))


def _hip_step_b32(sd, dtype, batch):
    image, audio, label = [t.to(DEV) for t in batch]
    m = _build(sd, dtype)
    loss = m.train_step(image, audio, label, want_pred=True)
    torch.cuda.synchronize()
    pred = m._last_outputs[0].float().cpu()
    g = {k: p.grad.detach().double().flatten().cpu() for k, p in m.named_parameters() if p.grad is not None}
    return float(loss.item()), pred, g


def test_bf16_forward_at_the_bench_shape_vs_oracle(conditioned, oracle_step_b32):
    """The benchmarked shape itself (B = 32 images + 64 audio clips, 224 x 224, bf16): train-mode forward logits and the CE loss of
    the conditioned weights against the oracle's f32 forward."""
    batch, ref, ref_loss, _ = oracle_step_b32
    loss, pred, _ = _hip_step_b32(conditioned[0], torch.bfloat16, batch)
    rel = float((pred - ref).norm() / ref.norm())
    print(f"bf16 @ B=32 224x224: logits rel L2 {rel:.3e} (std {float(ref.std()):.2f}), loss {loss:.5f} vs oracle {ref_loss:.5f}")
    assert float(ref.std()) > 1.0
    assert rel <= 2e-2, rel
    assert abs(loss - ref_loss) <= 5e-3 * max(1.0, ref_loss)


def test_gradients_at_the_bench_shape_vs_oracle(conditioned, oracle_step_b32):
    """The backward at the benchmarked shape (B = 32, 224 x 224) against the ORACLE's autograd step, for the benchmarked dtype (bf16)
    and for the f32 parity path.  No HIP kernel on the reference side: every parameter gradient of the hot path (backbone, ASPP,
    audio encoder, cross-modal attention, decoder head) is compared with oracle/cavp_oracle.py, which tests/test_oracle_golden.py
    pins to the reference's own outputs and gradients.  Bars: f32 as in the 96 x 96 case (whole-gradient cosine >= 0.9999); bf16 the
    statistics measured on MI355X for this fixture (round 5, against the f32 HIP step: whole 0.994, median 0.968, p05 0.863)."""
    batch, ref, ref_loss, g_ref = oracle_step_b32
    sd = conditioned[0]
    l32, p32, g32 = _hip_step_b32(sd, torch.float32, batch)
    r32 = _grad_stats(g_ref, g32)
    print(f"f32  vs ORACLE @ B=32 224x224: loss {l32:.6f} vs {ref_loss:.6f}; logits rel {float((p32 - ref).norm() / ref.norm()):.2e}; {r32}")
    assert r32["keys"] == len(g_ref) == len(g32), (r32["keys"], len(g_ref), len(g32))   # the same parameter set has gradients on both sides
    assert abs(l32 - ref_loss) <= 1e-4 * max(1.0, ref_loss)
    assert float((p32 - ref).norm() / ref.norm()) <= 1e-4
    assert r32["whole"] >= 0.9999 and r32["p05"] >= 0.999 and 0.999 <= r32["ratio"] <= 1.001, r32
    l16, _, g16 = _hip_step_b32(sd, torch.bfloat16, batch)
    r16 = _grad_stats(g_ref, g16)
    print(f"bf16 vs ORACLE @ B=32 224x224: loss {l16:.5f} vs {ref_loss:.5f}; {r16}")
    assert r16["keys"] == len(g_ref)
    assert abs(l16 - ref_loss) <= 5e-3 * max(1.0, ref_loss)
    assert r16["whole"] >= 0.985 and 0.97 <= r16["ratio"] <= 1.03, r16
    assert r16["med"] >= 0.95 and r16["p05"] >= 0.80, r16


def test_bf16_gradients_at_the_bench_shape_vs_f32_step(conditioned):
    """HIP vs HIP consistency screen (NOT the parity anchor - that is test_gradients_at_the_bench_shape_vs_oracle above): the bf16
    step's gradients against the f32 HIP step's on the conditioned weights at the benchmarked shape.
    Same statistics as the 96 x 96 case against the oracle; the larger batch averages the rounding noise down (measured
    whole-gradient cosine 0.994 here against 0.973 there)."""
    B, hw = 32, (224, 224)
    sd = conditioned[0]
    image, audio, label = [t.to(DEV) for t in learnable_inputs(B, hw, CFG["C"], seed=17)]
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        m = _build(sd, dt)
        loss = m.train_step(image, audio, label)
        torch.cuda.synchronize()
        res[dt] = (float(loss.item()), {k: p.grad.detach().double().flatten().cpu() for k, p in m.named_parameters() if p.grad is not None})
        del m
    (l32, g32), (l16, g16) = res[torch.float32], res[torch.bfloat16]
    assert set(g32) == set(g16)
    dot = sum(float(g32[k] @ g16[k]) for k in g32)
    na, nb = sum(float(g32[k] @ g32[k]) for k in g32), sum(float(g16[k] @ g16[k]) for k in g16)
    cos = sorted(float((g32[k] @ g16[k]) / (g32[k].norm() * g16[k].norm())) for k in g32
                 if float(g32[k].norm()) > 1e-12 * na ** 0.5 and g32[k].numel() >= 16)
    whole, med, p05 = dot / (na * nb) ** 0.5, cos[len(cos) // 2], cos[len(cos) // 20]
    print(f"bf16 vs f32 @ B=32 224x224: loss {l16:.5f} vs {l32:.5f}; whole-gradient cosine {whole:.4f}, norm ratio {(nb / na) ** 0.5:.4f}, "
          f"per-parameter cosine median {med:.4f} / p05 {p05:.4f} over {len(cos)} tensors")
    assert abs(l16 - l32) <= 5e-3 * max(1.0, l32)
    # measured on the deterministic fixture: 0.9941 / 1.0075, 0.968 / 0.863 (on three differently-conditioned weight sets from
    # non-deterministic fixture runs: whole 0.9970 .. 0.9997, median 0.979 .. 0.983, p05 0.897 .. 0.911)
    assert whole >= 0.985 and 0.97 <= (nb / na) ** 0.5 <= 1.03
    assert med >= 0.95 and p05 >= 0.80
