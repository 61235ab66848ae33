"""End-to-end training-step parity (forward with batch-stat BN + full backward through the HIP path) against the
golden vectors captured from the reference's own autograd (tests/golden/c1p_train.npz)."""
import types

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cavp_amd.synth import synth_inputs, synth_state_dict
from tests._golden_util import check_tap, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(cfg, dtype=torch.float32):
    from cavp_amd.cavp_model import CAVP
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=cfg["lds"], audio_backbone="vgg",
                                 num_classes=cfg["C"], batch_size=cfg["B"], local_rank="cpu")
    m = CAVP(50, None, num_classes=cfg["C"], args=args)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    m.train().to(DEV).set_compute_dtype(dtype)
    return m, sd


def _step(m, cfg, use_hip_ce):
    B = cfg["B"]
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=0)
    out, fus, pack = m(image.to(DEV), audio.to(DEV), None, False)      # trainer_cavp_vpo_mono.py:168
    if use_hip_ce:
        from cavp_amd import train_ops as T
        loss, dl = T.ce_loss(out.detach(), label.to(DEV), B)
        out.backward(dl)
        loss = float(loss.item())
    else:
        output = out[:B] + out[B:] * 0.0                               # :171
        loss_t = F.cross_entropy(output, label.to(DEV), ignore_index=255)   # loss/losser.py:60-62
        loss_t.backward()
        loss = float(loss_t.item())
    torch.cuda.synchronize()
    return out, fus, pack, loss


@pytest.mark.parametrize("case", ["c1p_train", "c1_train"])
@pytest.mark.parametrize("use_hip_ce", [False, True], ids=["torch_ce", "hip_ce"])
def test_train_step_matches_reference_f32(use_hip_ce, case):
    """c1p_train: the native 224 / OS16 / 2-class model; c1_train: config #1's model (OS8: layer3 / layer4 at 28 x 28 with
    dilation, 22 classes) - forward, CE loss and every parameter gradient against the reference's own autograd."""
    z, cfg = load_case(case)
    m, _ = _build(cfg)
    out, fus, pack, loss = _step(m, cfg, use_hip_ce)
    assert abs(loss - float(z["loss"][0])) <= 2e-5 * max(1.0, abs(float(z["loss"][0]))), (loss, float(z["loss"][0]))
    for k, t in dict(out_pred=out, out_fusion=fus, pack_visual=pack["visual"], pack_audio=pack["audio"],
                     pack_attn_v=pack["attn_v"]).items():
        scale = max(1.0, float(np.abs(z["sample/" + k]).max()))
        # batch-statistics BN at B=2 amplifies rounding: the reference's own f32 forward is 3.8e-3 (4e-4 relative) away
        # from an f64 evaluation of the same graph on these inputs (measured with the oracle), so two f32
        # implementations can only be expected to agree to ~1e-3 relative here (eval-mode parity bar stays 1e-3 abs).
        check_tap(z, k, t.detach(), 1.5e-3 * scale, what="train:")
    params = dict(m.named_parameters())
    keys, vals = list(z["grad_norm_keys"]), z["grad_norm_vals"]
    worst = (0.0, None)
    for k, v in zip(keys, vals):
        g = params[k].grad
        assert g is not None, f"no gradient for {k}"
        n = float(g.double().norm().item())
        rel = abs(n - v) / max(v, 1e-6)
        if rel > worst[0]:
            worst = (rel, k)
        # tolerance: the reference's own f32 gradient norms differ from an f64 evaluation by up to 1.1e-2 relative
        # (median 4e-3) on this B=2 batch-stat-BN step (measured with the oracle); 1.5e-2 is that noise floor.
        assert rel <= 1.5e-2 or abs(n - v) <= 1e-6, f"{k}: |grad| {n:.6g} vs reference {v:.6g} (rel {rel:.2e})"
    for k, p in params.items():
        if k not in keys:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{k} must not receive gradient"
    for s in [s for s in z.files if s.startswith("grad_sample/")]:
        k = s[len("grad_sample/"):]
        g = params[k].grad.detach().float().cpu().contiguous().flatten()
        ref = z[s]
        smp = g[:: max(1, g.numel() // 4096)][:4096].numpy()
        # elementwise noise floor of the reference itself (f32 vs f64, same measurement): 2.5-5.1e-2 of max|grad| for
        # backbone / ASPP / reduce tensors (the ASPP pooled-branch BN normalises over M = B = 2 samples, where dz is
        # analytically 0 and numerically rounding noise x rstd), <= 1e-2 elsewhere.
        noisy = k.startswith(("backbone.", "segment.aspp", "segment.reduce"))
        tol = (1.5e-1 if noisy else 2e-2) * max(1e-6, float(np.abs(ref).max()))  # 3x the measured floor; the tight bar is the B=8 test
        assert np.abs(smp - ref).max() <= tol, f"{k}: sampled grad err {np.abs(smp - ref).max():.3e} > {tol:.3e}"
    print("train step ok: loss", loss, "worst grad-norm rel err", worst)


def test_train_step_b8_matches_reference_f32(deterministic):
    """The operative element-wise gradient check on a well-conditioned batch (golden c1p_train_b8: 8 images + 16 audio clips,
    no BatchNorm over 2 samples), against the REFERENCE's own autograd and against the exact value of the same graph.

    Two f32 evaluations of this graph cannot agree arbitrarily well: the reference's own f32 gradients are 1.3e-2 .. 3.0e-2
    (relative L2 over the sample) away from the float64 value on the backbone / ASPP tensors and ~2e-3 elsewhere, its gradient
    norms up to 7e-3 (c1p_train_b8_f64.npz: the pinned oracle evaluated in float64, tools/make_golden.py::run_f64_arbiter).  So:
      (1) HIP f32 vs the exact value is held to 1.5 x the REFERENCE's f32 error against that value (+ 1e-3) on every sampled
          sentinel tensor, and the error distribution of all gradient norms to 1.5 x the reference's: this path is as accurate
          as the reference's own arithmetic;
      (2) HIP f32 vs reference f32 directly: 5e-3 / cosine 0.9995 where the reference itself is that close to the exact value
          (attention, projector, audio encoder, head); on the noisy tensors the group's worst and mean error vs the exact value
          stay within 1.25 x the reference's, each tensor within 4e-2 / cosine 0.999 of the reference.
    Measured (round 4): HIP vs exact 0.6e-2 .. 2.7e-2 on the noisy tensors where the reference's f32 run is 1.3e-2 .. 3.0e-2."""
    z, cfg = load_case("c1p_train_b8")
    z64 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1p_train_b8_f64.npz"), allow_pickle=True)
    assert cfg["B"] == 8
    m, _ = _build(cfg)
    out, fus, pack, loss = _step(m, cfg, use_hip_ce=False)
    assert abs(loss - float(z["loss"][0])) <= 2e-5 * max(1.0, abs(float(z["loss"][0]))), (loss, float(z["loss"][0]))
    assert abs(loss - float(z64["loss"][0])) <= 5e-5   # (the reference's f32 loss is 7.5e-6 from the exact value, this path 3.3e-5)
    for k, t in dict(out_pred=out, out_fusion=fus, pack_visual=pack["visual"], pack_audio=pack["audio"]).items():
        scale = max(1.0, float(np.abs(z["sample/" + k]).max()))
        check_tap(z, k, t.detach(), 1e-3 * scale, what="train b8:")
    params = dict(m.named_parameters())
    assert list(z["grad_norm_keys"]) == list(z64["grad_norm_keys"])
    worst_n = (0.0, None)
    en_hip, en_ref = [], []
    for k, v, v64 in zip(list(z["grad_norm_keys"]), z["grad_norm_vals"], z64["grad_norm_vals"]):
        g = params[k].grad
        assert g is not None, k
        n = float(g.double().norm().item())
        if v64 < 1e-6:
            continue
        en_hip.append(abs(n - v64) / v64)
        en_ref.append(abs(v - v64) / v64)
        worst_n = max(worst_n, (en_hip[-1], k))
        assert abs(n - v) / v <= 1e-2, f"{k}: |grad| {n:.6g} vs reference {v:.6g}"
    # the per-tensor errors of the norms are quasi-random (either run may be lucky on a tensor): their distributions are compared
    en_hip, en_ref = np.array(en_hip), np.array(en_ref)
    print(f"b8 gradient norms vs exact: HIP median {np.median(en_hip):.2e} p90 {np.quantile(en_hip, 0.9):.2e} max {en_hip.max():.2e} | "
          f"reference f32 median {np.median(en_ref):.2e} p90 {np.quantile(en_ref, 0.9):.2e} max {en_ref.max():.2e}")
    assert np.median(en_hip) <= 1.5 * np.median(en_ref) + 2e-4 and np.quantile(en_hip, 0.9) <= 1.5 * np.quantile(en_ref, 0.9) + 2e-4
    assert en_hip.max() <= 1.5 * en_ref.max()
    rows = []
    for s in [s for s in z.files if s.startswith("grad_sample/")]:
        k = s[len("grad_sample/"):]
        g = params[k].grad.detach().float().cpu().contiguous().flatten()
        ref, exact = z[s].astype(np.float64), z64[s]
        smp = g[:: max(1, g.numel() // 4096)][:4096].numpy().astype(np.float64)

        def rel_cos(a, b):
            return (float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)),
                    float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30)))
        (e_hip, c_hip), (e_ref, c_ref), (e_dir, c_dir) = rel_cos(smp, exact), rel_cos(ref, exact), rel_cos(smp, ref)
        rows.append((k, e_hip, e_ref, e_dir, c_dir))
    print(f"b8 train step: loss {loss:.6f}; worst gradient norm vs exact {worst_n[0]:.2e} ({worst_n[1]})")
    for k, e_hip, e_ref, e_dir, c_dir in rows:
        print(f"  {k:48s} HIP vs exact {e_hip:.2e} | reference vs exact {e_ref:.2e} | HIP vs reference {e_dir:.2e} cos {c_dir:.6f}")
    noisy = [r for r in rows if r[2] > 3.5e-3]      # backbone / ASPP / reduce: the reference's own f32 run is 0.6e-2 .. 3e-2 off
    quiet = [r for r in rows if r[2] <= 3.5e-3]
    assert len(noisy) >= 5 and len(quiet) >= 9
    for k, e_hip, e_ref, e_dir, c_dir in quiet:
        assert e_hip <= 1.5 * e_ref + 1e-3, f"{k}: vs exact {e_hip:.3e}, the reference's f32 run {e_ref:.3e}"
        assert e_dir <= 5e-3 and c_dir >= 0.9995, f"{k}: vs reference rel L2 {e_dir:.3e} cosine {c_dir:.6f}"
    # the noisy tensors' errors are rounding noise amplified by ~50 batch-statistics BatchNorm layers: quasi-random per tensor
    # (either run is the closer one on some of them), so the group is compared: no worse than the reference's f32 run overall
    assert max(r[1] for r in noisy) <= 1.25 * max(r[2] for r in noisy), [(r[0], r[1], r[2]) for r in noisy]
    assert np.mean([r[1] for r in noisy]) <= 1.25 * np.mean([r[2] for r in noisy]), [(r[0], r[1], r[2]) for r in noisy]
    for k, e_hip, e_ref, e_dir, c_dir in noisy:
        assert e_hip <= 3.0 * e_ref and e_dir <= 4e-2 and c_dir >= 0.999, f"{k}: vs exact {e_hip:.3e} (reference {e_ref:.3e}), vs reference {e_dir:.3e} cosine {c_dir:.6f}"


def test_running_stats_and_second_step():
    """BN running statistics follow nn.BatchNorm2d semantics (momentum 0.1, unbiased var) and a second step works."""
    from oracle import cavp_oracle as O
    z, cfg = load_case("c1p_train")
    m, sd = _build(cfg)
    bn = m.backbone.backbone.conv1[1]
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    _step(m, cfg, True)
    image, _, _ = synth_inputs(cfg["B"], cfg["hw"], audio_batch=2 * cfg["B"], num_classes=cfg["C"], seed=0)
    zc = F.conv2d(image, sd["backbone.backbone.conv1.0.weight"], None, 2, 1)
    mean, var = zc.mean((0, 2, 3)), zc.var((0, 2, 3), unbiased=True)
    assert float((bn.running_mean.cpu() - (0.9 * rm0.cpu() + 0.1 * mean)).abs().max()) <= 1e-5
    assert float((bn.running_var.cpu() - (0.9 * rv0.cpu() + 0.1 * var)).abs().max()) <= 1e-4
    assert int(bn.num_batches_tracked.item()) == 1
    g1 = m.segment.upsample.classifier.weight.grad.clone()
    _step(m, cfg, True)     # gradients accumulate like torch (.grad += new)
    g2 = m.segment.upsample.classifier.weight.grad
    assert float((g2 - g1).abs().max()) > 0


def _oracle_grads(sd, cfg, image, audio, label):
    from oracle import cavp_oracle as O
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    out, fus, _ = O.cavp_forward(sd2, image, audio, cfg["lds"], eval_mode=False)
    loss = O.ce_loss_train(out, label, cfg["B"])
    loss.backward()
    return out.detach(), float(loss.item()), {k: p.grad for k, p in params.items() if p.grad is not None}


def test_frozen_batchnorm_backward_vs_oracle():
    """Fine-tuning recipe: every BatchNorm module in eval() (running statistics, no update) while the rest trains.  Forward
    and the full backward against the CPU oracle's autograd over the same graph (bn_train=False); without batch statistics
    the graph is well conditioned, so the gradient bar is tight.  The running buffers must not move."""
    from oracle import cavp_oracle as O
    cfg = dict(C=3, B=2, hw=(64, 96), lds=[False, False, False])
    m, sd = _build(cfg)
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.eval()
    B = cfg["B"]
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=6)
    from cavp_amd import train_ops as T
    out, fus, pack = m(image.to(DEV), audio.to(DEV), None, False)
    assert out.requires_grad
    loss, dl = T.ce_loss(out.detach(), label.to(DEV), B)
    out.backward(dl)
    torch.cuda.synchronize()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    ro, rf, _ = O.cavp_forward(sd2, image, audio, cfg["lds"], eval_mode=False, bn_train=False)
    rl = O.ce_loss_train(ro, label, B)
    rl.backward()
    assert float((out.detach().cpu() - ro.detach()).abs().max()) <= 1e-3
    assert abs(float(loss.item()) - float(rl.item())) <= 1e-5 * max(1.0, abs(float(rl.item())))
    mine = dict(m.named_parameters())
    worst = (0.0, None)
    for k, p in params.items():
        if p.grad is None:
            continue
        a, b = mine[k].grad.detach().double().cpu().flatten(), p.grad.double().flatten()
        err = float((a - b).norm() / max(float(b.norm()), 1e-6 * float(rl.item())))
        worst = max(worst, (err, k))
        assert err <= 5e-3, (k, err)      # measured worst 2.0e-3 (first audio conv)
    print("frozen-BN backward: worst relative L2 gradient error", worst)
    for k, v in m.state_dict().items():
        if "running_" in k or "num_batches" in k:
            assert torch.equal(v.cpu(), sd[k]), k
    # the native fused step takes the same route
    m.zero_grad(set_to_none=True)
    l2 = float(m.train_step(image.to(DEV), audio.to(DEV), label.to(DEV), all_reduce=False).item())
    assert abs(l2 - float(rl.item())) <= 1e-5 * max(1.0, abs(l2))
    stem = next(k for k in params if k.startswith("backbone.") and k.endswith(".weight") and params[k].grad is not None)
    g = dict(m.named_parameters())[stem].grad.detach().double().cpu().flatten()   # far end of the backward chain
    b = params[stem].grad.double().flatten()
    assert float((g - b).norm() / b.norm()) <= 2e-3, stem


@pytest.mark.parametrize("dtype,norm_tol,cos_tol", [(torch.float32, 5e-3, 0.9995), (torch.bfloat16, 0.25, None)],
                         ids=["f32", "bf16"])
def test_train_step_b8_vs_oracle(dtype, norm_tol, cos_tol):
    """A better-conditioned step (B=8, so no 2-sample BatchNorm) against the CPU oracle's autograd: every parameter's
    gradient by norm and (f32) by direction.  bf16 is held to loss + gradient-norm statistics only: with random
    weights and batch-statistics BN this graph is chaotic under storage rounding - rounding every conv / BN / linear
    output of the *reference* graph to bf16 on the CPU (oracle monkey-patched, same inputs) already moves layer4 to
    cosine 0.85 and the ASPP output to 0.71 of the f32 run, the same figures this path shows (0.86 / 0.72) - so a
    directional bar against f32 would test the weights' conditioning, not the kernels (op-level bf16 parity:
    tests/test_gpu_train_ops.py)."""
    cfg = dict(C=3, B=8, hw=(64, 64), lds=[False, False, False])
    m, sd = _build(cfg, dtype)
    B = cfg["B"]
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=5)
    from cavp_amd import train_ops as T
    out, fus, pack = m(image.to(DEV), audio.to(DEV), None, False)
    loss, dl = T.ce_loss(out.detach(), label.to(DEV), B)
    out.backward(dl)
    torch.cuda.synchronize()
    ref_out, ref_loss, ref_g = _oracle_grads(sd, cfg, image, audio, label)
    assert abs(float(loss.item()) - ref_loss) <= (1e-4 if dtype == torch.float32 else 6e-2) * max(1.0, abs(ref_loss))
    params = dict(m.named_parameters())
    rels, coss = [], []
    for k, g in ref_g.items():
        mine = params[k].grad
        assert mine is not None, k
        a, b = mine.detach().double().cpu().flatten(), g.double().flatten()
        nb = float(b.norm())
        if nb < 1e-8:
            continue
        rels.append(abs(float(a.norm()) - nb) / nb)
        coss.append(float((a @ b) / (a.norm() * b.norm() + 1e-30)))
    rels, coss = np.array(rels), np.array(coss)
    print(f"{dtype}: loss {float(loss.item()):.5f} (oracle {ref_loss:.5f}); grad norm rel err median {np.median(rels):.2e} "
          f"max {rels.max():.2e}; cosine min {coss.min():.5f} median {np.median(coss):.6f}")
    # bf16: the statistic itself moves with any re-association of the same math (three epilogue variants of one
    # kernel gave medians 6.0e-2 / 8.3e-2 / 8.7e-2 on this seed, and the f32 atomics of the column reductions make it vary
    # from run to run as well), hence the loose bar; f32 is the parity test
    assert np.median(rels) <= norm_tol and rels.max() <= 12 * norm_tol
    if cos_tol is not None:
        assert np.median(coss) >= cos_tol and coss.min() >= 1 - 6 * (1 - cos_tol)


def test_train_step_bf16_b2_loss_only():
    """B=2 golden step in bf16: only the forward/loss is meaningful (the 2-sample BatchNorm of the ASPP pooled branch
    turns bf16 rounding into O(1) gradient noise; see test_train_step_b8_vs_oracle for the gradient bar)."""
    z, cfg = load_case("c1p_train")
    m, _ = _build(cfg, torch.bfloat16)
    out, fus, pack, loss = _step(m, cfg, True)
    ref_loss = float(z["loss"][0])
    # 25 runs of this very step gave 1.836 ... 1.895 (reference 1.9026): the f32 atomics of the column reductions re-order,
    # bf16 rounding flips, two-sample BatchNorm amplifies (tools/determinism_probe.py) - hence the wide bar
    assert abs(loss - ref_loss) <= 0.06 * abs(ref_loss) + 0.02, (loss, ref_loss)
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_native_train_step_matches_autograd_path():
    """CAVP.train_step (fused native step, flat gradient arena) == the autograd-node path, gradient by gradient."""
    cfg = dict(C=2, B=2, hw=(64, 64), lds=[False, False, False])
    B = cfg["B"]
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=9)
    m1, _ = _build(cfg)
    m2, _ = _build(cfg)
    from cavp_amd import train_ops as T
    out, _, _ = m1(image.to(DEV), audio.to(DEV), None, False)
    loss1, dl = T.ce_loss(out.detach(), label.to(DEV), B)
    out.backward(dl)
    loss2 = m2.train_step(image.to(DEV), audio.to(DEV), label.to(DEV))
    torch.cuda.synchronize()
    # the statistics / bias reductions use f32 atomics (run-to-run summation order), and this B=2 step amplifies that
    # (see above), so the two paths are compared statistically, not bitwise
    assert abs(float(loss1.item()) - float(loss2.item())) <= 2e-4
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p1.grad is None:
            assert p2.grad is None, k
            continue
        assert p2.grad is not None, k
        a, b = p1.grad.double().flatten(), p2.grad.double().flatten()
        if float(a.norm()) < 1e-10:
            continue
        cos = float((a @ b) / (a.norm() * b.norm()))
        # (one run in ~10 showed cos 0.9977 / a 3.6 % norm difference on the stem conv, the far end of the backward chain)
        assert cos >= 0.99 and abs(float(a.norm()) - float(b.norm())) <= 8e-2 * float(a.norm()), (k, cos)
    for (k, b1), (_, b2) in zip(m1.named_buffers(), m2.named_buffers()):
        assert torch.allclose(b1.float(), b2.float(), atol=1e-3, rtol=1e-3), k


def test_train_step_hipgraph_replay_matches_eager():
    """The whole training step captured as one hipGraph: replays track the eager step (weights re-packed inside the
    graph, running stats and gradients updated by every replay)."""
    cfg = dict(C=2, B=4, hw=(64, 64), lds=[False, False, False])
    B = cfg["B"]
    image, audio, label = [t.to(DEV) for t in synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=11)]
    m1, _ = _build(cfg)
    m2, _ = _build(cfg)
    l1 = m1.train_step(image, audio, label)
    replay = m2.capture_train_step(image, audio, label)   # warm-up + capture advance running stats 3x; reset below
    m2.load_state_dict(m1.state_dict())                   # same running stats as m1 BEFORE its step? -> compare grads only
    l2 = float(replay().item())      # the loss tensor is a static graph buffer: read it before the next replay
    torch.cuda.synchronize()
    assert abs(float(l1.item()) - l2) <= 5e-3
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p1.grad is None:
            assert p2.grad is None, k
            continue
        a, b = p1.grad.double().flatten(), p2.grad.double().flatten()
        if float(a.norm()) < 1e-10:
            continue
        cos = float((a @ b) / (a.norm() * b.norm()))
        assert cos >= 0.99, (k, cos)
    # a parameter update between replays is picked up (packing is inside the graph)
    with torch.no_grad():
        for p in m2.parameters():
            p.mul_(0.5)
    l3 = float(replay().item())
    assert abs(l3 - l2) > 1e-4


def test_split_graph_capture_matches_single_graph():
    """Data-parallel replay = two hipGraphs cut where the early gradient range (head, attention, audio encoder) is final, so
    that its all-reduce overlaps the rest of the backward.  On one GPU the cut must not change anything: same loss and the
    same flat gradient arena as the single-graph replay (up to the f32 atomics of the BN column reductions, whose order
    varies from run to run); the arena puts the late parameters first."""
    cfg = dict(C=2, B=4, hw=(64, 64), lds=[False, False, False])
    B = cfg["B"]
    image, audio, label = [t.to(DEV) for t in synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=12)]
    m1, _ = _build(cfg)
    m2, _ = _build(cfg)
    r1 = m1.capture_train_step(image, audio, label, split=False)
    r2 = m2.capture_train_step(image, audio, label, split=True)
    assert len(m2._train_graph) == 2 and len(m1._train_graph) == 1
    sd = {k: v.clone() for k, v in m1.state_dict().items()}
    for it in range(2):
        m1.load_state_dict(sd)
        m2.load_state_dict(sd)
        l1, l2 = float(r1().item()), float(r2().item())
        torch.cuda.synchronize()
        assert abs(l1 - l2) <= 1e-4 * abs(l1), (it, l1, l2)
        a, b = m1._grad_arena.flat.double(), m2._grad_arena.flat.double()
        cos = float((a @ b) / (a.norm() * b.norm()))
        assert cos >= 0.9995 and abs(float(a.norm() / b.norm()) - 1.0) <= 5e-3, (it, cos)
    ar = m2._grad_arena
    late = m2._late_grad_ids()
    assert 0 < ar.split < ar.flat.numel()
    assert all(id(p) in late for p in ar.params[:len(late)]) and all(id(p) not in late for p in ar.params[len(late):])
    early_bytes = (ar.flat.numel() - ar.split) / ar.flat.numel()
    assert early_bytes > 0.6, early_bytes      # the audio encoder alone is 61 % of the parameters


def test_ce_plus_contrast_through_autograd_path():
    """The reference trainer's loss (trainer_cavp_vpo_mono.py:183-189): CE on `out[:B] + out[B:]*0` PLUS ContrastLoss on
    out_fusion halves - gradients enter the HIP backward through BOTH out_pred and out_fusion.  Checked against the
    CPU oracle's autograd over the same graph (same RNG state for the anchor sampling)."""
    from cavp_amd.contrast import ContrastLoss
    from oracle import cavp_oracle as O
    from oracle.contrast_oracle import contrast_loss
    cfg = dict(C=3, B=4, hw=(64, 64), lds=[False, False, False])
    B = cfg["B"]
    m, sd = _build(cfg)
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=21)
    label[:, 8:40, 8:48] = 1
    label[:, 44:60, 4:60] = 2
    label[:, :4] = 255
    shuf = label.clone()
    shuf[2:] = 0
    crit = ContrastLoss(temperature=0.1, ignore_idx=255, max_views=32)
    out, fus, _ = m(image.to(DEV), audio.to(DEV), None, False)
    torch.manual_seed(77)
    l_ctr = crit(fus[:B], label.to(DEV), fus[B:], shuf.to(DEV))
    l_ce = F.cross_entropy(out[:B] + out[B:] * 0.0, label.to(DEV), ignore_index=255)
    (l_ce + l_ctr).backward()
    torch.cuda.synchronize()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    ro, rf, _ = O.cavp_forward(sd2, image, audio, cfg["lds"], eval_mode=False)
    torch.manual_seed(77)
    r_ctr = contrast_loss(rf[:B], label, rf[B:], shuf, 0.1, 255, 32)
    r_ce = O.ce_loss_train(ro, label, B)
    (r_ce + r_ctr).backward()
    assert abs(float(l_ctr.item()) - float(r_ctr.item())) <= 1e-3 * max(1.0, abs(float(r_ctr.item())))
    assert abs(float(l_ce.item()) - float(r_ce.item())) <= 1e-4 * max(1.0, abs(float(r_ce.item())))
    mine = dict(m.named_parameters())
    coss = []
    for k, p in params.items():
        if p.grad is None or float(p.grad.norm()) < 1e-9:
            continue
        a, b = mine[k].grad.double().cpu().flatten(), p.grad.double().flatten()
        coss.append(float((a @ b) / (a.norm() * b.norm() + 1e-30)))
    coss = np.array(coss)
    print(f"CE+contrast: l_ctr {float(l_ctr.item()):.5f} (oracle {float(r_ctr.item()):.5f}); grad cosine min {coss.min():.5f} median {np.median(coss):.6f}")
    assert np.median(coss) >= 0.9995 and coss.min() >= 0.99


@pytest.mark.parametrize("split", [False, True], ids=["one_graph", "two_graphs"])
def test_graph_replays_stay_correct(split):
    """EVERY replay of the captured training step - not only the first - reproduces the eager step, tensor by tensor.  (Round 1
    shipped a captured hipMemsetAsync in the weight-gradient path: four gradients were inf / 1e25 from the second replay on
    while the first replay, and the two graphs compared with each other, looked fine.)"""
    cfg = dict(C=3, B=4, hw=(64, 64), lds=[False, False, False])
    B = cfg["B"]
    image, audio, label = [t.to(DEV) for t in synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=2)]
    m, sd = _build(cfg)
    bn_state = {k: v for k, v in sd.items() if "running_" in k or "num_batches" in k}
    l_ref = float(m.train_step(image, audio, label, all_reduce=False).item())
    ref = m._grad_arena.flat.clone()
    step = m.capture_train_step(image, audio, label, split=split)
    names = {id(p): k for k, p in m.named_parameters()}
    for it in range(4):
        m.load_state_dict(bn_state, strict=False)
        loss = float(step().item())
        torch.cuda.synchronize()
        assert abs(loss - l_ref) <= 1e-4 * max(1.0, abs(l_ref)), (it, loss, l_ref)
        for p in m._grad_arena.params:
            v = m._grad_arena.views[id(p)]
            off = (v.data_ptr() - m._grad_arena.flat.data_ptr()) // 4
            r = ref[off:off + v.numel()].view(v.shape)
            # run-to-run noise of the f32 atomics reaches ~0.3 of the largest entry on the worst layer4 tensors at B = 4
            # (DESIGN.md 6c); a gradient that was not cleared / was cleared late is inf or >> 1
            e, s = float((v - r).norm()), float(r.norm())
            assert torch.isfinite(v).all() and e <= 0.5 * s + 1e-6, (it, names[id(p)], e, s)


def test_clip_shaped_ce_plus_contrast_matches_reference_golden():
    """Config #5 (AVSBench-MS): one clip = 5 frames batched as B = 5 (the reference loops the frames at B = 1,
    trainer_cavp_avs_obj.py:317-330), loss = CE on `out[:B] + out[B:]*0` + ContrastLoss on the fusion halves
    (trainer_cavp_vpo_mono.py:171-189).  Golden: the REFERENCE's own model + ContrastLoss + autograd on these inputs
    (tools/make_golden.py, case c5_clip_train): both loss terms and every parameter gradient."""
    from cavp_amd.contrast import ContrastLoss
    z, cfg = load_case("c5_clip_train")
    B = cfg["B"]
    assert B == 5
    m, _ = _build(cfg)
    image, audio, _ = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=0)
    label = torch.from_numpy(z["label"].astype(np.int64)).to(DEV)
    label_shuf = torch.from_numpy(z["label_shuffle"].astype(np.int64)).to(DEV)
    out, fus, _ = m(image.to(DEV), audio.to(DEV), None, False)
    crit = ContrastLoss(temperature=0.1, ignore_idx=255, max_views=512)
    torch.manual_seed(4321)                         # the generator seeds its anchor sampling the same way
    l_ctr = crit(fus[:B], label, fus[B:], label_shuf)
    l_ce = F.cross_entropy(out[:B] + out[B:] * 0.0, label, ignore_index=255)
    (l_ce + l_ctr).backward()
    torch.cuda.synchronize()
    r_ce, r_ctr = float(z["loss_ce"][0]), float(z["loss_ctr"][0])
    print(f"clip: CE {float(l_ce):.6f} (reference {r_ce:.6f}), contrast {float(l_ctr):.6f} (reference {r_ctr:.6f})")
    assert abs(float(l_ce) - r_ce) <= 1e-4 * max(1.0, abs(r_ce))
    # the anchors are pixels of the fusion map picked by label only (same randperm stream): the loss differs by rounding
    assert abs(float(l_ctr) - r_ctr) <= 2e-3 * max(1.0, abs(r_ctr))
    params = dict(m.named_parameters())
    keys, vals = list(z["grad_norm_keys"]), z["grad_norm_vals"]
    rels = []
    for k, v in zip(keys, vals):
        g = params[k].grad
        assert g is not None, f"no gradient for {k}"
        rels.append(abs(float(g.double().norm()) - v) / max(v, 1e-9))
    rels = np.array(rels)
    print(f"clip: gradient-norm rel. error median {np.median(rels):.2e} max {rels.max():.2e} ({keys[int(rels.argmax())]})")
    # same noise floor as the B = 2 fixtures (batch-statistics BatchNorm on 5 samples in the ASPP pooling branch)
    assert np.median(rels) <= 5e-3 and rels.max() <= 5e-2
    for s_ in [s_ for s_ in z.files if s_.startswith("grad_sample/")]:
        k = s_[len("grad_sample/"):]
        g = params[k].grad.detach().float().cpu().contiguous().flatten()
        ref = z[s_]
        smp = g[:: max(1, g.numel() // 4096)][:4096].numpy()
        a, b = smp.astype(np.float64), ref.astype(np.float64)
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        assert cos >= 0.99, (k, cos)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_deterministic_mode_is_bit_reproducible(dtype):
    """Opt-in deterministic training (the reference sets cudnn.deterministic = True, main_vpo_mono.py:39-41): with the mode on,
    two runs of the same step - eager and as hipGraph replays - give bit-identical losses and gradients; with it off the f32
    atomics make them differ.  The deterministic result agrees with the default one up to that summation-order noise."""
    from cavp_amd import _lib
    cfg = dict(C=3, B=4, hw=(64, 64), lds=[False, False, False])
    B = cfg["B"]
    image, audio, label = [t.to(DEV) for t in synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=2)]

    def run(m, sd, step=None):
        m.load_state_dict({k: v for k, v in sd.items() if "running_" in k or "num_batches" in k}, strict=False)
        loss = (step() if step is not None else m.train_step(image, audio, label, all_reduce=False)).clone()
        torch.cuda.synchronize()
        return loss, m._grad_arena.flat.clone()

    m, sd = _build(cfg, dtype)
    l_a, g_a = run(m, sd)
    l_b, g_b = run(m, sd)
    assert not torch.equal(g_a, g_b), "the default mode is expected to be order-dependent (f32 atomics)"
    _lib.set_deterministic(True)
    try:
        assert _lib.is_deterministic()
        m2, sd2 = _build(cfg, dtype)
        l1, g1 = run(m2, sd2)
        l2, g2 = run(m2, sd2)
        assert torch.equal(l1, l2) and torch.equal(g1, g2), "eager steps differ in deterministic mode"
        step = m2.capture_train_step(image, audio, label)
        for _ in range(3):
            l3, g3 = run(m2, sd2, step)
            assert torch.equal(l3, l1) and torch.equal(g3, g1), "graph replay differs from the eager step in deterministic mode"
        if dtype == torch.float32:   # same math, other summation order
            assert abs(float(l1) - float(l_a)) <= 1e-4 * max(1.0, abs(float(l_a)))
            cos = float((g1.double() @ g_a.double()) / (g1.double().norm() * g_a.double().norm()))
            assert cos >= 0.999, cos
    finally:
        _lib.set_deterministic(False)
    assert not _lib.is_deterministic()


def test_second_stream_equals_single_stream():
    """The audio encoder runs on a second stream (forward and backward).  With every BatchNorm frozen the step has no chaotic
    amplifier and its only run-to-run noise is the order of a few f32 atomics (~1e-6), so a missing dependency or a shared scratch
    buffer between the streams would show: gradients with the second stream == gradients without, eagerly and as graph replays."""
    import cavp_amd.train as TR
    cfg = dict(C=3, B=4, hw=(96, 96), lds=[False, False, False])
    image, audio, label = [t.to(DEV) for t in synth_inputs(cfg["B"], cfg["hw"], audio_batch=2 * cfg["B"], num_classes=cfg["C"], seed=8)]

    def build():
        m, _ = _build(cfg)
        for mod in m.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.eval()
        return m

    old = TR._SIDE_STREAM
    try:
        TR._SIDE_STREAM = False
        m0 = build()
        l0 = float(m0.train_step(image, audio, label, all_reduce=False).item())
        ref = m0._grad_arena.flat.clone()
        TR._SIDE_STREAM = True
        m1 = build()
        for _ in range(3):
            l1 = float(m1.train_step(image, audio, label, all_reduce=False).item())
            torch.cuda.synchronize()
            assert m1._side_stream is not None
            err = float((m1._grad_arena.flat - ref).norm() / ref.norm())
            assert abs(l1 - l0) <= 1e-6 * max(1.0, abs(l0)) and err <= 1e-5, (l1, l0, err)
        step = m1.capture_train_step(image, audio, label)
        for _ in range(3):
            l2 = float(step().item())
            torch.cuda.synchronize()
            err = float((m1._grad_arena.flat - ref).norm() / ref.norm())
            assert abs(l2 - l0) <= 1e-6 * max(1.0, abs(l0)) and err <= 1e-5, (l2, l0, err)
        # inference forward: bit-identical with and without the second stream
        m1.eval()
        with torch.no_grad():
            a = m1(image, audio[:cfg["B"]], eval_mode=True)[0].clone()
            TR._SIDE_STREAM = False
            b = m1(image, audio[:cfg["B"]], eval_mode=True)[0]
        assert torch.equal(a, b)
    finally:
        TR._SIDE_STREAM = old


@pytest.mark.parametrize("streams", [False, True], ids=["one_stream", "default_streams"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_fused_bn_backward_statistics_equal_separate_reduce(dtype, streams):
    """Round 5: the data-gradient launch that completes the gradient of a BatchNorm + ReLU output applies the ReLU mask and sums
    g and g * zhat per tile in its epilogue (cavp_conv2d_nhwc_bnbwd); the BatchNorm backward then skips its reduce pass.  Whole
    training step (batch statistics, B = 8) with the fusion == the step with the separate reduce: same loss, same gradients up
    to summation order (f32) / up to the storage rounding of g (bf16), eagerly and as a graph replay; and the fusion is in use.
    `default_streams`: the product configuration (audio encoder on the side stream, down-sample branches on the branch stream).
    HIP on both sides - a consistency screen of the fused route; the parity anchors of the route are the reference-golden train steps
    (test_train_step_b8_matches_reference_f32 and friends), which run with it as the default."""
    import cavp_amd.train as TR
    from cavp_amd import train_ops as T
    cfg = dict(C=3, B=8, hw=(96, 96), lds=[False, False, False])
    image, audio, label = [t.to(DEV) for t in synth_inputs(cfg["B"], cfg["hw"], audio_batch=2 * cfg["B"], num_classes=cfg["C"], seed=10)]
    old, old_side = TR._FUSE_BN_BWD, TR._SIDE_STREAM
    calls = {"n": 0}
    orig = T.bn_act_bwd_reduce

    def counted(*a, **k):   # the separate reduce passes that still run
        calls["n"] += 1
        return orig(*a, **k)
    try:
        TR._SIDE_STREAM = streams
        TR._FUSE_BN_BWD = False
        T.bn_act_bwd_reduce = counted
        m0, _ = _build(cfg)
        m0.set_compute_dtype(dtype)
        m0.train()
        l0 = float(m0.train_step(image, audio, label, all_reduce=False).item())
        ref = m0._grad_arena.flat.clone().double()
        n_plain, calls["n"] = calls["n"], 0
        TR._FUSE_BN_BWD = True
        m1, _ = _build(cfg)
        m1.set_compute_dtype(dtype)
        m1.train()
        l1 = float(m1.train_step(image, audio, label, all_reduce=False).item())
        torch.cuda.synchronize()
        assert n_plain - calls["n"] >= 20, f"only {n_plain - calls['n']} of {n_plain} BatchNorm layers took the fused statistics"
        calls["n"] = n_plain - calls["n"]
        g1 = m1._grad_arena.flat.clone().double()
        cos = float((g1 @ ref) / (g1.norm() * ref.norm()))
        rel = float((g1 - ref).norm() / ref.norm())
        print(f"fused BN-backward statistics ({dtype}): {calls['n']} layers, loss {l1:.6f} vs {l0:.6f}, gradient cosine {cos:.6f}, rel err {rel:.2e}")
        assert abs(l1 - l0) <= 1e-6 * max(1.0, abs(l0))          # the forward is untouched
        # f32: summation order only (measured 2.7e-6 on MI355X); bf16: g is stored rounded once more on the fused route (measured
        # cosine 0.99980, relative error 1.9e-2)
        assert (cos >= 0.9999999 and rel <= 5e-5) if dtype == torch.float32 else (cos >= 0.999 and rel <= 5e-2), (cos, rel)
        T.bn_act_bwd_reduce = orig
        step = m1.capture_train_step(image, audio, label)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        g2 = m1._grad_arena.flat.clone().double()
        cos2 = float((g2 @ ref) / (g2.norm() * ref.norm()))
        assert cos2 >= (0.9999999 if dtype == torch.float32 else 0.999), cos2
    finally:
        T.bn_act_bwd_reduce = orig
        TR._FUSE_BN_BWD, TR._SIDE_STREAM = old, old_side


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_stem_pool_fused_with_bn_apply_is_bit_identical(dtype):
    """Round 6: the stem's bn1 -> ReLU -> max pool (resnet.py:187-190) runs as ONE pass over the conv output (cavp_maxpool_affine_nhwc:
    the pool compares act(z * scale + shift) rounded to the storage dtype, exactly the values scale_shift_act used to store), the
    un-pooled activation is never written; the backward routes the pooled gradient through the recorded arg-max and is then the same
    BatchNorm + ReLU backward as before.  Deterministic mode: loss and EVERY gradient bit-identical to the two-launch route.
    (HIP on both sides: a bit-identity screen; the reference-golden train steps run with the fused route as the default.)"""
    import cavp_amd.train as TR
    from cavp_amd import _lib
    cfg = dict(C=3, B=4, hw=(64, 96), lds=[False, False, False])
    image, audio, label = [t.to(DEV) for t in synth_inputs(cfg["B"], cfg["hw"], audio_batch=2 * cfg["B"], num_classes=cfg["C"], seed=12)]
    old = TR._FUSE_STEM_POOL
    _lib.set_deterministic(True)
    try:
        res = []
        for fused in (False, True):
            TR._FUSE_STEM_POOL = fused
            m, _ = _build(cfg)
            m.set_compute_dtype(dtype)
            m.train()
            loss = m.train_step(image, audio, label, all_reduce=False)
            torch.cuda.synchronize()
            res.append((float(loss.item()), m._grad_arena.flat.clone()))
        assert res[0][0] == res[1][0], (res[0][0], res[1][0])
        assert torch.equal(res[0][1], res[1][1]), float((res[0][1] - res[1][1]).abs().max())
    finally:
        _lib.set_deterministic(False)
        TR._FUSE_STEM_POOL = old
