"""Pin the CPU oracle (oracle/cavp_oracle.py) against outputs of the reference itself (tests/golden/*.npz,
made by tools/make_golden.py in the authoring container).  CPU only."""
import numpy as np
import pytest
import torch

from cavp_amd.synth import synth_inputs, synth_state_dict
from oracle import cavp_oracle as O
from tests._golden_util import check_tap, load_case
from tests.shapes import cavp_state_shapes

ATOL = 1e-5  # restatement equals the reference to <= 1e-5


@pytest.mark.parametrize("case", ["c1p_eval", "ragged_eval", "c1_eval"])
def test_oracle_eval_matches_reference(case):
    z, cfg = load_case(case)
    sd = synth_state_dict(cavp_state_shapes(cfg["C"]), seed=1)
    image, audio, _ = synth_inputs(cfg["B"], cfg["hw"], num_classes=cfg["C"], seed=0)
    taps = {}
    with torch.no_grad():
        out, fus, pack = O.cavp_forward(sd, image, audio, cfg["lds"], eval_mode=True, taps=taps)
    taps.update(out_pred=out, out_fusion=fus, pack_audio=pack["audio"], pack_visual=pack["visual"],
                pack_attn_v=pack["attn_v"])
    for k in sorted(taps):
        scale = max(1.0, float(np.abs(z["sample/" + k]).max()))
        check_tap(z, k, taps[k], ATOL * scale, what=case + ":")
    if "full/out_pred" in z:
        assert np.abs(out.numpy() - z["full/out_pred"]).max() <= ATOL * 10


@pytest.mark.parametrize("case", ["c1p_train", "c1_train", "c1p_train_b8"])
def test_oracle_train_matches_reference(case):
    """c1_train = config #1's model (OS8, 22 classes) in training mode; c1p_train_b8 = the native model on a well-conditioned
    batch (8 images + 16 audio clips), the tight gradient target of tests/test_gpu_train_model.py."""
    z, cfg = load_case(case)
    B = cfg["B"]
    sd = synth_state_dict(cavp_state_shapes(cfg["C"]), seed=1)
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    taps = {}
    out, fus, pack = O.cavp_forward(sd2, image, audio, cfg["lds"], eval_mode=False, taps=taps)
    loss = O.ce_loss_train(out, label, B)
    assert abs(loss.item() - float(z["loss"][0])) <= 1e-5
    taps.update(out_pred=out, out_fusion=fus, pack_audio=pack["audio"], pack_visual=pack["visual"],
                pack_attn_v=pack["attn_v"])
    for k in sorted(taps):
        scale = max(1.0, float(np.abs(z["sample/" + k]).max()))
        check_tap(z, k, taps[k], 2 * ATOL * scale, what="train:")
    loss.backward()
    keys = list(z["grad_norm_keys"])
    vals = z["grad_norm_vals"]
    for k, v in zip(keys, vals):
        g = params[k].grad
        assert g is not None, k
        n = g.double().norm().item()
        assert abs(n - v) <= 1e-4 * max(v, 1e-3), (k, n, v)
    # parameters the reference leaves without grad (pos_embed_*, cls_head, dead audio branch has shared weights)
    for k, p in params.items():
        if k not in keys:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    for k in [s[len("grad_sample/"):] for s in z.files if s.startswith("grad_sample/")]:
        g = params[k].grad
        ref = z["grad_sample/" + k]
        n = g.numel()
        s = g.flatten()[:: max(1, n // 4096)][:4096].numpy()
        # (c1_train: OS8 keeps 13 more batch-statistics BatchNorm layers at 28 x 28, and the f32 summation order of the conv
        # backward differs with the thread count the fixture was generated with: 1.6e-4 of max|grad| measured on the stem)
        tol = 1e-5 if case == "c1p_train" else 1e-3   # (c1p_train_b8: the summation-order remark applies as well)
        assert np.abs(s - ref).max() <= tol * max(1.0, np.abs(ref).max()), k


def test_oracle_pvt_matches_reference():
    """config #4: seg_model="PVT" (PVTv2-B5 backbone, latent 112) eval forward."""
    z, cfg = load_case("pvt_eval")
    sd = synth_state_dict(cavp_state_shapes(cfg["C"], "PVT"), seed=1)
    image, audio, _ = synth_inputs(cfg["B"], cfg["hw"], num_classes=cfg["C"], seed=0)
    taps = {}
    with torch.no_grad():
        out, fus, pack = O.cavp_forward(sd, image, audio, cfg["lds"], eval_mode=True, taps=taps, seg_model="PVT")
    taps = {k: v for k, v in taps.items() if k.startswith("stage")}
    taps.update(out_pred=out, out_fusion=fus, pack_visual=pack["visual"], pack_attn_v=pack["attn_v"])
    for k in sorted(taps):
        scale = max(1.0, float(np.abs(z["sample/" + k]).max()))
        check_tap(z, k, taps[k], 2 * ATOL * scale, what="pvt:")


def _timm_drop_scales(z, B):
    """The DropPath factors of the fixture, re-drawn exactly as timm 0.4.9 draws them (one torch.rand((B,1,1)) per branch with
    probability > 0, PVTv2-B5's linspace(0, 0.1, 52) schedule, pvt.py:229) and checked against what the reference applied."""
    depth = sum(O.PVT_DEPTHS)
    probs = [float(x) for x in torch.linspace(0, 0.1, depth) for _ in range(2)]
    torch.manual_seed(int(z["seed"][0]))
    scales, k = [], 0
    for p in probs:
        if p > 0:
            s = (1 - p + torch.rand((B, 1, 1))).floor_().view(B) / (1 - p)
            assert np.abs(s.numpy() - z["drop_scales"][k]).max() <= 1e-4, (k, s, z["drop_scales"][k])
            scales.append(s)
            k += 1
        else:
            scales.append(None)
    assert k == z["drop_scales"].shape[0]
    return scales


def test_oracle_pvt_train_matches_reference():
    """config #4's model in train mode (golden pvt_train from the reference's own autograd): DropPath, batch-stat BN in the
    decoder, CE, every parameter's gradient norm and sampled sentinel gradients."""
    z, cfg = load_case("pvt_train")
    B = cfg["B"]
    sd = synth_state_dict(cavp_state_shapes(cfg["C"], "PVT"), seed=1)
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=3)
    scales = _timm_drop_scales(z, B)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    taps = {}
    out, fus, pack = O.cavp_forward(sd2, image, audio, cfg["lds"], eval_mode=False, taps=taps, seg_model="PVT", drop_scales=scales)
    loss = O.ce_loss_train(out, label, B)
    assert abs(loss.item() - float(z["loss"][0])) <= 1e-5
    taps = {k: v for k, v in taps.items() if k.startswith("stage")}
    taps.update(out_pred=out, out_fusion=fus, pack_visual=pack["visual"], pack_attn_v=pack["attn_v"])
    for k in sorted(taps):
        scale = max(1.0, float(np.abs(z["sample/" + k]).max()))
        check_tap(z, k, taps[k], 5 * ATOL * scale, what="pvt train:")   # 52 blocks deep: 2.3e-5 on the attention gate
    loss.backward()
    for k, v in zip(list(z["grad_norm_keys"]), z["grad_norm_vals"]):
        g = params[k].grad
        assert g is not None, k
        n = g.double().norm().item()
        # the whole backbone gradient passes through the ASPP pooling branch's 2-sample BatchNorm: two f32 evaluations of this
        # step (thread counts of the fixture run and of this run) differ by 2.6e-4 (median) .. 1.1e-3 (norm4.bias) in norm
        assert abs(n - v) <= 3e-3 * max(v, 1e-3), (k, n, v)
    for k in [s[len("grad_sample/"):] for s in z.files if s.startswith("grad_sample/")]:
        g, ref = params[k].grad, z["grad_sample/" + k]
        s = g.flatten()[:: max(1, g.numel() // 4096)][:4096].numpy()
        assert np.abs(s - ref).max() <= 3e-3 * max(1e-2, np.abs(ref).max()), (k, np.abs(s - ref).max(), np.abs(ref).max())


def test_oracle_clip_ce_plus_contrast_matches_reference():
    """config #5: a 5-frame clip batched as B = 5, CE + ContrastLoss on the fusion halves (golden c5_clip_train from the
    reference's own model, loss and autograd): pins oracle.cavp_oracle + oracle.contrast_oracle together on that shape."""
    import torch
    from oracle.contrast_oracle import contrast_loss
    z, cfg = load_case("c5_clip_train")
    B = cfg["B"]
    sd = synth_state_dict(cavp_state_shapes(cfg["C"]), seed=1)
    image, audio, _ = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=0)
    label = torch.from_numpy(z["label"].astype(np.int64))
    label_shuf = torch.from_numpy(z["label_shuffle"].astype(np.int64))
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    out, fus, _ = O.cavp_forward(sd2, image, audio, cfg["lds"], eval_mode=False)
    torch.manual_seed(4321)
    l_ctr = contrast_loss(fus[:B], label, fus[B:], label_shuf, 0.1, 255, 512)
    l_ce = O.ce_loss_train(out, label, B)
    assert abs(l_ce.item() - float(z["loss_ce"][0])) <= 1e-5
    assert abs(l_ctr.item() - float(z["loss_ctr"][0])) <= 1e-4
    (l_ce + l_ctr).backward()
    for k, v in zip(list(z["grad_norm_keys"]), z["grad_norm_vals"]):
        n = params[k].grad.double().norm().item()
        assert abs(n - v) <= 2e-4 * max(v, 1e-3), (k, n, v)
