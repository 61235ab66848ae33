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


def test_oracle_train_matches_reference():
    z, cfg = load_case("c1p_train")
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
        assert np.abs(s - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k


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
