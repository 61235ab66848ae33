"""End-to-end parity of the MI355X CAVP forward (HIP kernels through the C-ABI) against
  (1) the committed golden vectors produced by the reference itself, and
  (2) the CPU oracle on the same seeded inputs.
fp32 path bar: |logit - reference| <= 1e-3 per pixel (BASELINE.json north_star)."""
import types

import numpy as np
import pytest
import torch

from cavp_amd.synth import synth_inputs, synth_state_dict
from tests._golden_util import check_tap, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOGIT_TOL = 1e-3


def _args(cfg):
    return types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=cfg["lds"], audio_backbone="vgg",
                                 num_classes=cfg["C"], batch_size=cfg["B"], local_rank="cpu")


def build_model(cfg, dtype=torch.float32):
    from cavp_amd.cavp_model import CAVP
    m = CAVP(50, None, num_classes=cfg["C"], audio_backbone_pretrain_path=None, visual_backbone=50, args=_args(cfg))
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    m.eval().to(DEV).set_compute_dtype(dtype)
    return m, sd


@pytest.mark.parametrize("case", ["c1p_eval", "ragged_eval", "c1_eval"])
def test_forward_matches_reference_golden_f32(case):
    z, cfg = load_case(case)
    m, _ = build_model(cfg)
    image, audio, _ = synth_inputs(cfg["B"], cfg["hw"], num_classes=cfg["C"], seed=0)
    taps = {}
    with torch.no_grad():
        out, fus, pack = m._forward_hip(image.to(DEV), audio.to(DEV), duplicate_visual=False, taps=taps)
    torch.cuda.synchronize()
    assert out.shape == (cfg["B"], cfg["C"]) + tuple(cfg["hw"]) and out.is_contiguous()
    taps.update(out_fusion=fus, pack_audio=pack["audio"], pack_visual=pack["visual"], pack_attn_v=pack["attn_v"])
    report = {}
    for k in sorted(taps):
        scale = max(1.0, float(np.abs(z["sample/" + k]).max()))
        report[k] = check_tap(z, k, taps[k], 2e-4 * scale, what=case + ":")
    report["out_pred"] = check_tap(z, "out_pred", out, LOGIT_TOL, what=case + ":")
    if "full/out_pred" in z:
        err = np.abs(out.cpu().numpy() - z["full/out_pred"]).max()
        assert err <= LOGIT_TOL, f"full logits max err {err:.3e}"
        report["out_pred_full"] = float(err)
    print(case, {k: f"{v:.2e}" for k, v in report.items()})


def test_forward_matches_oracle_f32_and_api():
    """Same seeded inputs through the public forward() (eval_mode=True) vs the CPU oracle, full tensors."""
    from oracle import cavp_oracle as O
    cfg = dict(C=5, B=2, hw=(64, 96), lds=[False, False, False])
    m, sd = build_model(cfg)
    image, audio, _ = synth_inputs(cfg["B"], cfg["hw"], num_classes=cfg["C"], seed=3)
    with torch.no_grad():
        out, fus, pack = m(image.to(DEV), audio.to(DEV), eval_mode=True)
        ro, rf, rp = O.cavp_forward(sd, image, audio, cfg["lds"], eval_mode=True)
    assert float((out.cpu() - ro).abs().max()) <= LOGIT_TOL
    assert float((fus.cpu() - rf).abs().max()) <= 1e-3
    assert float((pack["visual"].cpu() - rp["visual"]).abs().max()) <= 1e-3 * max(1.0, float(rp["visual"].abs().max()))
    assert float((pack["audio"].cpu() - rp["audio"]).abs().max()) <= 1e-3
    assert float((pack["attn_v"].cpu() - rp["attn_v"]).abs().max()) <= 1e-4
    assert pack["attn_v"].shape == rp["attn_v"].shape and fus.shape == rf.shape


def test_train_call_convention_no_grad():
    """model(image[B], audio[2B], None, ow_flag) with BN in eval state: visual features duplicated to 2B
    (cavp_model.py:175-188); compared with the oracle's forward_train using running stats."""
    from oracle import cavp_oracle as O
    cfg = dict(C=2, B=2, hw=(64, 64), lds=[False, False, False])
    m, sd = build_model(cfg)
    image, audio, _ = synth_inputs(cfg["B"], cfg["hw"], audio_batch=4, num_classes=2, seed=4)
    with torch.no_grad():
        out, fus, pack = m(image.to(DEV), audio.to(DEV), None, False)
        ro, rf, rp = O.cavp_forward(sd, image, audio, cfg["lds"], eval_mode=False, bn_train=False)
    assert out.shape == (4, 2, 64, 64)
    assert float((out.cpu() - ro).abs().max()) <= LOGIT_TOL
    assert float((fus.cpu() - rf).abs().max()) <= 1e-3


def test_forward_bf16_tracks_reference():
    """bf16 storage / f32 accumulate path: reported, looser bar (not the 1e-3 claim)."""
    z, cfg = load_case("c1p_eval")
    m, _ = build_model(cfg, torch.bfloat16)
    image, audio, _ = synth_inputs(cfg["B"], cfg["hw"], num_classes=cfg["C"], seed=0)
    with torch.no_grad():
        out, fus, pack = m(image.to(DEV), audio.to(DEV), eval_mode=True)
    ref = z["full/out_pred"]
    err = np.abs(out.cpu().numpy() - ref)
    rel = err.max() / np.abs(ref).max()
    print(f"bf16 logits: max abs err {err.max():.4f}, mean {err.mean():.4f}, rel-to-max {rel:.4f}")
    assert rel <= 0.08 and err.mean() <= 0.05 * np.abs(ref).mean() + 0.02
    agree = (out.argmax(1).cpu().numpy() == ref.argmax(1)).mean()
    assert agree >= 0.97, f"argmax agreement {agree:.4f}"


def test_cpu_tensors_fail_loudly():
    from cavp_amd._lib import CavpError
    cfg = dict(C=2, B=1, hw=(32, 32), lds=[False, False, False])
    m, _ = build_model(cfg)
    image, audio, _ = synth_inputs(1, (32, 32), seed=0)
    with pytest.raises(CavpError):
        m(image, audio, eval_mode=True)


@pytest.mark.parametrize("case", ["pvt_eval", "pvt_eval_512"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, None)], ids=["f32", "bf16"])
def test_pvt_forward_matches_reference_golden(dtype, tol, case):
    """config #4: PVTv2-B5 backbone (MFMA softmax attention, SR convs, depth-wise MLP) + 112-d fusion + decoder, at 256 x 256
    and at the config's own 512 x 512 (config_avss.py:12-13; 16384 fusion tokens, 71 classes)."""
    from cavp_amd.cavp_model import CAVP
    z, cfg = load_case(case)
    a = _args(cfg)
    a.seg_model = "PVT"
    a.allow_random_pvt = True   # synthetic weights are loaded right after
    m = CAVP(50, None, num_classes=cfg["C"], args=a)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    m.eval().to(DEV).set_compute_dtype(dtype)
    image, audio, _ = synth_inputs(cfg["B"], cfg["hw"], num_classes=cfg["C"], seed=0)
    taps = {}
    with torch.no_grad():
        out, fus, pack = m._forward_hip(image.to(DEV), audio.to(DEV), duplicate_visual=False, taps=taps)
    torch.cuda.synchronize()
    got = {f"stage{i + 1}": taps[f"layer{i + 1}"] for i in range(4)}
    got.update(out_pred=out, out_fusion=fus, pack_visual=pack["visual"], pack_attn_v=pack["attn_v"])
    rep = {}
    for k, t in got.items():
        ref = z["sample/" + k]
        scale = max(1.0, float(np.abs(ref).max()))
        if tol is not None:
            rep[k] = check_tap(z, k, t, (tol if k == "out_pred" else 3e-4) * scale, what="pvt:")
        else:
            from tests._golden_util import sample
            s, _ = sample(t)
            rel = float(np.abs(s - ref).max() / scale)
            rep[k] = rel
            # measured: <= 2.9e-2 of the tap's largest value (stage 3: 40 blocks deep, sampled elements), logits 9e-3
            assert rel <= 0.05, (k, rel)
    print("pvt", dtype, {k: f"{v:.2e}" for k, v in rep.items()})
