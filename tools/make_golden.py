#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE model (read-only import from /root/reference).

Authoring-container only: /root/reference does not exist on the GPU box and nothing here is imported by the
product or the tests.  The reference publishes no golden vectors, so these fixtures ARE the pin
for oracle/cavp_oracle.py: weights and inputs come from cavp_amd.synth (pure function of key/shape/seed), are
loaded into the reference `CAVP` with strict=True, and the reference's outputs are stored as
  * full tensors where small (final logits of the B=2/C=2 case, fea_a, attn_v samples ...)
  * for big tensors: a fixed strided sample (<= 4096 values) + fp64 sum / abs-sum checksums.

Import recipe = SURVEY.md Appendix C (stub packages in tools/_shims for loguru/easydict/timm/torchvision which
are not installed here; hard-coded checkpoint load at resnet.py:224-227 neutralised).

usage: python tools/make_golden.py [--out tests/golden]
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "_shims"), "/root/reference", REPO]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import models.visual.backbones.resnet as _R  # noqa: E402

_R.load_model = lambda model, f, is_restore=False: model
from easydict import EasyDict  # noqa: E402
from models.cavp_model import CAVP  # noqa: E402

from cavp_amd.synth import synth_inputs, synth_state_dict  # noqa: E402

NSAMP = 4096

CASES = {
    # name: dict(os, C, B, hw, mode)
    "c1p_eval": dict(lds=[False, False, False], C=2, B=2, hw=(224, 224), mode="eval"),     # config_avss_binary.py shape
    "c1_eval": dict(lds=[False, True, True], C=22, B=2, hw=(224, 224), mode="eval"),        # config_vpo_ss.py plumbing @224
    "ragged_eval": dict(lds=[False, False, False], C=24, B=3, hw=(96, 160), mode="eval"),   # odd batch, H != W
    "c1p_train": dict(lds=[False, False, False], C=2, B=2, hw=(224, 224), mode="train"),   # BN batch stats, audio 2B, CE grads
    # the same step on a well-conditioned batch (B = 8 images + 16 audio clips): the batch-statistics BatchNorm of the ASPP pooling
    # branch normalises over 8 samples instead of 2, so the REFERENCE's own f32 gradients are a tight target (round-4 review)
    "c1p_train_b8": dict(lds=[False, False, False], C=2, B=8, hw=(224, 224), mode="train"),
    # config #1's model (config_vpo_ss.py plumbing: OS8, 22 classes) in training mode
    "c1_train": dict(lds=[False, True, True], C=22, B=2, hw=(224, 224), mode="train"),
    # config #5 (AVSBench-MS: a clip = 5 frames batched as B = 5, reference loops the frames at B = 1,
    # trainer_cavp_avs_obj.py:317-330): CE + ContrastLoss on the fusion halves, trainer_cavp_vpo_mono.py:171-189
    "c5_clip_train": dict(lds=[False, False, False], C=2, B=5, hw=(224, 224), mode="train", contrast=True),
}


def clip_labels(B, hw, num_classes, seed):
    """Blocky label maps for the clip fixture (ContrastLoss needs classes with many pixels): one rectangle of class 1 per frame
    that drifts over the clip, a few rows of ignore_index; the shuffled-audio labels are the matched ones where the shuffled
    clip index equals the original and background elsewhere (trainer_cavp_vpo_mono.py:173-179)."""
    g = torch.Generator().manual_seed(seed)
    gt = torch.zeros((B,) + tuple(hw), dtype=torch.long)
    for b in range(B):
        h0 = 20 + 12 * b
        w0 = int(torch.randint(10, hw[1] - 130, (1,), generator=g))
        gt[b, h0:h0 + 90, w0:w0 + 120] = 1 if num_classes == 2 else 1 + b % (num_classes - 1)
        gt[b, :4, :] = 255
    gs = gt.clone()
    gs[1:] = 0
    return gt, gs

SENTINELS = [
    "backbone.backbone.conv1.0.weight", "backbone.backbone.layer1.0.conv1.weight",
    "backbone.backbone.layer4.2.conv3.weight", "backbone.backbone.layer4.2.bn3.weight",
    "segment.aspp.red_conv.weight", "segment.aspp.map_convs.2.weight", "segment.reduce.0.weight",
    "cross_att.blocks.0.attn.q.weight", "cross_att.blocks.0.attn.k.weight", "cross_att.blocks.0.mlp.fc2.weight",
    "cross_att.blocks.0.norm1.weight", "cross_att.patch_embed_a.proj.weight", "visual_projector.fc1.weight",
    "audio_backbone.backbone.embeddings.0.weight", "audio_backbone.backbone.features.0.weight",
    "segment.upsample.last_conv.0.weight", "segment.upsample.classifier.weight", "segment.upsample.classifier.bias",
]


def sample(t):
    t = t.detach().to(torch.float32).contiguous().flatten()
    n = t.numel()
    stride = max(1, n // NSAMP)
    return t[::stride][:NSAMP].numpy().copy(), np.array([t.double().sum().item(), t.double().abs().sum().item(), n],
                                                        dtype=np.float64)


def run_case(name, cfg, out_dir):
    args = EasyDict(seg_model="DeepLabV3Plus", last_three_dilation_stride=cfg["lds"], audio_backbone="vgg",
                    num_classes=cfg["C"], batch_size=cfg["B"], local_rank="cpu")
    m = CAVP(50, None, num_classes=cfg["C"], audio_backbone_pretrain_path=None, visual_backbone=50, args=args)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    if name == "c1p_eval":  # the key tree + shapes of the reference state_dict, C = num_classes
        import json
        shapes = {k: ["C" if (k.startswith("segment.upsample.classifier") and i == 0) else int(d)
                      for i, d in enumerate(v.shape)] for k, v in m.state_dict().items()}
        with open(os.path.join(out_dir, "state_dict_shapes.json"), "w") as f:
            json.dump(shapes, f, indent=0)
    train = cfg["mode"] == "train"
    B = cfg["B"]
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B if train else B, num_classes=cfg["C"], seed=0)

    taps = {}

    def hook(key):
        def fn(mod, inp, out):
            taps[key] = out
        return fn

    bb = m.backbone.backbone
    for i in range(4):
        getattr(bb, f"layer{i + 1}").register_forward_hook(hook(f"layer{i + 1}"))
    m.segment.aspp.register_forward_hook(hook("aspp"))
    m.segment.upsample.last_conv.register_forward_hook(hook("last_conv"))
    m.segment.upsample.classifier.register_forward_hook(hook("logits_lowres"))
    orig_fusion = m.forward_fusion

    def fusion(visual, fea_a):
        taps["fea_v"], taps["fea_a"] = visual, fea_a
        return orig_fusion(visual, fea_a)

    m.forward_fusion = fusion

    store = {}
    if train:
        m.train()
        out, fus, pack = m(image, audio, None, False)
        output = out[:B] + out[B:] * 0.0                       # trainer_cavp_vpo_mono.py:171
        if cfg.get("contrast"):
            from loss.contrastive_aud import ContrastLoss
            label, label_shuf = clip_labels(B, cfg["hw"], cfg["C"], seed=21)
            store["label"] = label.numpy().astype(np.int16)
            store["label_shuffle"] = label_shuf.numpy().astype(np.int16)
            crit = ContrastLoss(temperature=0.1, ignore_idx=255, max_views=512)
            torch.manual_seed(4321)                               # ContrastLoss samples anchors with torch.randperm
            l_ctr = crit(fus[:B], label, fus[B:], label_shuf)       # :178-181
            l_ce = F.cross_entropy(output, label, ignore_index=255)
            loss = l_ce + l_ctr                                   # :189
            store["loss_ce"] = np.array([l_ce.item()], dtype=np.float64)
            store["loss_ctr"] = np.array([l_ctr.item()], dtype=np.float64)
        else:
            loss = F.cross_entropy(output, label, ignore_index=255)  # loss/losser.py:60-62
        loss.backward()
        store["loss"] = np.array([loss.item()], dtype=np.float64)
        gn = {}
        for k, p in m.named_parameters():
            if p.grad is not None:
                gn[k] = p.grad.double().norm().item()
        store["grad_norm_keys"] = np.array(sorted(gn), dtype=object)
        store["grad_norm_vals"] = np.array([gn[k] for k in sorted(gn)], dtype=np.float64)
        params = dict(m.named_parameters())
        for k in SENTINELS:
            s, c = sample(params[k].grad)
            store["grad_sample/" + k], store["grad_cksum/" + k] = s, c
    else:
        m.eval()
        with torch.no_grad():
            out, fus, pack = m(image, audio, eval_mode=True)

    taps.update(out_pred=out, out_fusion=fus, pack_audio=pack["audio"], pack_visual=pack["visual"],
                pack_attn_v=pack["attn_v"])
    for k, t in taps.items():
        s, c = sample(t)
        store["sample/" + k], store["cksum/" + k] = s, c
        store["shape/" + k] = np.array(t.shape, dtype=np.int64)
    if out.numel() <= 2 * 2 * 224 * 224:
        store["full/out_pred"] = out.detach().numpy().astype(np.float32)
    store["full/fea_a"] = taps["fea_a"].detach().numpy().astype(np.float32)
    store["cfg/lds"] = np.array(cfg["lds"], dtype=np.int64)
    store["cfg/CBHW"] = np.array([cfg["C"], B, cfg["hw"][0], cfg["hw"][1]], dtype=np.int64)
    store["cfg/train"] = np.array([int(train)], dtype=np.int64)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e3:.0f} kB); |out|max={out.abs().max().item():.3f} "
          f"std={out.std().item():.3f}; fusion std={fus.std().item():.3f}; fea_a mean={taps['fea_a'].mean().item():.3f}; "
          + " ".join(f"{k}:{taps[k].std().item():.2f}" for k in ("layer1", "layer2", "layer3", "layer4", "aspp")))


def run_pvt(out_dir, hw=(256, 256), C=71, B=1, name="pvt_eval"):
    """config #4 (seg_model="PVT", config_avss.py shape family): reference forward with the hard-coded
    `torch.load("../ckpts/pretrained/pvt_v2_b5.pth")` (cavp_model.py:109) patched to a fresh random state."""
    import models.cavp_model as CM
    from models.visual.backbones.pvt.pvt import pvt_v2_b5
    real_load = torch.load

    def fake_load(path, *a, **k):
        if "pvt_v2_b5" in str(path):
            sd = pvt_v2_b5().state_dict()
            sd["head.weight"], sd["head.bias"] = torch.zeros(1), torch.zeros(1)
            return sd
        return real_load(path, *a, **k)
    torch.load = fake_load
    try:
        args = EasyDict(seg_model="PVT", last_three_dilation_stride=[False, False, False], audio_backbone="vgg",
                        num_classes=C, batch_size=B, local_rank="cpu")
        m = CAVP(50, None, num_classes=C, audio_backbone_pretrain_path=None, visual_backbone=50, args=args)
    finally:
        torch.load = real_load
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    import json
    with open(os.path.join(out_dir, "state_dict_shapes_pvt.json" if name == "pvt_eval" else os.devnull), "w") as f:
        json.dump({k: ["C" if (k.startswith("segment.upsample.classifier") and i == 0) else int(d) for i, d in enumerate(v.shape)]
                   for k, v in m.state_dict().items()}, f, indent=0)
    image, audio, _ = synth_inputs(B, hw, num_classes=C, seed=0)
    taps = {}
    orig = m.backbone.forward_features

    def ff(x):
        outs = orig(x)
        for i, o in enumerate(outs):
            taps[f"stage{i + 1}"] = o
        return outs
    m.backbone.forward_features = ff
    m.eval()
    with torch.no_grad():
        out, fus, pack = m(image, audio, eval_mode=True)
    taps.update(out_pred=out, out_fusion=fus, pack_visual=pack["visual"], pack_attn_v=pack["attn_v"])
    store = {}
    for k, t in taps.items():
        s_, c_ = sample(t)
        store["sample/" + k], store["cksum/" + k] = s_, c_
        store["shape/" + k] = np.array(t.shape, dtype=np.int64)
    store["cfg/CBHW"] = np.array([C, B, hw[0], hw[1]], dtype=np.int64)
    store["cfg/lds"] = np.array([0, 0, 0], dtype=np.int64)
    store["cfg/train"] = np.array([0], dtype=np.int64)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name}: wrote {path}; |out|max={out.abs().max().item():.3f} " +
          " ".join(f"{k}:{taps[k].std().item():.2f}" for k in ("stage1", "stage2", "stage3", "stage4")))


def run_f64_arbiter(out_dir, case="c1p_train_b8"):
    """NOT a reference output: the ORACLE (oracle/cavp_oracle.py, pinned to the reference to <= 1e-5 in f32 by
    tests/test_oracle_golden.py) evaluated in float64 on the inputs of `case` - the exact-arithmetic value of the same graph.
    It arbitrates between two f32 evaluations: at B = 8 the reference's OWN f32 gradients are 1.3e-2 .. 3.0e-2 (relative L2) away
    from it on the backbone / ASPP tensors (50 batch-statistics BatchNorm layers amplify rounding), so "HIP f32 vs reference f32"
    cannot be held tighter than that, while "HIP f32 is as close to the exact value as the reference's f32 run is" can
    (tests/test_gpu_train_model.py::test_train_step_b8_matches_reference_f32)."""
    sys.path.insert(0, REPO)
    from oracle import cavp_oracle as O
    z = np.load(os.path.join(out_dir, case + ".npz"), allow_pickle=True)
    C, B, H, W = [int(v) for v in z["cfg/CBHW"]]
    lds = [bool(v) for v in z["cfg/lds"]]
    shapes = {k: tuple(C if d == "C" else d for d in v) for k, v in json_load(os.path.join(out_dir, "state_dict_shapes.json")).items()}
    sd = synth_state_dict(shapes, seed=1)
    image, audio, label = synth_inputs(B, (H, W), audio_batch=2 * B, num_classes=C, seed=0)
    params = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    sd2.update(params)
    out, fus, pack = O.cavp_forward(sd2, image.double(), audio.double(), lds, eval_mode=False, taps={})
    loss = O.ce_loss_train(out, label, B)
    loss.backward()
    store = {"loss": np.array([loss.item()], dtype=np.float64)}
    keys = list(z["grad_norm_keys"])
    store["grad_norm_keys"] = np.array(keys, dtype=object)
    store["grad_norm_vals"] = np.array([params[k].grad.norm().item() for k in keys], dtype=np.float64)
    for s_ in [s_ for s_ in z.files if s_.startswith("grad_sample/")]:
        k = s_[len("grad_sample/"):]
        g = params[k].grad.flatten()
        store["grad_sample/" + k] = g[:: max(1, g.numel() // NSAMP)][:NSAMP].numpy().astype(np.float64)
    path = os.path.join(out_dir, case + "_f64.npz")
    np.savez_compressed(path, **store)
    print(f"{case}_f64 (oracle in float64): wrote {path}; loss {loss.item():.9f} (reference f32 {float(z['loss'][0]):.9f})")


def json_load(path):
    import json
    with open(path) as f:
        return json.load(f)


PVT_SENTINELS = [
    "backbone.patch_embed1.proj.weight", "backbone.patch_embed1.norm.weight", "backbone.block1.0.attn.sr.weight",
    "backbone.block1.0.attn.q.weight", "backbone.block1.2.attn.kv.weight", "backbone.block2.3.mlp.dwconv.dwconv.weight",
    "backbone.block2.3.mlp.dwconv.dwconv.bias", "backbone.patch_embed3.proj.weight", "backbone.block3.17.attn.proj.weight",
    "backbone.block3.39.mlp.fc1.weight", "backbone.block3.5.attn.norm.weight", "backbone.block4.1.attn.kv.weight",
    "backbone.block4.2.mlp.fc2.weight", "backbone.norm4.weight", "segment.aspp.red_conv.weight", "segment.reduce.0.weight",
    "cross_att.blocks.0.attn.q.weight", "visual_projector.fc1.weight", "segment.upsample.classifier.weight",
]


def run_pvt_train(out_dir, hw=(64, 96), C=5, B=2, name="pvt_train", seed=99):
    """config #4's model in TRAIN mode (batch-stat BN in the decoder, timm DropPath with drop_path_rate 0.1 in the backbone,
    pvt.py:413-421): forward_train + CE (trainer_cavp_vpo_mono.py:171,187) + backward.  torch.manual_seed(seed) right before
    the forward: the backbone is the first consumer of the RNG, so the DropPath draws (one torch.rand((B,1,1)) per branch
    with probability > 0, in forward order) are reproducible; the factors mask / keep_prob are stored as well."""
    import models.cavp_model as CM   # noqa: F401
    from models.visual.backbones.pvt.pvt import pvt_v2_b5
    import timm.models.layers as TL
    real_load = torch.load

    def fake_load(path, *a, **k):
        if "pvt_v2_b5" in str(path):
            sd = pvt_v2_b5().state_dict()
            sd["head.weight"], sd["head.bias"] = torch.zeros(1), torch.zeros(1)
            return sd
        return real_load(path, *a, **k)
    torch.load = fake_load
    try:
        args = EasyDict(seg_model="PVT", last_three_dilation_stride=[False, False, False], audio_backbone="vgg",
                        num_classes=C, batch_size=B, local_rank="cpu")
        m = CAVP(50, None, num_classes=C, audio_backbone_pretrain_path=None, visual_backbone=50, args=args)
    finally:
        torch.load = real_load
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    image, audio, label = synth_inputs(B, hw, audio_batch=2 * B, num_classes=C, seed=3)
    drawn = []
    for mod in m.modules():
        if isinstance(mod, TL.DropPath):
            def hook(mod_, inp, out, _d=drawn):
                x = inp[0]
                if mod_.drop_prob and mod_.training:
                    nz = x.flatten(1).abs().sum(1) > 0
                    ratio = (out.flatten(1).abs().sum(1) / x.flatten(1).abs().sum(1).clamp_min(1e-30))
                    _d.append(torch.where(nz, ratio, torch.zeros_like(ratio)).detach().float())
            mod.register_forward_hook(hook)
    taps = {}
    orig = m.backbone.forward_features

    def ff(x):
        outs = orig(x)
        for i, o in enumerate(outs):
            taps[f"stage{i + 1}"] = o
        return outs
    m.backbone.forward_features = ff
    m.train()
    torch.manual_seed(seed)
    out, fus, pack = m(image, audio, None, False)
    output = out[:B] + out[B:] * 0.0
    loss = F.cross_entropy(output, label, ignore_index=255)
    loss.backward()
    store = {"loss": np.array([loss.item()], dtype=np.float64), "drop_scales": torch.stack(drawn).numpy(),
             "seed": np.array([seed], dtype=np.int64)}
    gn = {k: p.grad.double().norm().item() for k, p in m.named_parameters() if p.grad is not None}
    store["grad_norm_keys"] = np.array(sorted(gn), dtype=object)
    store["grad_norm_vals"] = np.array([gn[k] for k in sorted(gn)], dtype=np.float64)
    params = dict(m.named_parameters())
    for k in PVT_SENTINELS:
        s_, c_ = sample(params[k].grad)
        store["grad_sample/" + k], store["grad_cksum/" + k] = s_, c_
    taps.update(out_pred=out, out_fusion=fus, pack_visual=pack["visual"], pack_attn_v=pack["attn_v"])
    for k, t in taps.items():
        s_, c_ = sample(t)
        store["sample/" + k], store["cksum/" + k] = s_, c_
        store["shape/" + k] = np.array(t.shape, dtype=np.int64)
    store["cfg/CBHW"] = np.array([C, B, hw[0], hw[1]], dtype=np.int64)
    store["cfg/lds"] = np.array([0, 0, 0], dtype=np.int64)
    store["cfg/train"] = np.array([1], dtype=np.int64)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e3:.0f} kB); loss {loss.item():.5f}; {len(drawn)} DropPath draws, "
          f"{int((torch.stack(drawn) == 0).sum())} dropped; |out|max={out.abs().max().item():.3f}")


def contrast_inputs(seed=7, B=2, C=304, hw=(56, 56), full=(224, 224), num_classes=4):
    """Synthetic ContrastLoss inputs: blocky label maps so that several classes keep >= max_views pixels at 56x56."""
    g = torch.Generator().manual_seed(seed)
    em = torch.randn((B, C) + hw, generator=g)
    es = em * 0.5 + torch.randn((B, C) + hw, generator=g)
    gt = torch.zeros((B,) + full, dtype=torch.long)
    for b in range(B):
        for k in range(1, num_classes):
            h0 = int(torch.randint(0, full[0] - 100, (1,), generator=g)); w0 = int(torch.randint(0, full[1] - 120, (1,), generator=g))
            gt[b, h0:h0 + 60 + 20 * k, w0:w0 + 120] = k
        gt[b, :8, :] = 255
    gs = gt.clone()
    gs[1:] = 0                       # image >= 1: shuffled audio does not match -> background (trainer :176-179)
    return em, gt, es, gs


def run_contrast(out_dir):
    from loss.contrastive_aud import ContrastLoss
    em, gt, es, gs = contrast_inputs()
    em.requires_grad_(True); es.requires_grad_(True)
    crit = ContrastLoss(temperature=0.1, ignore_idx=255, max_views=512)
    torch.manual_seed(1234)
    loss = crit(em, gt, es, gs)
    loss.backward()
    store = {"loss": np.array([loss.item()], dtype=np.float64)}
    for k, t in (("d_match", em.grad), ("d_shuffle", es.grad)):
        s_, c_ = sample(t)
        store["sample/" + k], store["cksum/" + k] = s_, c_
        nz = t.abs().sum(1).flatten().nonzero().flatten()
        store["nnz_pixels/" + k] = np.array([nz.numel()], dtype=np.int64)
    path = os.path.join(out_dir, "contrast.npz")
    np.savez_compressed(path, **store)
    print(f"contrast: wrote {path}; loss={loss.item():.6f}; |d_match|={em.grad.norm().item():.4e} |d_shuffle|={es.grad.norm().item():.4e}")


def run_optstep(out_dir, steps=2, hw=(96, 96), B=4, C=3):
    """Harness row of SURVEY.md §8c: (inputs, labels, seed) -> loss, and the sentinel WEIGHTS after each optimiser step.
    Uses the reference's own `engine.utils.group_weight` and `engine.lr_policy.WarmUpPolyLR`; `set_group_lr` and the
    optimiser construction are restated from main_vpo_mono.py:45-65,118-125 (that file drags in the whole trainer), the
    learning-rate update from trainer_cavp_vpo_mono.py:73-85; hyper-parameters = config/config_avss_binary.py:52-57."""
    from engine.lr_policy import WarmUpPolyLR
    from engine.utils import group_weight
    hyp = EasyDict(lr=1e-3, lr_power=0.9, momentum=0.9, weight_decay=1e-4, use_baseline=False)
    args = EasyDict(seg_model="DeepLabV3Plus", last_three_dilation_stride=[False, False, False], audio_backbone="vgg",
                    num_classes=C, batch_size=B, local_rank="cpu")
    m = CAVP(50, None, num_classes=C, audio_backbone_pretrain_path=None, visual_backbone=50, args=args)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    groups = []
    groups = group_weight(groups, m.backbone, norm_layer=torch.nn.BatchNorm2d, lr=hyp.lr)
    groups.append({"params": m.visual_projector.parameters(), "lr": hyp.lr * 1})
    groups.append({"params": m.cross_att.parameters(), "lr": hyp.lr * 1})
    for module in m.segment.business_layer:
        groups = group_weight(groups, module, torch.nn.BatchNorm2d, hyp.lr * 10.0)
    names = {id(p): k for k, p in m.named_parameters()}
    group_names = [[names[id(p)] for p in (list(g["params"]))] for g in groups]
    for g, gn in zip(groups, group_names):   # generators were consumed above
        g["params"] = [dict(m.named_parameters())[k] for k in gn]
    opt_v = torch.optim.SGD(groups, lr=hyp.lr, momentum=hyp.momentum, weight_decay=hyp.weight_decay)
    opt_a = torch.optim.Adam(params=m.audio_backbone.parameters(), lr=hyp.lr)
    total_iters = 50
    sched = WarmUpPolyLR(hyp.lr, hyp.lr_power, total_iters, 0)
    import json
    with open(os.path.join(out_dir, "optim_groups.json"), "w") as f:
        json.dump({"groups": group_names,
                   "lr_mult": [g["lr"] / hyp.lr for g in opt_v.param_groups],
                   "weight_decay": [g["weight_decay"] for g in opt_v.param_groups],
                   "audio": [k for k, _ in m.audio_backbone.named_parameters()]}, f, indent=0)
    image, audio, label = synth_inputs(B, hw, audio_batch=2 * B, num_classes=C, seed=4)
    store = {"cfg/CBHW": np.array([C, B, hw[0], hw[1]], dtype=np.int64), "cfg/total_iters": np.array([total_iters]),
             "cfg/hyp": np.array([hyp.lr, hyp.lr_power, hyp.momentum, hyp.weight_decay], dtype=np.float64)}
    m.train()
    params = dict(m.named_parameters())
    for k in SENTINELS:
        store["w0/" + k], _ = sample(params[k])
    losses, lrs = [], []
    for it in range(steps):
        lr = float(sched.get_lr(it))
        for g in opt_v.param_groups[:4]:            # trainer_cavp_vpo_mono.py:73-79
            g["lr"] = lr
        for g in opt_v.param_groups[4:]:
            g["lr"] = lr * 10.0
        opt_v.zero_grad()
        opt_a.zero_grad()
        out, fus, pack = m(image, audio, None, False)
        output = out[:B] + out[B:] * 0.0
        loss = F.cross_entropy(output, label, ignore_index=255)
        loss.backward()
        opt_v.step()
        opt_a.step()
        losses.append(loss.item())
        lrs.append(lr)
        for k in SENTINELS:
            s, c = sample(params[k])
            store[f"w{it + 1}/" + k], store[f"wck{it + 1}/" + k] = s, c
    store["loss"] = np.array(losses, dtype=np.float64)
    store["lr"] = np.array(lrs, dtype=np.float64)
    path = os.path.join(out_dir, "optstep.npz")
    np.savez_compressed(path, **store)
    print(f"optstep: wrote {path}; losses={losses}; lrs={lrs}; groups={[len(g) for g in group_names]}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    for name, cfg in CASES.items():
        if a.only and name != a.only:
            continue
        run_case(name, cfg, a.out)
    if not a.only or a.only == "c1p_train_b8_f64":
        run_f64_arbiter(a.out)
    if not a.only or a.only == "contrast":
        run_contrast(a.out)
    if not a.only or a.only == "pvt":
        run_pvt(a.out)
    if not a.only or a.only == "pvt512":   # config #4 at its own resolution (config_avss.py:12-13): samples + checksums only
        run_pvt(a.out, hw=(512, 512), name="pvt_eval_512")
    if not a.only or a.only == "pvt_train":
        run_pvt_train(a.out)
    if not a.only or a.only == "optstep":
        run_optstep(a.out)
