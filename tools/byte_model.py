#!/usr/bin/env python3
"""The "perfectly fused" activation-traffic minimum of SURVEY.md section 8d, re-derived by hooks on the REFERENCE model
(authoring container only, like tools/make_golden.py): every conv / linear / pool reads its input once and writes its output
once, BatchNorm / activation / LayerNorm fused into their producers, one extra read per residual join, both interpolates in +
out, weights once per batch.  Prints elements per frame for C1' (the survey's 70.07 M at C = 2 and 73.97 M at C = 71 are the
check of the method) and for C4 (PVTv2-B5, 512 x 512, 71 classes), which BASELINE.md section 2 left "to be derived".

usage: python tools/byte_model.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "_shims"), "/root/reference", REPO]

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import models.visual.backbones.resnet as _R  # noqa: E402

_R.load_model = lambda model, f, is_restore=False: model
from easydict import EasyDict  # noqa: E402
from models.cavp_model import CAVP  # noqa: E402


def count(seg_model, hw, C, B=1, train=False):
    args = EasyDict(seg_model=seg_model, last_three_dilation_stride=[False, False, False], audio_backbone="vgg", num_classes=C,
                    batch_size=B, local_rank="cpu")
    real_load = torch.load

    def fake_load(path, *a, **k):   # cavp_model.py:109 loads "../ckpts/pretrained/pvt_v2_b5.pth": a fresh random state instead
        if "pvt_v2_b5" in str(path):
            from models.visual.backbones.pvt.pvt import pvt_v2_b5
            sd = pvt_v2_b5().state_dict()
            sd["head.weight"], sd["head.bias"] = torch.zeros(1), torch.zeros(1)
            return sd
        return real_load(path, *a, **k)
    torch.load = fake_load
    try:
        m = CAVP(50, None, num_classes=C, audio_backbone_pretrain_path=None, visual_backbone=50, args=args).eval()
    finally:
        torch.load = real_load
    tot = {"io": 0, "residual": 0, "interp": 0}
    leaf = (nn.Conv2d, nn.Linear, nn.MaxPool2d, nn.AvgPool2d, nn.AdaptiveAvgPool2d)

    def io_hook(mod, inp, out):
        tot["io"] += sum(t.numel() for t in inp if torch.is_tensor(t)) + out.numel()

    def res_hook(n):
        def fn(mod, inp, out):
            o = out[0] if isinstance(out, (tuple, list)) else out
            tot["residual"] += n * o.numel()
        return fn
    for mod in m.modules():
        if isinstance(mod, leaf):
            mod.register_forward_hook(io_hook)
        name = type(mod).__name__
        if name == "Bottleneck":
            mod.register_forward_hook(res_hook(1))
        elif name == "Block":          # PVTv2 block / the fusion block: x + attn(..), x + mlp(..)
            mod.register_forward_hook(res_hook(2))
    orig = F.interpolate

    def interp(x, *a, **k):
        y = orig(x, *a, **k)
        tot["interp"] += x.numel() + y.numel()
        return y
    F.interpolate = interp
    try:
        with torch.no_grad():
            if train:   # forward_train: B images + 2B audio clips, the fusion block and the head run on 2B (cavp_model.py:175-188)
                m.train()
                m(torch.randn(B, 3, *hw), torch.rand(2 * B, 1, 96, 64), None, False)
            else:
                m(torch.randn(B, 3, *hw), torch.rand(B, 1, 96, 64), eval_mode=True)
    finally:
        F.interpolate = orig
    weights = sum(p.numel() for p in m.parameters())
    return {k: v / B for k, v in tot.items()}, weights


if __name__ == "__main__":
    for name, seg, hw, C in (("C1' (C = 2)", "DeepLabV3Plus", (224, 224), 2), ("C1' (C = 71)", "DeepLabV3Plus", (224, 224), 71),
                             ("C4 (PVTv2-B5, 512 x 512, C = 71)", "PVT", (512, 512), 71)):
        tt, _ = count(seg, hw, C, B=2, train=True)
        t, w = count(seg, hw, C)
        act = sum(t.values())
        print(f"{name}: train-mode forward {sum(tt.values()) / 1e6:.2f} M elem/frame = {sum(tt.values()) / act:.3f} x the eval forward")
        print(f"{name}: activations {act / 1e6:.2f} M elem/frame  (conv/linear/pool in+out {t['io'] / 1e6:.2f}, residual reads "
              f"{t['residual'] / 1e6:.2f}, interpolates {t['interp'] / 1e6:.2f}); weights {w / 1e6:.2f} M elem/batch; "
              f"bf16 @B=32: {(act + w / 32) * 2 / 1e6:.1f} MB/frame, @B=8: {(act + w / 8) * 2 / 1e6:.1f} MB/frame")
