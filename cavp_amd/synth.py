"""Deterministic synthetic weights and inputs.

There is no network and no checkpoint on the GPU box, and the reference publishes no test vectors, so every
parity run in this repo uses weights and inputs that are a pure function of (tensor name, shape, seed).  The
same function is used by tools/make_golden.py (which feeds the values to the *reference* model in the
authoring container) and by tests/ + bench.py on the GPU box, so nothing bigger than the outputs has to be
committed.

The rule is keyed on the state_dict key, not on construction order, so it is independent of how the module
tree is built.  Gains are chosen so activations stay O(1) through the 55-conv backbone and logits come out
O(1) (random default init gives |logit| < 0.21, too flat for a 1e-3 test - SURVEY.md §8c).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping, Tuple

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) * 2654435761 + seed * 97) % (2**63 - 1))
    return g


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int, all_keys: Mapping[str, object]) -> torch.Tensor:
    """Value for one state_dict entry.  `all_keys` is only consulted to tell BatchNorm from LayerNorm."""
    g = _gen(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    prefix = name[: -len(leaf) - 1] if "." in name else ""
    shape = tuple(shape)
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        return torch.randn(shape, generator=g) * 0.1
    if leaf == "running_var":
        return torch.rand(shape, generator=g) + 0.5
    if name.startswith("cross_att.pos_embed"):
        # never added on the path (attn.py:235-238) - small non-zero so an accidental use would be visible
        return torch.randn(shape, generator=g) * 0.02
    is_norm = (prefix + ".running_mean") in all_keys or (len(shape) == 1 and "norm" in prefix.rsplit(".", 1)[-1])
    if is_norm:
        if leaf == "weight":
            w = torch.rand(shape, generator=g) + 0.5
            # last BN of each bottleneck feeds the residual sum: damp it so the trunk does not blow up
            if prefix.endswith("bn3") or ".downsample." in prefix + ".":
                w = w * 0.5
            return w
        return torch.randn(shape, generator=g) * 0.1
    if leaf == "bias":
        return torch.randn(shape, generator=g) * 0.05
    # conv / linear weight: He-normal on fan_in
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    gain = math.sqrt(2.0 / max(fan_in, 1))
    if name.endswith("classifier.weight"):
        gain *= 2.0
    return torch.randn(shape, generator=g) * gain


def synth_state_dict(shapes: Mapping[str, Tuple[int, ...]], seed: int = 1) -> Dict[str, torch.Tensor]:
    """shapes: {key: shape} (e.g. {k: v.shape for k, v in model.state_dict().items()})."""
    return {k: synth_tensor(k, tuple(s), seed, shapes) for k, s in shapes.items()}


def synth_inputs(batch: int, image_hw=(224, 224), audio_batch: int | None = None, num_classes: int = 2, seed: int = 0):
    """image ~ N(0,1) (ImageNet-normalised range); audio = log-mel in [-1, 1] (utils/sourcesep.py:38-47 range);
    labels uniform in [0, C) with 2 % ignore_index=255 (loss/losser.py:60)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    ab = batch if audio_batch is None else audio_batch
    image = torch.randn((batch, 3) + tuple(image_hw), generator=g)
    audio = (torch.rand((ab, 1, 96, 64), generator=g) * 2 - 1).clamp_(-1, 1)
    label = torch.randint(0, num_classes, (batch,) + tuple(image_hw), generator=g)
    ign = torch.rand((batch,) + tuple(image_hw), generator=g) < 0.02
    label[ign] = 255
    return image, audio, label


def learnable_inputs(batch: int, image_hw=(96, 96), num_classes: int = 3, seed: int = 0):
    """A synthetic batch a network can LEARN (uniformly random labels, as in synth_inputs, can only be answered with the class
    prior: after a few training steps the logits collapse to a constant): low-frequency random images, label = the quantised
    low-pass of channel 0 (equal-population classes), 2B audio clips as in synth_inputs.  Used to condition the synthetic
    weights by a short training run (tools/conditioned_probe.py, tests/test_gpu_conditioned_parity.py)."""
    H, W = image_hw
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    low = torch.randn(batch, 3, max(1, H // 8), max(1, W // 8), generator=g)
    image = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False) * 1.5 \
        + 0.3 * torch.randn(batch, 3, H, W, generator=g)
    smooth = torch.nn.functional.avg_pool2d(image[:, :1], 9, 1, 4).squeeze(1)
    edges = torch.quantile(smooth.flatten()[:1 << 20], torch.linspace(0, 1, num_classes + 1)[1:-1])
    label = torch.bucketize(smooth, edges).long()
    audio = (torch.rand((2 * batch, 1, 96, 64), generator=g) * 2 - 1).clamp_(-1, 1)
    return image.contiguous(), audio, label.contiguous()
