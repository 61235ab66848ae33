"""Optimiser side of the training harness ("caller/harness rows"), MI355X-native.

The reference drives the model with two torch optimisers (main_vpo_mono.py:45-65,118-125):
  * SGD(momentum, weight_decay) over `set_group_lr(model)`: backbone {decay, no-decay} at lr, visual_projector and
    cross_att (all parameters, default weight decay) at lr, each of the four `segment.business_layer` modules
    {decay, no-decay} at 10 x lr - where `group_weight` (engine/utils.py:642-688) puts conv / linear weights in the decay
    group and their biases plus every norm-layer parameter in the no-decay (weight_decay = 0) group;
  * Adam over `audio_backbone` at the constant base lr;
and a warm-up + polynomial learning-rate schedule (engine/lr_policy.py:30-43, trainer lr_step :73-85).

Because `cavp_amd.cavp_model.CAVP` keeps the reference's module tree, those torch optimisers work on it unchanged.  This
module is the fused alternative for `CAVP.train_step`: every parameter is updated by ONE kernel launch
(`cavp_optimizer_step`) straight from the flat gradient arena, with a device-resident job table built once.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List

import torch
import torch.nn as nn

from . import _lib
from .ops import _ptr, _stream

_NORMS = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.GroupNorm, nn.LayerNorm)   # SyncBatchNorm is a BatchNorm subclass
_CONVS = (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)


def group_weight(groups: List[dict], module: nn.Module, lr: float) -> List[dict]:
    """engine/utils.py:642-688: [weights of conv / linear] at `lr` with the optimiser's weight decay, [their biases +
    all norm-layer parameters] at `lr` with weight_decay = 0.  Raises if a parameter of `module` is in neither set (the
    reference asserts the same)."""
    decay, no_decay = [], []
    for m in module.modules():
        if isinstance(m, (nn.Linear,) + _CONVS):
            decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
        elif isinstance(m, _NORMS):
            if m.weight is not None:
                no_decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
    if len(list(module.parameters())) != len(decay) + len(no_decay):
        raise _lib.CavpError("group_weight: a parameter outside Linear / Conv / norm layers (engine/utils.py:685)")
    groups.append(dict(params=decay, lr=lr))
    groups.append(dict(params=no_decay, weight_decay=0.0, lr=lr))
    return groups


def set_group_lr(model, lr: float, use_baseline: bool = False) -> List[dict]:
    """main_vpo_mono.py:45-65.  Group order (the trainer's lr_step indexes it): 0-1 backbone, 2 projector, 3 cross
    attention, 4.. decoder modules at 10 x lr."""
    groups: List[dict] = []
    group_weight(groups, model.backbone, lr)
    if not use_baseline:
        groups.append({"params": list(model.visual_projector.parameters()), "lr": lr})
        groups.append({"params": list(model.cross_att.parameters()), "lr": lr})
    for module in model.segment.business_layer:
        group_weight(groups, module, lr * 10.0)
    return groups


def warmup_poly_lr(start_lr: float, lr_power: float, total_iters: int, warmup_steps: int = 0,
                   end_lr: float = 1e-8) -> Callable[[int], float]:
    """engine/lr_policy.py:30-43 (WarmUpPolyLR.get_lr)."""
    total = float(total_iters)

    def get_lr(cur_iter: int) -> float:
        if cur_iter < warmup_steps:
            return start_lr * (cur_iter / warmup_steps)
        lr = start_lr * ((1.0 - float(cur_iter) / total) ** lr_power)
        return float(min(max(lr, end_lr), start_lr))
    return get_lr


class OptJob(C.Structure):
    """struct cavp_opt_job (include/cavp_hip.h)."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64),
                ("lr_mult", C.c_float), ("weight_decay", C.c_float), ("blk0", C.c_int32), ("kind", C.c_int32),
                ("vec", C.c_int32), ("pad_", C.c_int32)]


class FusedSGDAdam:
    """SGD(momentum, weight_decay) on `set_group_lr(model)` + Adam(lr = base lr) on `model.audio_backbone`, fused.

    `arena` is the model's flat gradient arena (`CAVP.train_step` creates it; gradients are its views).  `step(lr)`
    takes the current learning rate of the poly schedule: the visual groups use lr x {1, 10} (trainer lr_step), the
    audio Adam keeps the constant base lr (trainer_cavp_vpo_mono.py:84 only logs it)."""

    def __init__(self, model, arena, lr: float, momentum: float = 0.9, weight_decay: float = 1e-4,
                 use_baseline: bool = False, betas=(0.9, 0.999), eps: float = 1e-8):
        self.base_lr, self.momentum, self.betas, self.eps = float(lr), float(momentum), betas, float(eps)
        self.steps = 0
        import weakref
        self._model = weakref.ref(model) if hasattr(model, "params_changed") else (lambda: None)
        dev = arena.flat.device
        specs = []   # (param, kind, lr_mult, wd)
        for g in set_group_lr(model, 1.0, use_baseline):
            for p in g["params"]:
                specs.append((p, 0, float(g["lr"]), float(g.get("weight_decay", weight_decay))))
        for p in model.audio_backbone.parameters():
            specs.append((p, 1, 1.0, 0.0))                       # torch.optim.Adam default weight_decay = 0
        seen = set()
        for p, *_ in specs:
            if id(p) in seen:
                raise _lib.CavpError("FusedSGDAdam: a parameter appears in two groups")
            seen.add(id(p))
        # torch.optim skips parameters whose .grad is None: cross_att.pos_embed_v / pos_embed_a and audio_backbone.cls_head never
        # receive a gradient on this path (cavp_model.py / attn.py:235-238), so they get neither weight decay nor momentum
        never = {id(p) for p in getattr(model, "params_without_grad", lambda: [])()}
        specs = [s for s in specs if s[0].requires_grad and id(s[0]) in arena.views and id(s[0]) not in never]
        # state: one flat buffer for momentum / first moments, one for Adam's second moments
        offs, tot = [], 0
        for p, *_ in specs:
            offs.append(tot)
            tot += (p.numel() + 3) // 4 * 4
        self.state_m = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.state_v = torch.zeros(tot, dtype=torch.float32, device=dev)
        lib = _lib.load()
        jobs = (OptJob * len(specs))()
        blk = 0
        self._keep = []
        for i, ((p, kind, lr_mult, wd), off) in enumerate(zip(specs, offs)):
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise _lib.CavpError("FusedSGDAdam: contiguous f32 parameters on the arena's device required")
            g = arena.views[id(p)]
            m = self.state_m[off:off + p.numel()]
            v = self.state_v[off:off + p.numel()]
            vec = int(all(t.data_ptr() % 16 == 0 for t in (p, g, m)))
            jobs[i] = OptJob(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr_mult, wd, blk, kind, vec, 0)
            blk += lib.cavp_optimizer_blocks(p.numel())
            self._keep.append(p)
        self.njobs, self.total_blocks = len(specs), blk
        raw = bytes(jobs)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)   # device-resident job table
        self.params = [s[0] for s in specs]

    def step(self, lr: float) -> None:
        self.steps += 1
        if self._model() is not None:
            self._model().params_changed()   # weights change through raw pointers: no tensor version is bumped
        b1, b2 = self.betas
        st = _lib.load().cavp_optimizer_step(_ptr(self.table), self.njobs, self.total_blocks, C.c_float(lr),
                                             C.c_float(self.base_lr), C.c_float(self.momentum), C.c_float(b1),
                                             C.c_float(b2), C.c_float(self.eps), C.c_int64(self.steps),
                                             C.c_void_p(_stream()))
        _lib.check(st, "cavp_optimizer_step")

    def state_dict(self) -> Dict[str, object]:
        return {"steps": self.steps, "m": self.state_m.clone(), "v": self.state_v.clone()}

    def load_state_dict(self, sd) -> None:
        self.steps = int(sd["steps"])
        self.state_m.copy_(sd["m"])
        self.state_v.copy_(sd["v"])
