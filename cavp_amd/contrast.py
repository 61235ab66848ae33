"""MI355X path of the reference's live contrastive loss, `loss/contrastive_aud.py::ContrastLoss` (SURVEY.md §8a row a13,
config #5).  Same constructor and `forward(embeds_match, gt_match, embeds_shuffle, gt_shuffle)`.

Split of work:
  host   - nearest-downsampling of the label maps and the class-balanced sampling (contrastive_aud.py:18-22,76-141):
           pure index bookkeeping on the (small) label tensors, done on the CPU with the SAME sequence of
           `torch.randperm` calls on the default CPU generator as the reference, so the sampled anchors are identical
           for an identical RNG state;
  device - L2-normalise + gather of the N anchors, S = A A^T / T on the f32 MFMA igemm, the row-wise InfoNCE, and the
           whole backward (dS, (dS + dS^T) A via the wgrad GEMM, normalisation backward, scatter into the feature
           gradient) - libcavp_hip.so only, no torch arithmetic.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from . import train_ops as T
from .ops import _ptr, _stream


def nearest_indices(n_in: int, n_out: int) -> np.ndarray:
    """F.interpolate(mode='nearest') source index: min(floor(dst * float32(in / out)), in - 1)."""
    scale = np.float32(n_in) / np.float32(n_out)
    return np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64), n_in - 1)


# Host copies of down-sampled label maps, keyed by the label tensor's identity (+ version counter) and the target size.  The class-balanced
# sampling is index bookkeeping on the labels and runs on the host with the reference's own torch.randperm sequence; its only
# device dependency is this download.  Read inside the loss call it is a device-to-host copy on the launching stream: the host
# then waits for the whole forward pass that was queued in front of it, and the device idles while the host samples and launches
# (2.3 ms of an 18 ms config-#5 step in round 3).  Two ways out, both kept:
#   * `ContrastLoss.prefetch_labels(gt_match, gt_shuffle, size)` right after the batch reaches the device (before the model's
#     forward): the reduction + download run there, behind nothing, and the loss call finds the copy here;
#   * a label tensor OBJECT that was already downloaded and has not been written since (same object, same version counter) is
#     not downloaded again (validation-style loops over fixed batches, bench.py's synthetic step).
#     CONTRACT: the hit test is identity + torch's version counter, so it sees torch writes only.  A label buffer refilled behind
#     torch's back (a raw-pointer kernel, .data, a DLPack / numpy alias) must be followed by
#     invalidate_label_cache() before the next loss call (the returned arrays are read-only: callers never write into a cached copy).
_LABEL_CACHE: dict = {}
_LABEL_CACHE_MAX = 8


def invalidate_label_cache() -> None:
    """Forget every cached label download (see the contract above)."""
    _LABEL_CACHE.clear()


def _label_key(gt: torch.Tensor, size):
    return (id(gt), tuple(size))


def downsample_labels(gt: torch.Tensor, size: Tuple[int, int]) -> np.ndarray:
    """[B, H, W] int labels -> [B, h*w] (nearest).  Device labels are reduced ON the device (cavp_label_nearest) and only the
    B*h*w int32 result is downloaded for the host-side sampling (a 224 x 224 int64 batch is 16 x larger, a 512 x 512 one 84 x)."""
    if gt.is_cuda and gt.dim() == 3:
        # a hit needs the SAME tensor object (held through a weak reference: a new tensor the allocator placed at the old address
        # is another object), unwritten since (version counter)
        key = _label_key(gt, size)
        hit = _LABEL_CACHE.get(key)
        if hit is not None and hit[0]() is gt and hit[1] == gt._version:
            return hit[2]
        out = _downsample_labels_device(gt, size)
        out.setflags(write=False)   # the cached array is handed out by reference: a caller that wrote into it would poison later hits
        for k in [k for k, v in _LABEL_CACHE.items() if v[0]() is None]:
            del _LABEL_CACHE[k]
        if len(_LABEL_CACHE) >= _LABEL_CACHE_MAX:
            _LABEL_CACHE.pop(next(iter(_LABEL_CACHE)))
        _LABEL_CACHE[key] = (weakref.ref(gt), gt._version, out)
        return out
    g = gt.detach().cpu().numpy()
    hi, wi = nearest_indices(g.shape[1], size[0]), nearest_indices(g.shape[2], size[1])
    return g[:, hi][:, :, wi].reshape(g.shape[0], -1)


def _downsample_labels_device(gt: torch.Tensor, size: Tuple[int, int]) -> np.ndarray:
    if True:
        g = gt.detach()
        if g.dtype != torch.int64 or not g.is_contiguous():
            g = g.to(torch.int64).contiguous()
        out = torch.empty((g.shape[0], size[0], size[1]), dtype=torch.int32, device=g.device)
        _lib.check(_lib.load().cavp_label_nearest(_ptr(g), _ptr(out), g.shape[0], g.shape[1], g.shape[2], size[0], size[1],
                                                  C.c_void_p(_stream())), "cavp_label_nearest")
        return out.cpu().numpy().astype(np.int64).reshape(g.shape[0], -1)


_PINNED: dict = {}


def _upload_i32(arrs, dev):
    """Several small int32 host arrays -> one device tensor through a PINNED staging buffer (a pageable torch.from_numpy().to(dev)
    is a blocking copy each); returns the device views.  One staging buffer per call slot (two rotate) so that a copy still in
    flight is not overwritten by the next step's indices."""
    n = sum(int(a.size) for a in arrs)
    # staging slots per (device, thread): nn.DataParallel replica threads and loss calls on different devices must not share one
    # (the event is recorded after the host writes, so a second user of the slot could overwrite a buffer whose copy is pending)
    import threading
    own = _PINNED.setdefault((torch.device(dev).index or 0, threading.get_ident()), {})
    slot = own.get("slot", 0)
    own["slot"] = slot ^ 1
    buf, ev = own.get(slot, (None, None))
    if ev is not None:
        ev.synchronize()       # the copy that last used this staging buffer has run (two steps ago: normally long done)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1 << 14), dtype=torch.int32).pin_memory()
    views, o = [], 0
    host = buf.numpy()
    for a in arrs:
        host[o:o + a.size] = a.reshape(-1)
        o += a.size
    d = buf[:n].to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    own[slot] = (buf, ev)
    o = 0
    for a in arrs:
        views.append(d[o:o + a.size])
        o += a.size
    return views


class SamplePlan:
    """Anchors chosen by `extraction_samples`: first `n_match` rows come from embeds_match, the rest from embeds_shuffle."""
    __slots__ = ("b", "p", "labels", "n_match", "n")

    def __init__(self, b, p, labels, n_match):
        self.b, self.p, self.labels, self.n_match, self.n = b, p, labels, n_match, len(b)


def sample_anchors(gt_match: np.ndarray, gt_shuffle: np.ndarray, ignore_idx: int, max_views: int) -> Optional[SamplePlan]:
    """contrastive_aud.py:76-141 on [B, hw] label arrays; consumes torch.randperm exactly like the reference."""
    B, HW = gt_match.shape
    # flat pixel indices throughout; (image, pixel) only for the few hundred anchors that are kept (the first version split all
    # B * hw indices and gathered three B * hw-sized arrays per call: 4 of the 6 ms this function cost per step at B = 30)
    gm = gt_match.reshape(-1)
    fg_idx = np.flatnonzero((gm > 0) & (gm != ignore_idx))
    fg_l = gm[fg_idx]
    sel: List[np.ndarray] = []
    sel_l: List[np.ndarray] = []
    for item in np.unique(fg_l):                       # torch.unique: sorted ascending
        cur = np.flatnonzero(fg_l == item)
        if cur.shape[0] < max_views:
            continue
        r = torch.randperm(cur.shape[0]).numpy()[:max_views]
        sel.append(fg_idx[cur[r]]); sel_l.append(fg_l[cur[r]])
    if not sel:
        return None
    bg_idx = np.flatnonzero(gm == 0)
    sample_num = int(min(max_views, fg_idx.shape[0], bg_idx.shape[0]))
    i1 = torch.randperm(bg_idx.shape[0]).numpy()[:sample_num]
    i2 = torch.randperm(fg_idx.shape[0]).numpy()[:sample_num]
    idx = np.concatenate(sel + [bg_idx[i1], fg_idx[i2]])
    # shuffle-branch candidates live at the MATCH foreground pixels
    lab = np.concatenate(sel_l + [np.zeros(sample_num, dtype=gm.dtype), gt_shuffle.reshape(-1)[fg_idx[i2]]])
    b, p = np.divmod(idx, HW)
    return SamplePlan(b.astype(np.int32), p.astype(np.int32), lab.astype(np.int32), len(idx) - sample_num)


def _strides_bcp(x: torch.Tensor) -> Tuple[int, int, int]:
    """element strides (batch, channel, pixel) of a [B, C, H, W] tensor whose (H, W) plane is a uniform pixel grid."""
    sb, sc, sh, sw = x.stride()
    if x.shape[3] > 1 and x.shape[2] > 1 and sh != sw * x.shape[3]:
        raise _lib.CavpError("feature map must have a uniform pixel stride (NCHW-contiguous or channels-last)")
    return sb, sc, sw


class _InfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, em, es, plan: SamplePlan, temperature: float, eps: float):
        lib = _lib.load()
        dev = em.device
        st = C.c_void_p(_stream())
        Cc = em.shape[1]
        n, npad = plan.n, (plan.n + 3) // 4 * 4
        ib, ip, lab = _upload_i32((plan.b, plan.p, plan.labels), dev)
        A = torch.zeros((npad, Cc), dtype=torch.float32, device=dev)
        norms = torch.empty(npad, dtype=torch.float32, device=dev)
        for x, lo, hi in ((em, 0, plan.n_match), (es, plan.n_match, n)):
            if hi > lo:
                sb, sc, sp = _strides_bcp(x)
                _lib.check(lib.cavp_gather_l2norm(_ptr(x), sb, sc, sp, _ptr(ib[lo:]), _ptr(ip[lo:]), hi - lo, Cc,
                                                  C.c_float(1e-12), _ptr(A[lo:]), _ptr(norms[lo:]), st), "cavp_gather_l2norm")
        S = torch.empty((npad, npad), dtype=torch.float32, device=dev)
        inv_t = torch.full((npad,), 1.0 / temperature, dtype=torch.float32, device=dev)
        ops.linear(A, A.view(npad, 1, 1, Cc), S, scale=inv_t)          # S = A A^T / T on the f32 MFMA path
        rows = torch.empty(npad, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        need_grad = em.requires_grad or es.requires_grad
        dS = torch.empty_like(S) if need_grad else None
        _lib.check(lib.cavp_infonce_rows(_ptr(S), _ptr(lab), n, npad, C.c_float(eps), _ptr(rows), _ptr(loss), _ptr(dS),
                                         C.c_float(1.0), st), "cavp_infonce_rows")
        ctx.saved = (A, norms, dS, ib, ip, plan, em, es, temperature)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        A, norms, dS, ib, ip, plan, em, es, temperature = ctx.saved
        st = C.c_void_p(_stream())
        npad, Cc = A.shape
        n = plan.n
        G = torch.empty_like(dS)
        # dL/dA = (dS + dS^T) A / T  (anchors and contrasts are the same tensor), scaled by the incoming gradient
        # (the upstream gradient is a device scalar: it is multiplied in on the device - float(gout) made the host wait for the
        # whole forward + loss queue before it could launch the backward)
        gs = gout.detach().reshape(1).to(torch.float32)
        _lib.check(lib.cavp_symm_add_scaled(_ptr(dS), _ptr(G), npad, C.c_float(1.0 / temperature), _ptr(gs), st), "cavp_symm_add_scaled")
        dA = torch.zeros((npad, Cc), dtype=torch.float32, device=A.device)
        T.linear_wgrad(A, G, dA)
        grads = []
        for x, lo, hi in ((em, 0, plan.n_match), (es, plan.n_match, n)):
            b, c, h, w = x.shape
            g = torch.zeros((b, h, w, c), dtype=torch.float32, device=x.device)   # NHWC memory, returned as an NCHW view
            if hi > lo:
                _lib.check(lib.cavp_l2norm_bwd_scatter(_ptr(dA[lo:]), _ptr(A[lo:]), _ptr(norms[lo:]), _ptr(ib[lo:]), _ptr(ip[lo:]),
                                                       hi - lo, Cc, _ptr(g), h * w * c, 1, c, st), "cavp_l2norm_bwd_scatter")
            grads.append(g.permute(0, 3, 1, 2))
        return grads[0], grads[1], None, None, None


class ContrastLoss(nn.Module):
    def __init__(self, temperature, ignore_idx, max_views):
        super().__init__()
        self.ignore_idx = ignore_idx
        self.ood_idx = 254
        self.eps = 1e-12
        self.temperature = temperature
        self.max_views = max_views

    @staticmethod
    def prefetch_labels(gt_match, gt_shuffle, size) -> None:
        """Optional, for trainers: call right after the batch is on the device (before the model's forward) with the spatial size
        of the feature map the loss will see (out_fusion: H/4 x W/4).  The label reduction and its download then do not wait
        behind the forward pass; forward() finds the host copies by the tensors' identity + version."""
        for g in (gt_match, gt_shuffle):
            downsample_labels(g, tuple(size))

    def forward(self, embeds_match, gt_match, embeds_shuffle, gt_shuffle):
        if not embeds_match.is_cuda:
            raise _lib.CavpError("ContrastLoss (MI355X path) needs HIP device tensors: there is no CPU fallback")
        if embeds_match.dtype != torch.float32 or embeds_shuffle.dtype != torch.float32:
            raise _lib.CavpError("ContrastLoss expects the f32 out_fusion features")
        size = tuple(embeds_match.shape[2:])
        plan = sample_anchors(downsample_labels(gt_match, size), downsample_labels(gt_shuffle, size), self.ignore_idx,
                              self.max_views)
        if plan is None:
            return torch.tensor([0.0], device=gt_match.device)          # contrastive_aud.py:35-36
        return _InfoNCEFn.apply(embeds_match, embeds_shuffle, plan, float(self.temperature), float(self.eps))
