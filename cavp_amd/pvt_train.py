"""Training pass of the PVTv2-B5 backbone (reference models/visual/backbones/pvt/pvt.py, seg_model="PVT", config #4) on
the TrainPass tape (cavp_amd/train.py): forward_features with every activation the backward needs kept, and a hand-written
backward through the spatial-reduction attention (csrc/pvt_train.hip), the depth-wise conv MLP, the overlapping patch
embeddings, the LayerNorms and timm's DropPath (stochastic depth, pvt.py:143-144,167-168).

Differences from the eval forward (cavp_amd/pvt.py): the spatial-reduction conv (kernel = stride = sr) runs as a token GEMM
over a space-to-depth rearrangement of the normalised tokens, so that its data and weight gradients are plain GEMMs as
well; the depth-wise conv applies the GELU itself and stores gelu' for the backward (_dwconv_gelu)."""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from . import train_ops as T
from ._lib import ACT_GELU, ACT_NONE, CavpError
from .train import V, TrainPass, _P, _as4


def _pack_sr(tp: TrainPass, key: str, conv, pre_jobs: list) -> _P:
    """The sr x sr / stride-sr conv as a Linear over [B, N/sr^2, sr*sr*C] patches: weight rows in OHWI order."""
    p = _P()
    p.weight, p.bias = conv.weight, conv.bias
    p.real_weight = p.real_bias = None
    p.real_cout = 0
    p.kh = p.kw = 1
    p.stride, p.pad, p.dil = 1, 0, 1
    p.cout, p.cin = conv.out_channels, conv.in_channels * conv.kernel_size[0] * conv.kernel_size[1]
    w_ohwi = torch.empty((p.cout, p.cin), dtype=torch.float32, device=tp.dev)
    pre_jobs.append((conv.weight, w_ohwi, None))   # OIHW -> OHWI in f32: the pack kernel with an f32 destination is that permutation
    p.w = torch.empty((p.cout, 1, 1, p.cin), dtype=tp.dt, device=tp.dev)
    p.wT = torch.empty((p.cin, 1, 1, p.cout), dtype=tp.dt, device=tp.dev)
    tp._pack_jobs.append((w_ohwi, p.w, p.wT))
    tp.P[key] = p
    return p


def _sr_linear(tp: TrainPass, x: V, p: _P) -> V:
    """y = patches @ W^T + b; the weight gradient lands in OHWI order and is un-permuted into the conv's OIHW gradient."""
    y = V(tp.empty(x.t.shape[:-1] + (p.cout,)))
    ops.conv2d(_as4(x.t), p.w, _as4(y.t), shift=p.bias.detach())

    def bwd():
        g = y.g
        if g is None:
            return
        tmp = torch.empty((p.cout, 1, 1, p.cin), dtype=torch.float32, device=tp.dev)
        gw = tp.grad_buffer(p.weight)
        tp.defer_wgrad(dict(x=_as4(x.t), dy=_as4(g), dw=tmp, kh=1, kw=1, stride=1, pad=0, dil=1, dbias=tp.grad_buffer(p.bias),
                            overwrite=True),
                       after=lambda: T.unpack_weight_grad(tmp, gw, accumulate=True))   # OHWI GEMM result -> the conv's OIHW .grad
        if x.needs_grad:
            def dg(o, r, mul=None):
                T.conv2d_dgrad(_as4(g), p.wT, _as4(o), kh=1, kw=1, stride=1, pad=0, dil=1, residual=_as4(r) if r is not None else None)
            tp.acc(x, dg)
    tp.tape.append(bwd)
    return y


def _space_to_depth(tp: TrainPass, x: V, B: int, H: int, W: int, C: int, s: int) -> V:
    y = V(tp.empty((B, (H // s) * (W // s), s * s * C)))
    T.space_to_depth(x.t, y.t, B, H, W, C, s)

    def bwd():
        if y.g is None:
            return
        dx = tp.empty(x.t.shape)
        T.space_to_depth(y.g, dx, B, H, W, C, s, inverse=True)
        tp.acc_add(x, dx)
    tp.tape.append(bwd)
    return y


def _patch_embed0(tp: TrainPass, image: torch.Tensor, conv) -> V:
    """OverlapPatchEmbed.proj of stage 1 (7 x 7, stride 4, pad 3, 3 input channels; pvt.py:187-188).  No input gradient."""
    B, _, H, W = image.shape
    ks, st, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    y = V(tp.empty((B, (H + 2 * pad - ks) // st + 1, (W + 2 * pad - ks) // st + 1, conv.out_channels)))
    ops.conv_smallcin_kxk(image, conv.weight.detach(), conv.bias.detach(), y.t, ks, st, pad)

    def bwd():
        if y.g is None:
            return
        g = y.g if y.g.is_contiguous() else tp._dense_copy(y.g)
        T.colsum(g, tp.grad_buffer(conv.bias))
        T.conv_smallcin_kxk_wgrad(image, g, tp.grad_buffer(conv.weight), ks, st, pad)
    tp.tape.append(bwd)
    return y


def _sra_attention(tp: TrainPass, q: V, kv: V, heads: int, scale: float) -> V:
    o = V(tp.empty(q.t.shape))
    ops.sra_attention(q.t, kv.t, o.t, heads, scale)

    def bwd():
        if o.g is None:
            return
        dq = tp.empty(q.t.shape)
        dkv = tp.empty(kv.t.shape)   # the split partials are f32; their fixed-order sum is stored in the compute dtype
        T.sra_attention_bwd(q.t, kv.t, o.g if o.g.is_contiguous() else tp._dense_copy(o.g), dq, dkv, heads, scale)
        tp.acc_add(q, dq)
        tp.acc_add(kv, dkv)
    tp.tape.append(bwd)
    return o


def _dwconv_gelu(tp: TrainPass, x: V, conv, w9c: torch.Tensor, B: int, H: int, W: int) -> V:
    """GELU(DWConv(x)) (pvt.py:46-55,320-326): depth-wise 3x3 + bias on the tokens viewed as NHWC pixels, GELU in the same
    kernel with gelu'(t) as second output.  The returned activation carries that tensor as `grad_mul`: the data-gradient GEMM
    of fc2 multiplies it into its epilogue (TrainPass.acc), so this op's backward receives d(pre-activation) directly - no
    separate GELU / GELU-backward pass, the pre-activation itself is never stored."""
    hid = x.t.shape[-1]
    y = V(tp.empty(x.t.shape))
    deriv = tp.empty(x.t.shape)
    ops.dwconv3x3(x.t.view(B, H, W, hid), w9c, conv.bias.detach(), y.t.view(B, H, W, hid), act=ACT_GELU,
                  aux=deriv.view(B, H, W, hid))
    y.grad_mul = deriv

    def bwd():
        g = y.g
        if g is None:
            return
        if not y.g_premul:
            raise CavpError("fused GELU: the gradient must come from a data-gradient GEMM (TrainPass.acc)")
        g4 = g.view(B, H, W, hid)
        # weight / bias gradient and data gradient in ONE walk over g (both need its 3 x 3 neighbourhoods)
        dx = tp.empty(x.t.shape) if x.needs_grad else None
        T.dwconv3x3_wgrad(x.t.view(B, H, W, hid), g4, tp.grad_buffer(conv.weight), tp.grad_buffer(conv.bias),
                          w9c if dx is not None else None, dx.view(B, H, W, hid) if dx is not None else None)
        if dx is not None:
            tp.acc_add(x, dx)
    tp.tape.append(bwd)
    return y


def _residual_drop_path(tp: TrainPass, x: V, branch: V, scale: torch.Tensor, ln=None):
    """x + DropPath(branch) with the per-sample factor `scale` = mask / keep_prob (f32 [B]).  With `ln` (the LayerNorm every
    residual output of a PVT block feeds next - norm2, the next block's norm1 or the stage norm) the sum and its normalisation leave
    ONE kernel (cavp_layernorm_residual) and (sum, normalised) are returned."""
    y = V(tp.empty(x.t.shape))
    nrm = None
    if ln is None:
        T.row_scale_add(x.t, branch.t, scale, y.t)
    else:
        nrm = V(tp.empty(x.t.shape))
        ops.layernorm_residual(x.t, branch.t, scale, ln.weight.detach(), ln.bias.detach(), y.t, nrm.t, ln.eps)
    y.dp_scale = scale

    def bwd():
        if y.g is None:
            return
        gb = y.g_scaled          # written by the LayerNorm backward that completed y.g (TrainPass.layernorm), else computed here
        if gb is None:
            gb = tp.empty(branch.t.shape)
            T.row_scale_add(None, y.g, scale, gb)
        tp.acc_add(branch, gb)
        tp.acc_add(x, y.g)
    tp.tape.append(bwd)
    if ln is None:
        return y
    return y, tp.layernorm(y, ln, _done=nrm)


def draw_drop_path_scales(bb, B: int, device, _refresh_only: bool = False) -> List[Optional[torch.Tensor]]:
    """One f32 [B] factor (mask / keep) per residual branch, in forward order (attention, MLP of every block); None where
    the block's probability is 0 or the backbone is in eval mode.  Drawn like timm 0.4.9's drop_path - `floor(keep +
    torch.rand((B, 1, 1)))`, one draw per branch from the default CPU generator - then moved in ONE copy into a persistent
    device buffer (`bb._dp_buf`); an eager pass works on a private clone of it.  While a hipGraph is being captured nothing is drawn: the graph reads that buffer, and
    `CAVP.capture_train_step`'s replay() refreshes it before every launch (refresh_drop_path)."""
    probs = [blk.drop_prob for i in range(4) for blk in getattr(bb, f"block{i + 1}") for _ in range(2)]
    if not bb.training or not any(probs):
        return [None] * len(probs)
    n = sum(1 for p in probs if p > 0)
    buf = getattr(bb, "_dp_buf", None)
    dev = torch.device(device)
    if buf is None or tuple(buf.shape) != (n, B) or buf.device != dev:
        buf = torch.ones((n, B), dtype=torch.float32, device=dev)
        bb.__dict__["_dp_buf"] = buf          # not a registered buffer: it must stay out of the state_dict
    if not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        # one torch.rand((B, 1, 1)) per branch, in forward order (the reference's draws); the arithmetic on all rows at once
        keeps = torch.tensor([1.0 - p for p in probs if p > 0], dtype=torch.float32).view(n, 1)
        rows = torch.stack([torch.rand((B, 1, 1), dtype=torch.float32).view(B) for p in probs if p > 0])
        rows = rows.add_(keeps).floor_().div_(keeps)
        if dev.type == "cuda":
            # through a pinned staging buffer, asynchronously: a pageable-memory copy blocks the host until the stream has drained,
            # which put the host in lock-step with the device and exposed this function's ~1 ms of CPU work in EVERY replay of a
            # captured step (1.1 .. 1.7 ms of idle device between two replays of the PVTv2-B5 step).  The event keeps the host
            # from overwriting the staging buffer before the previous upload has run: at most one step ahead.
            pin, ev = getattr(bb, "_dp_pin", None), getattr(bb, "_dp_ev", None)
            if pin is None or pin.shape != rows.shape:
                pin, ev = torch.empty(rows.shape, dtype=torch.float32).pin_memory(), None
                bb.__dict__["_dp_pin"] = pin
            if ev is not None:
                ev.synchronize()
            pin.copy_(rows)
            buf.copy_(pin, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            bb.__dict__["_dp_ev"] = ev
        else:
            buf.copy_(rows)
        if _refresh_only:
            return []
        # an eager pass gets its OWN copy of the masks: its backward closures may run after a later forward (two forwards before
        # one backward, gradient accumulation with a deferred backward) has redrawn the persistent buffer.  Only a captured
        # graph reads the persistent buffer itself (replay() refreshes it before every launch).
        use = buf.clone()
    else:
        use = buf
    out, k = [], 0
    for p in probs:
        out.append(use[k] if p > 0 else None)
        k += 1 if p > 0 else 0
    return out


def refresh_drop_path(bb, B: int, device) -> None:
    """New DropPath masks for the next replay of a captured training step."""
    draw_drop_path_scales(bb, B, device, _refresh_only=True)


def pvt_train_forward(tp: TrainPass, bb, image: torch.Tensor, drop_scales: Optional[list] = None) -> List[V]:
    """forward_features (pvt.py:291-306) on the tape; returns the 4 stage maps as NHWC V's.  `drop_scales` overrides the
    DropPath factors (tests: the masks a reference run drew)."""
    B = image.shape[0]
    pre_jobs: list = []
    dw_packed: dict = {}
    for i in range(4):
        pe = getattr(bb, f"patch_embed{i + 1}")
        if i > 0:
            tp.pack(f"pvt.pe{i}", pe.proj)
        for j, blk in enumerate(getattr(bb, f"block{i + 1}")):
            k = f"pvt.b{i}.{j}."
            for nme in ("q", "kv", "proj"):
                tp.pack(k + nme, getattr(blk.attn, nme))
            if blk.attn.sr_ratio > 1:
                _pack_sr(tp, k + "sr", blk.attn.sr, pre_jobs)
            tp.pack(k + "fc1", blk.mlp.fc1)
            tp.pack(k + "fc2", blk.mlp.fc2)
            # depth-wise taps [C][1][3][3] -> [9][C] f32: the OIHW -> OHWI permutation of a [1][C][3][3] weight
            dw = blk.mlp.dwconv.dwconv.weight
            w9c = torch.empty((9, dw.shape[0]), dtype=torch.float32, device=tp.dev)
            pre_jobs.append((dw.detach().view(1, dw.shape[0], 3, 3), w9c, None))
            dw_packed[k] = w9c
    T.pack_weights_multi(pre_jobs, torch.float32)
    tp.flush_packs()
    scales = drop_scales if drop_scales is not None else draw_drop_path_scales(bb, B, image.device)
    si = 0
    feats: List[V] = []
    x4: Optional[V] = None
    for i in range(4):
        pe = getattr(bb, f"patch_embed{i + 1}")
        t = _patch_embed0(tp, image, pe.proj) if i == 0 else tp.conv(x4, f"pvt.pe{i}")
        _, H, W, C = t.t.shape
        N = H * W
        x = tp.layernorm(tp.reshape(t, (B, N, C)), pe.norm)
        pend = None   # (x, branch, factor): the stream is x + DropPath(branch), materialised by the NEXT norm's kernel

        def norm_of(ln):
            nonlocal x, pend
            if pend is None:
                return tp.layernorm(x, ln)
            x, nrm = _residual_drop_path(tp, pend[0], pend[1], pend[2], ln=ln)
            pend = None
            return nrm

        for j, blk in enumerate(getattr(bb, f"block{i + 1}")):
            k = f"pvt.b{i}.{j}."
            at = blk.attn
            n1 = norm_of(blk.norm1)
            q = tp.conv(n1, k + "q")
            if at.sr_ratio > 1:
                sr = at.sr_ratio
                if H % sr or W % sr:
                    raise CavpError("PVT spatial-reduction conv needs H, W divisible by sr_ratio")
                xs = tp.layernorm(_sr_linear(tp, _space_to_depth(tp, n1, B, H, W, C, sr), tp.P[k + "sr"]), at.norm)
            else:
                xs = n1
            kv = tp.conv(xs, k + "kv")
            o = _sra_attention(tp, q, kv, at.num_heads, at.scale)
            s_att, s_mlp = scales[si], scales[si + 1]
            si += 2
            if s_att is None:
                x = tp.conv(o, k + "proj", residual=x)
            else:
                pend = (x, tp.conv(o, k + "proj"), s_att)
            n2 = norm_of(blk.norm2)
            h1 = tp.conv(n2, k + "fc1")
            h2 = _dwconv_gelu(tp, h1, blk.mlp.dwconv.dwconv, dw_packed[k], B, H, W)
            if s_mlp is None:
                x = tp.conv(h2, k + "fc2", residual=x)
            else:
                pend = (x, tp.conv(h2, k + "fc2"), s_mlp)
        x = norm_of(getattr(bb, f"norm{i + 1}"))
        x4 = tp.reshape(x, (B, H, W, C))
        feats.append(x4)
    return feats
