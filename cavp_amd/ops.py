"""Thin Python wrappers over the C-ABI (include/cavp_hip.h).  torch is used only for device memory and the current
HIP stream; every arithmetic operation is a kernel in libcavp_hip.so.

Tensors are NHWC "views": a torch tensor of shape [N, H, W, C] whose strides are (H*W*ld, W*ld, ld, 1) with
ld >= C — i.e. possibly a channel slice `buf[..., c0:c1]` of a wider buffer (free concat)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_LEAKY, ACT_NONE, ACT_RELU, BF16, F32, ConvDesc  # noqa: F401

_TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16}
_CODE = {torch.float32: F32, torch.bfloat16: BF16}

_workspace = {}
_retired = []   # outgrown workspaces: never freed, because a captured hipGraph may have baked their address in


def torch_dtype(code: int) -> torch.dtype:
    return _TORCH_DTYPE[code]


def dtype_code(t: torch.dtype) -> int:
    try:
        return _CODE[t]
    except KeyError:
        raise _lib.CavpError(f"unsupported dtype {t}: the HIP path computes in float32 or bfloat16") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.CavpError("libcavp_hip kernels need HIP device tensors; got a CPU tensor (there is no CPU fallback)")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _nhwc(t: torch.Tensor) -> Tuple[int, int, int, int, int]:
    """(N, H, W, C, ld) of an NHWC view; validates the stride pattern."""
    if t.dim() != 4:
        raise _lib.CavpError(f"expected an NHWC 4-d view, got shape {tuple(t.shape)}")
    n, h, w, c = t.shape
    sn, sh, sw, sc = t.stride()
    # size-1 dims carry arbitrary strides: take ld from the innermost spatial dim that actually strides
    if w > 1:
        ld = sw
    elif h > 1:
        ld = sh
    elif n > 1:
        ld = sn
    else:
        ld = c
    if c > 1 and sc != 1:
        raise _lib.CavpError("NHWC view must be dense along channels")
    if (w > 1 and sw != ld) or (h > 1 and sh != w * ld) or (n > 1 and sn != h * w * ld) or ld < c:
        raise _lib.CavpError(f"not a dense NHWC view: shape {tuple(t.shape)} stride {t.stride()}")
    return n, h, w, c, ld


_ws_slot = [0]


class workspace_slot:
    """Kernels launched on a second stream concurrently with the main one (TrainPass side sections) must not share the split-K /
    weight-gradient slab scratch: inside this context `workspace()` hands out slot `k`'s own buffer."""

    def __init__(self, k: int):
        self.k = k

    def __enter__(self):
        self.prev = _ws_slot[0]
        _ws_slot[0] = self.k

    def __exit__(self, *exc):
        _ws_slot[0] = self.prev
        return False


def workspace(nbytes: int, device) -> Optional[torch.Tensor]:
    """Grow-only scratch buffer per device (split-K slabs).  Must not grow while a graph is being captured.  A buffer that
    is outgrown later (say validation at another resolution after capture_train_step) is RETIRED, not freed: graphs captured
    earlier keep replaying into the address they recorded, and the caching allocator must never hand that memory to anyone
    else.  Sizes at least double, so the retired buffers sum to less than the live one."""
    if nbytes <= 0:
        return None
    key = (torch.device(device).index or 0, _ws_slot[0])
    ws = _workspace.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.CavpError("workspace would grow during hipGraph capture; run one eager warm-up pass first")
        if ws is not None:
            _retired.append(ws)
        ws = torch.empty(max(nbytes, 64 << 20, 2 * (ws.numel() if ws is not None else 0)), dtype=torch.uint8, device=device)
        _workspace[key] = ws
    return ws


def conv2d(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, kh: int = 1, kw: int = 1, stride: int = 1,
           pad: int = 0, dil: int = 1, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
           nbias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act: int = ACT_NONE,
           splitk: int = 0, tile: int = 0, want_tile_stats: bool = False, stride_w: int = 0,
           res_rows: int = 0, aux: Optional[torch.Tensor] = None, aux_mode: int = 0):
    """out = act((conv(x, w) + nbias[n]) * scale + shift + residual); x/out/residual NHWC views, w OHWI packed.
    res_rows > 0: `residual` has res_rows pixel rows and output pixel p adds row p % res_rows (a batch-periodic residual).
    aux / aux_mode: 1 = act is GELU and gelu'(pre-activation) is stored into `aux` (out's shape); 2 = the result is multiplied
    by `aux` before the residual is added (include/cavp_hip.h, cavp_conv_desc.aux_mode).
    want_tile_stats=True (plain convs only) returns (out, stats) where stats is None when this launch cannot produce the
    fused BatchNorm statistics, else (tile_stats f32 [tiles][Cout][2], tiles, rows_per_tile)."""
    _need_gpu(x, w, out, scale, shift, nbias, residual)
    lib = _lib.load()
    n, h, wd, cin, ldx = _nhwc(x)
    no, ho, wo, cout, ldy = _nhwc(out)
    dt = dtype_code(x.dtype)
    if w.dtype != x.dtype or out.dtype != x.dtype:
        raise _lib.CavpError("conv2d: x, w and out must share one dtype")
    if w.numel() != cout * kh * kw * cin or not w.is_contiguous():
        raise _lib.CavpError(f"conv2d: packed weight has {w.numel()} elements, expected {cout}x{kh}x{kw}x{cin}")
    eho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    ewo = (wd + 2 * pad - dil * (kw - 1) - 1) // (stride_w or stride) + 1
    if (no, ho, wo) != (n, eho, ewo):
        raise _lib.CavpError(f"conv2d: out view is {(no, ho, wo)}, expected {(n, eho, ewo)}")
    ldr = 0
    if residual is not None:
        rn, rh, rw, rc, ldr = _nhwc(residual)
        if res_rows:
            if rn * rh * rw != res_rows or rc != cout or residual.dtype != x.dtype or (n * eho * ewo) % res_rows:
                raise _lib.CavpError("conv2d: a periodic residual must have res_rows rows dividing the output's")
        elif (rn, rh, rw, rc) != (n, eho, ewo, cout) or residual.dtype != x.dtype:
            raise _lib.CavpError("conv2d: residual must match the output view")
    ld_aux = 0
    if aux_mode:
        an, ah, aw, ac, ld_aux = _nhwc(aux)
        if (an, ah, aw, ac) != (n, eho, ewo, cout) or aux.dtype != x.dtype:
            raise _lib.CavpError("conv2d: aux must match the output view")
    for name, v in (("scale", scale), ("shift", shift)):
        if v is not None and (v.dtype != torch.float32 or v.numel() != cout or not v.is_contiguous()):
            raise _lib.CavpError(f"conv2d: {name} must be a contiguous f32 [{cout}]")
    if nbias is not None and (nbias.dtype != torch.float32 or nbias.numel() != n * cout or not nbias.is_contiguous()):
        raise _lib.CavpError(f"conv2d: nbias must be a contiguous f32 [{n},{cout}]")
    d = ConvDesc(dtype=dt, N=n, H=h, W=wd, Cin=cin, ldx=ldx, Cout=cout, ldy=ldy, KH=kh, KW=kw, stride=stride, pad=pad,
                 dil=dil, ldr=ldr, act=act, splitk=splitk, tile=tile, up=0, Ho=0, Wo=0, stride_w=stride_w,
                 res_rows=res_rows if residual is not None else 0, aux_mode=aux_mode, ld_aux=ld_aux)
    nbytes = lib.cavp_conv2d_workspace_bytes(C.byref(d))
    ws = workspace(nbytes, x.device)
    stats = None
    if want_tile_stats:
        tiles, rpt = C.c_int32(0), C.c_int32(0)
        if lib.cavp_conv2d_tile_stats_layout(C.byref(d), C.byref(tiles), C.byref(rpt)) and out.data_ptr() % 16 == 0:
            stats = (torch.empty((tiles.value, cout, 2), dtype=torch.float32, device=x.device), tiles.value, rpt.value)
    st = lib.cavp_conv2d_nhwc_aux(C.byref(d), _ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(nbias), _ptr(residual),
                                  _ptr(out), _ptr(aux) if aux_mode else None, _ptr(ws),
                                  C.c_size_t(ws.numel() if ws is not None else 0),
                                  _ptr(stats[0]) if stats is not None else None, C.c_void_p(_stream()))
    _lib.check(st, f"cavp_conv2d_nhwc N{n} H{h} W{wd} Cin{cin} Cout{cout} k{kh}x{kw} s{stride} p{pad} d{dil}")
    if want_tile_stats:
        return out, stats
    return out


def linear(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias=None, scale=None, residual=None,
           nbias=None, act: int = ACT_NONE, splitk: int = 0, tile: int = 0, res_rows: int = 0) -> torch.Tensor:
    """x: [B, T, Cin] (or [B, Cin]) view, out: [B, T, Cout].  Linear = 1x1 conv over N=B, H=1, W=T."""
    def as4(t):
        if t is None:
            return None
        if t.dim() == 2:
            return t.unsqueeze(1).unsqueeze(1)
        if t.dim() == 3:
            return t.unsqueeze(1)
        return t
    conv2d(as4(x), w, as4(out), scale=scale, shift=bias, nbias=nbias, residual=as4(residual), act=act,
           splitk=splitk, tile=tile, res_rows=res_rows)
    return out


def conv3x3_smallcin_nchw(x_nchw: torch.Tensor, w_oihw: torch.Tensor, out: torch.Tensor, *, stride: int, scale=None,
                          shift=None, act: int = ACT_NONE) -> torch.Tensor:
    _need_gpu(x_nchw, w_oihw, out, scale, shift)
    lib = _lib.load()
    if x_nchw.dtype != torch.float32 or w_oihw.dtype != torch.float32 or not x_nchw.is_contiguous() or not w_oihw.is_contiguous():
        raise _lib.CavpError("conv3x3_smallcin_nchw: x and w must be contiguous f32 (NCHW / OIHW)")
    n, cin, h, w = x_nchw.shape
    cout = w_oihw.shape[0]
    no, ho, wo, co, ldy = _nhwc(out)
    if tuple(w_oihw.shape) != (cout, cin, 3, 3) or co != cout or ldy != cout:
        raise _lib.CavpError("conv3x3_smallcin_nchw: shape mismatch")
    if (no, ho, wo) != (n, (h - 1) // stride + 1, (w - 1) // stride + 1):
        raise _lib.CavpError("conv3x3_smallcin_nchw: bad output view")
    st = lib.cavp_conv3x3_smallcin_nchw(dtype_code(out.dtype), _ptr(x_nchw), _ptr(w_oihw), _ptr(scale), _ptr(shift),
                                        _ptr(out), n, cin, h, w, cout, stride, act, C.c_void_p(_stream()))
    _lib.check(st, "cavp_conv3x3_smallcin_nchw")
    return out


def maxpool(x: torch.Tensor, out: torch.Tensor, k: int, stride: int, pad: int,
            argmax: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
            act: int = ACT_NONE) -> torch.Tensor:
    """argmax (optional): uint8 tensor of out's shape receiving the window-relative arg-max (for maxpool_bwd).
    scale / shift (f32 [C], both or neither) + act: the pool runs over act(x * scale + shift) - cavp_maxpool_affine_nhwc."""
    _need_gpu(x, out, scale, shift)
    n, h, w, c, ld = _nhwc(x)
    no, ho, wo, co, ldo = _nhwc(out)
    if ld != c or ldo != co or co != c or out.dtype != x.dtype:
        raise _lib.CavpError("maxpool: dense NHWC tensors of one dtype required")
    if (no, ho, wo) != (n, (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1):
        raise _lib.CavpError("maxpool: bad output shape")
    if argmax is not None and (argmax.dtype != torch.uint8 or not argmax.is_contiguous() or argmax.numel() != out.numel()
                               or argmax.device != x.device):
        raise _lib.CavpError("maxpool: argmax must be a dense uint8 tensor of the output's shape")
    if (scale is None) != (shift is None):
        raise _lib.CavpError("maxpool: scale and shift come together")
    if scale is not None:
        if scale.dtype != torch.float32 or shift.dtype != torch.float32 or scale.numel() != c or shift.numel() != c:
            raise _lib.CavpError("maxpool: scale / shift must be f32 [C]")
        st = _lib.load().cavp_maxpool_affine_nhwc(dtype_code(x.dtype), _ptr(x), _ptr(scale), _ptr(shift), act, _ptr(out), _ptr(argmax),
                                                  n, h, w, c, k, stride, pad, C.c_void_p(_stream()))
        _lib.check(st, "cavp_maxpool_affine_nhwc")
        return out
    st = _lib.load().cavp_maxpool_nhwc(dtype_code(x.dtype), _ptr(x), _ptr(out), _ptr(argmax), n, h, w, c, k, stride, pad,
                                       C.c_void_p(_stream()))
    _lib.check(st, "cavp_maxpool_nhwc")
    return out


def global_avgpool(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _need_gpu(x, out)
    n, h, w, c, ld = _nhwc(x)
    if out.dtype != torch.float32 or out.numel() != n * c or not out.is_contiguous():
        raise _lib.CavpError("global_avgpool: out must be contiguous f32 [N, C]")
    st = _lib.load().cavp_global_avgpool_nhwc(dtype_code(x.dtype), _ptr(x), _ptr(out), n, h * w, c, ld,
                                              C.c_void_p(_stream()))
    _lib.check(st, "cavp_global_avgpool_nhwc")
    return out


def bilinear(x: torch.Tensor, out: torch.Tensor, align_corners: bool) -> torch.Tensor:
    _need_gpu(x, out)
    n, hi, wi, c, ldx = _nhwc(x)
    no, ho, wo, co, ldy = _nhwc(out)
    if no != n or co != c or out.dtype != x.dtype:
        raise _lib.CavpError("bilinear: batch / channel / dtype mismatch")
    st = _lib.load().cavp_bilinear_nhwc(dtype_code(x.dtype), _ptr(x), _ptr(out), n, hi, wi, c, ldx, ho, wo, ldy,
                                        int(align_corners), C.c_void_p(_stream()))
    _lib.check(st, "cavp_bilinear_nhwc")
    return out


def bilinear_to_nchw(x: torch.Tensor, out_nchw: torch.Tensor, align_corners: bool) -> torch.Tensor:
    _need_gpu(x, out_nchw)
    n, hi, wi, c, ldx = _nhwc(x)
    if out_nchw.dtype != torch.float32 or not out_nchw.is_contiguous() or out_nchw.shape[:2] != (n, c):
        raise _lib.CavpError("bilinear_to_nchw: out must be contiguous f32 [N, C, Ho, Wo]")
    ho, wo = out_nchw.shape[-2:]
    st = _lib.load().cavp_bilinear_nhwc_to_nchw(dtype_code(x.dtype), _ptr(x), _ptr(out_nchw), n, hi, wi, c, ldx, ho, wo,
                                                int(align_corners), C.c_void_p(_stream()))
    _lib.check(st, "cavp_bilinear_nhwc_to_nchw")
    return out_nchw


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, eps: float) -> torch.Tensor:
    """x, out: [..., C] row-dense views (last-dim stride 1, uniform row stride)."""
    _need_gpu(x, gamma, beta, out)
    c = x.shape[-1]
    rows = x.numel() // c
    x2, o2 = x.reshape(rows, c), out.reshape(rows, c)
    if x2.data_ptr() != x.data_ptr() or o2.data_ptr() != out.data_ptr() or x2.stride(1) != 1 or o2.stride(1) != 1:
        raise _lib.CavpError("layernorm: rows must be viewable as [rows, C] without a copy")
    if gamma.dtype != torch.float32 or beta.dtype != torch.float32 or out.dtype != x.dtype:
        raise _lib.CavpError("layernorm: gamma/beta f32, out dtype == x dtype")
    st = _lib.load().cavp_layernorm(dtype_code(x.dtype), _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(o2), rows, c,
                                    x2.stride(0) if rows > 1 else c, o2.stride(0) if rows > 1 else c,
                                    C.c_float(eps), C.c_void_p(_stream()))
    _lib.check(st, "cavp_layernorm")
    return out


def layernorm_residual(x: torch.Tensor, branch: torch.Tensor, row_scale: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                       out_sum: torch.Tensor, out_norm: torch.Tensor, eps: float):
    """out_sum = x + row_scale[group of the row] * branch, out_norm = LayerNorm(out_sum) in one pass; dense [groups * n, C]
    tensors of one dtype, row_scale f32 [groups] (timm drop_path's mask / keep per image)."""
    _need_gpu(x, branch, row_scale, gamma, beta, out_sum, out_norm)
    c = x.shape[-1]
    rows = x.numel() // c
    ts = (x, branch, out_sum, out_norm)
    if not all(t.is_contiguous() and t.shape == x.shape and t.dtype == x.dtype for t in ts) or row_scale.dtype != torch.float32 \
            or not row_scale.is_contiguous() or rows % row_scale.numel() or gamma.dtype != torch.float32 or beta.dtype != torch.float32:
        raise _lib.CavpError("layernorm_residual: dense tensors of one shape / dtype, f32 gamma / beta and an f32 factor per row group")
    st = _lib.load().cavp_layernorm_residual(dtype_code(x.dtype), _ptr(x), _ptr(branch), _ptr(row_scale), rows // row_scale.numel(),
                                             _ptr(gamma), _ptr(beta), _ptr(out_sum), _ptr(out_norm), rows, c, c, c, C.c_float(eps),
                                             C.c_void_p(_stream()))
    _lib.check(st, "cavp_layernorm_residual")
    return out_sum, out_norm


def attn_gate(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, attn: torch.Tensor, heads: int,
              scale: float) -> torch.Tensor:
    """q may hold fewer batch items than out / k / v (a divisor): batch item b then reads q[b % q_batch]."""
    _need_gpu(q, k, v, out, attn)
    qb = q.shape[0]
    b, t, c = out.shape
    if q.shape[1:] != out.shape[1:] or b % qb:
        raise _lib.CavpError("attn_gate: q must be [q_batch, T, C] with q_batch dividing the batch")
    for name, ten in (("q", q), ("k", k), ("v", v), ("out", out), ("attn", attn)):
        if not ten.is_contiguous():
            raise _lib.CavpError(f"attn_gate: {name} must be contiguous")
    if k.numel() != b * c or v.numel() != b * c or attn.numel() != b * heads * t:
        raise _lib.CavpError("attn_gate: shape mismatch")
    if attn.dtype != torch.float32 or len({q.dtype, k.dtype, v.dtype, out.dtype}) != 1:
        raise _lib.CavpError("attn_gate: dtype mismatch")
    st = _lib.load().cavp_attn_gate(dtype_code(q.dtype), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(attn), b, t, heads,
                                    c // heads, C.c_float(scale), qb, C.c_void_p(_stream()))
    _lib.check(st, "cavp_attn_gate")
    return out


def attn1_usable(attn_mod) -> bool:
    """The one-key collapse fits this attention module: supported shape AND f32, contiguous q / proj weights (a model cast with
    .bfloat16() / .half() keeps the q GEMM + gate + proj GEMM route instead of failing in attn1_prepare)."""
    wq, wp = attn_mod.q.weight, attn_mod.proj.weight
    return (attn1_supported(attn_mod.q.in_features, attn_mod.num_heads) and wq.dtype == torch.float32 and wp.dtype == torch.float32
            and wq.is_contiguous() and wp.is_contiguous())


def attn1_supported(c: int, heads: int) -> bool:
    """The one-key attention collapse (cavp_attn1_*) covers this shape (4 heads, C <= 512, C % 8 == 0)."""
    return bool(_lib.load().cavp_attn1_supported(int(c), int(heads)))


def attn1_prepare(wq: torch.Tensor, wp: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float):
    """u[b,h,:] = scale Wq[h-slice,:]^T k[b,h-slice], p[b,h,:] = Wp[:,h-slice] v[b,h-slice]: f32 [B, heads, C] each.
    wq, wp: the f32 [C, C] nn.Linear weights of attn.q / attn.proj; k, v: [B, C] in the compute dtype."""
    _need_gpu(wq, wp, k, v)
    b, c = k.shape
    if wq.dtype != torch.float32 or wp.dtype != torch.float32 or tuple(wq.shape) != (c, c) or tuple(wp.shape) != (c, c):
        raise _lib.CavpError("attn1_prepare: wq / wp must be f32 [C, C]")
    if k.dtype != v.dtype or v.shape != k.shape or not (wq.is_contiguous() and wp.is_contiguous() and k.is_contiguous() and v.is_contiguous()):
        raise _lib.CavpError("attn1_prepare: k, v must be contiguous [B, C] of one dtype")
    u = torch.empty((b, heads, c), dtype=torch.float32, device=k.device)
    pm = torch.empty((b, heads, c), dtype=torch.float32, device=k.device)
    st = _lib.load().cavp_attn1_prepare(dtype_code(k.dtype), _ptr(wq), _ptr(wp), _ptr(k), _ptr(v), _ptr(u), _ptr(pm), b, c, heads,
                                        C.c_float(scale), C.c_void_p(_stream()))
    _lib.check(st, "cavp_attn1_prepare")
    return u, pm


def attn1_fwd(x: torch.Tensor, u: torch.Tensor, pm: torch.Tensor, bp: Optional[torch.Tensor], out: torch.Tensor, attn: torch.Tensor):
    """out[b,t,:] = x[b % xb, t, :] + bp + sum_h sigmoid(x . u[b,h]) pm[b,h];  attn[b,h,t] = the gate.  x: [xb, T, C], out: [B, T, C]."""
    _need_gpu(x, u, pm, out, attn)
    xb, t, c = x.shape
    b, heads = u.shape[0], u.shape[1]
    if tuple(out.shape) != (b, t, c) or b % xb or tuple(attn.shape) != (b, heads, t) or attn.dtype != torch.float32 or out.dtype != x.dtype:
        raise _lib.CavpError("attn1_fwd: shape / dtype mismatch")
    for ten in (x, u, pm, out, attn):
        if not ten.is_contiguous():
            raise _lib.CavpError("attn1_fwd: contiguous tensors required")
    st = _lib.load().cavp_attn1_fwd(dtype_code(x.dtype), _ptr(x), _ptr(u), _ptr(pm), _ptr(bp) if bp is not None else None, _ptr(out),
                                    _ptr(attn), b, xb, t, c, heads, C.c_void_p(_stream()))
    _lib.check(st, "cavp_attn1_fwd")
    return out


def bn_fold(gamma, beta, mean, var, eps: float, scale: torch.Tensor, shift: torch.Tensor) -> None:
    _need_gpu(gamma, beta, mean, var, scale, shift)
    for t in (gamma, beta, mean, var, scale, shift):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.CavpError("bn_fold: contiguous f32 tensors required")
    st = _lib.load().cavp_bn_fold(_ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), C.c_float(eps), _ptr(scale),
                                  _ptr(shift), gamma.numel(), C.c_void_p(_stream()))
    _lib.check(st, "cavp_bn_fold")


def pack_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """torch OIHW (or [out, in]) f32 parameter -> OHWI packed tensor of `dtype` on the same device."""
    _need_gpu(w)
    wd = w.detach()
    if wd.dtype != torch.float32 or not wd.is_contiguous():
        raise _lib.CavpError("pack_weight: contiguous f32 parameter required")
    if wd.dim() == 2:
        cout, cin, kh, kw = wd.shape[0], wd.shape[1], 1, 1
    else:
        cout, cin, kh, kw = wd.shape
    out = torch.empty((cout, kh, kw, cin), dtype=dtype, device=w.device)
    st = _lib.load().cavp_pack_weight_ohwi(dtype_code(dtype), _ptr(wd), _ptr(out), cout, cin, kh, kw,
                                           C.c_void_p(_stream()))
    _lib.check(st, "cavp_pack_weight_ohwi")
    return out


def cast(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    _need_gpu(src, dst)
    if not src.is_contiguous() or not dst.is_contiguous() or src.numel() != dst.numel():
        raise _lib.CavpError("cast: contiguous tensors of equal size required")
    st = _lib.load().cavp_cast(dtype_code(src.dtype), _ptr(src), dtype_code(dst.dtype), _ptr(dst), src.numel(),
                               C.c_void_p(_stream()))
    _lib.check(st, "cavp_cast")
    return dst


# ---- PVTv2 -------------------------------------------------------------------------------------------------------
def sra_attention(q: torch.Tensor, kv: torch.Tensor, out: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """q, out: [B, Nq, heads*64]; kv: [B, Nk, 2*heads*64] (k | v); softmax attention per head (pvt.py:102-130)."""
    _need_gpu(q, kv, out)
    b, nq, c = q.shape
    nk = kv.shape[1]
    if not (q.is_contiguous() and kv.is_contiguous() and out.is_contiguous()) or kv.shape[2] != 2 * c or out.shape != q.shape:
        raise _lib.CavpError("sra_attention: contiguous [B,Nq,C] / [B,Nk,2C] tensors required")
    st = _lib.load().cavp_sra_attention(dtype_code(q.dtype), _ptr(q), _ptr(kv), _ptr(out), b, nq, nk, heads, c // heads,
                                        C.c_float(scale), C.c_void_p(_stream()))
    _lib.check(st, f"cavp_sra_attention B{b} Nq{nq} Nk{nk} heads{heads}")
    return out


def pack_dwconv_weight(w: torch.Tensor) -> torch.Tensor:
    wd = w.detach().contiguous()
    c = wd.shape[0]
    out = torch.empty((9, c), dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().cavp_pack_dwconv_weight(_ptr(wd), _ptr(out), c, C.c_void_p(_stream())), "cavp_pack_dwconv_weight")
    return out


def dwconv3x3(x: torch.Tensor, w9c: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, act: int = ACT_NONE,
              aux: Optional[torch.Tensor] = None):
    """aux (out's shape, act must be ACT_GELU): receives gelu'(pre-activation) for the backward (cavp_dwconv3x3_nhwc_aux)."""
    _need_gpu(x, w9c, bias, out, aux)
    n, h, w, c, ld = _nhwc(x)
    if ld != c or not out.is_contiguous() or out.shape != x.shape:
        raise _lib.CavpError("dwconv3x3: dense NHWC tensors required")
    if aux is not None and (not aux.is_contiguous() or aux.shape != out.shape or aux.dtype != out.dtype):
        raise _lib.CavpError("dwconv3x3: aux must be a dense tensor of the output's shape and dtype")
    st = _lib.load().cavp_dwconv3x3_nhwc_aux(dtype_code(x.dtype), _ptr(x), _ptr(w9c), _ptr(bias), _ptr(out), _ptr(aux), n, h, w, c,
                                             act, C.c_void_p(_stream()))
    _lib.check(st, "cavp_dwconv3x3_nhwc")
    return out


def dwconv3x3_bwd_data(dy: torch.Tensor, w9c: torch.Tensor, dx: torch.Tensor):
    """dx of the depth-wise 3x3 conv from the forward's packed [9][C] weights (the taps are read in reverse order)."""
    _need_gpu(dy, w9c, dx)
    n, h, w, c, ld = _nhwc(dy)
    if ld != c or not dx.is_contiguous() or dx.shape != dy.shape or dx.dtype != dy.dtype:
        raise _lib.CavpError("dwconv3x3_bwd_data: dense NHWC tensors required")
    st = _lib.load().cavp_dwconv3x3_bwd_data_nhwc(dtype_code(dy.dtype), _ptr(dy), _ptr(w9c), _ptr(dx), n, h, w, c,
                                                  C.c_void_p(_stream()))
    _lib.check(st, "cavp_dwconv3x3_bwd_data_nhwc")
    return dx


def conv_smallcin_kxk(x_nchw: torch.Tensor, w_oihw: torch.Tensor, bias, out: torch.Tensor, ks: int, stride: int, pad: int):
    _need_gpu(x_nchw, w_oihw, bias, out)
    n, cin, h, w = x_nchw.shape
    cout = w_oihw.shape[0]
    st = _lib.load().cavp_conv_smallcin_kxk_nchw(dtype_code(out.dtype), _ptr(x_nchw.contiguous()), _ptr(w_oihw.contiguous()),
                                                 _ptr(bias), _ptr(out), n, cin, h, w, cout, ks, stride, pad,
                                                 C.c_void_p(_stream()))
    _lib.check(st, "cavp_conv_smallcin_kxk_nchw")
    return out
