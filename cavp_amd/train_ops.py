"""ctypes wrappers for the training-side C-ABI entry points (include/cavp_hip.h, second half).  Same rules as ops.py:
torch supplies device memory and the current stream only."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib, ops
from ._lib import ConvDesc
from .ops import _need_gpu, _nhwc, _ptr, _stream, dtype_code


def _rows(t: torch.Tensor) -> Tuple[int, int, int]:
    """(rows, C, ld) of a tensor viewed as [rows][C] with uniform row stride ld (NHWC views, [B,T,C], [M,C])."""
    c = t.shape[-1]
    if t.dim() == 1:
        return 1, c, c
    if t.stride(-1) != 1 and c > 1:
        raise _lib.CavpError("last dim must be contiguous")
    lead = [(s, st) for s, st in zip(t.shape[:-1], t.stride()[:-1]) if s > 1]
    rows = 1
    for s, _ in lead:
        rows *= s
    if not lead:
        return 1, c, c
    ld = lead[-1][1]
    exp = ld
    for s, st in reversed(lead):
        if st != exp:
            raise _lib.CavpError(f"not a uniform-row view: shape {tuple(t.shape)} stride {t.stride()}")
        exp *= s
    if ld < c:
        raise _lib.CavpError("row stride smaller than the row")
    return rows, c, ld


def _check(st, what):
    _lib.check(st, what)


def _s():
    return C.c_void_p(_stream())


def conv2d_dgrad(dy: torch.Tensor, w_t: torch.Tensor, dx: torch.Tensor, *, kh: int, kw: int, stride: int, pad: int,
                 dil: int, residual: Optional[torch.Tensor] = None, scale=None, shift=None, act: int = 0,
                 mul: Optional[torch.Tensor] = None, bnb: Optional[dict] = None):
    """Data gradient of a forward conv (kh x kw, stride, pad, dil): dx = conv_transpose(dy, w) [* mul] (+ residual).
    mul (dx's shape): element-wise multiplier applied in the epilogue - the gelu' tensor of a fused fc1 + GELU forward.
    dy: [N,Ho,Wo,Cout_f] view, w_t: cavp_pack_weight_dgrad weights [Cin_f][kh][kw][Cout_f], dx: [N,H,W,Cin_f] view.
    bnb (dx is the gradient of a BatchNorm + activation output): dict(z, out | None, scale, shift, mean, rstd, act) - the launch
    stores dx * act'(.) and writes the BatchNorm-backward partial sums per pixel tile (cavp_conv2d_nhwc_bnbwd).  Returns
    (partials f32 [tiles][C][2], tiles) then - or, with bnb["sums"] (pre-zeroed f32 [2][C]: the tiles add into it with atomics),
    (sums, 0) - or None when this launch cannot carry them (dx is then the plain gradient)."""
    _need_gpu(dy, w_t, dx, residual)
    lib = _lib.load()
    n, ho, wo, cof, ldx = _nhwc(dy)
    n2, h, w, cif, ldy = _nhwc(dx)
    if n2 != n or w_t.numel() != cif * kh * kw * cof or w_t.dtype != dy.dtype or dx.dtype != dy.dtype:
        raise _lib.CavpError("conv2d_dgrad: shape / dtype mismatch")
    ldr = 0
    if residual is not None:
        rn, rh, rw, rc, ldr = _nhwc(residual)
        if (rn, rh, rw, rc) != (n, h, w, cif):
            raise _lib.CavpError("conv2d_dgrad: residual must match dx")
    padt = dil * (kh - 1) - pad
    if padt < 0:
        raise _lib.CavpError("conv2d_dgrad: pad > dil*(k-1) is not supported")
    ld_aux = 0
    if mul is not None:
        mn, mh, mw, mc, ld_aux = _nhwc(mul)
        if (mn, mh, mw, mc) != (n, h, w, cif) or mul.dtype != dy.dtype:
            raise _lib.CavpError("conv2d_dgrad: mul must match dx")
    d = ConvDesc(dtype=dtype_code(dy.dtype), N=n, H=ho, W=wo, Cin=cof, ldx=ldx, Cout=cif, ldy=ldy, KH=kh, KW=kw,
                 stride=1, pad=padt, dil=dil, ldr=ldr, act=act, splitk=0, tile=0, up=stride, Ho=h, Wo=w, stride_w=0,
                 aux_mode=2 if mul is not None else 0, ld_aux=ld_aux)
    if stride == 1:
        eh, ew = ho + 2 * padt - dil * (kh - 1), wo + 2 * padt - dil * (kw - 1)
        if (eh, ew) != (h, w):
            raise _lib.CavpError(f"conv2d_dgrad: dx extent {(h, w)} != {(eh, ew)}")
    nbytes = lib.cavp_conv2d_workspace_bytes(C.byref(d))
    ws = ops.workspace(nbytes, dy.device)
    if bnb is not None:
        tiles, rpt = C.c_int32(0), C.c_int32(0)
        zt = bnb["z"]
        ot = bnb.get("out")
        ok = (mul is None and scale is None and shift is None and act == 0 and zt.dtype == dx.dtype and
              lib.cavp_conv2d_bnbwd_layout(C.byref(d), C.byref(tiles), C.byref(rpt)))
        if ok:
            zn, zh, zw, zc, ld_z = _nhwc(zt)
            ld_out = 0
            if ot is not None:
                on, oh, ow, oc, ld_out = _nhwc(ot)
                if (on, oh, ow, oc) != (n, h, w, cif) or ot.dtype != dx.dtype:
                    raise _lib.CavpError("conv2d_dgrad: bnb['out'] must match dx")
            if (zn, zh, zw, zc) != (n, h, w, cif):
                raise _lib.CavpError("conv2d_dgrad: bnb['z'] must match dx")
            _need_gpu(zt, ot, bnb["mean"], bnb["rstd"], bnb.get("scale"), bnb.get("shift"))
            part = torch.empty((tiles.value, cif, 2), dtype=torch.float32, device=dx.device)
            pv = lambda t: None if t is None else t.data_ptr()   # noqa: E731  (c_void_p structure fields take an int or None)
            ba = _lib.BnBwdArgs(z=pv(zt), out=pv(ot), ld_z=ld_z, ld_out=ld_out, fwd_scale=pv(bnb.get("scale")),
                                fwd_shift=pv(bnb.get("shift")), mean=pv(bnb["mean"]), rstd=pv(bnb["rstd"]), act=int(bnb["act"]),
                                pad_=0, partials=pv(part))
            st = lib.cavp_conv2d_nhwc_bnbwd(C.byref(d), _ptr(dy), _ptr(w_t), _ptr(residual), _ptr(dx), C.byref(ba), _ptr(ws),
                                            C.c_size_t(ws.numel() if ws is not None else 0), _s())
            if st not in (_lib.ERR_UNSUPPORTED, _lib.ERR_ALIGN):
                _check(st, "cavp_conv2d_nhwc_bnbwd")
                return part, tiles.value
            # (the launch-time predicate - 16-byte alignment of y / the residual - is stricter than the layout query's: the launch
            # cannot carry the statistics after all; fall through to the plain gradient and return None as the contract says)
    st = lib.cavp_conv2d_nhwc_aux(C.byref(d), _ptr(dy), _ptr(w_t), _ptr(scale), _ptr(shift), None, _ptr(residual), _ptr(dx),
                                  _ptr(mul), _ptr(ws), C.c_size_t(ws.numel() if ws is not None else 0), None, _s())
    _check(st, "cavp_conv2d_nhwc(dgrad)")
    return None if bnb is not None else dx


def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, dw_ohwi: torch.Tensor, *, kh: int, kw: int, stride: int, pad: int,
                 dil: int, splitk: int = 0, dbias: Optional[torch.Tensor] = None, dw_oihw: bool = False,
                 overwrite: bool = False) -> torch.Tensor:
    """dw_ohwi (f32 [Cout][kh][kw][Cin]) += wgrad(x, dy); dbias (optional f32 [Cout]) += column sums of dy.
    dw_oihw=True: the destination is a torch-layout [Cout][Cin][kh][kw] gradient (no separate unpack pass).
    overwrite=True: dw = wgrad(x, dy) (beta = 0: the destination is neither read nor assumed to be zero)."""
    _need_gpu(x, dy, dw_ohwi, dbias)
    n, h, w, cin, ldx = _nhwc(x)
    n2, ho, wo, cout, ldy = _nhwc(dy)
    if n2 != n or x.dtype != dy.dtype or dw_ohwi.dtype != torch.float32 or dw_ohwi.numel() != cout * kh * kw * cin \
            or not dw_ohwi.is_contiguous():
        raise _lib.CavpError("conv2d_wgrad: shape / dtype mismatch")
    eho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    ewo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    if (ho, wo) != (eho, ewo):
        raise _lib.CavpError("conv2d_wgrad: dy extent does not match the forward conv")
    d = ConvDesc(dtype=dtype_code(x.dtype), N=n, H=h, W=w, Cin=cin, ldx=ldx, Cout=cout, ldy=ldy, KH=kh, KW=kw,
                 stride=stride, pad=pad, dil=dil, ldr=0, act=0, splitk=splitk, tile=0, up=0, Ho=0, Wo=0, stride_w=0,
                 dw_oihw=int(dw_oihw), dw_overwrite=int(overwrite))
    lib = _lib.load()
    ws = ops.workspace(lib.cavp_conv2d_wgrad_workspace_bytes(C.byref(d)), x.device)
    if dbias is not None and (dbias.dtype != torch.float32 or dbias.numel() != cout or not dbias.is_contiguous()):
        raise _lib.CavpError("conv2d_wgrad: dbias must be a dense f32 [Cout] tensor")
    _check(lib.cavp_conv2d_wgrad_nhwc(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw_ohwi), _ptr(dbias), _ptr(ws),
                                      C.c_size_t(ws.numel() if ws is not None else 0), _s()), "cavp_conv2d_wgrad_nhwc")
    return dw_ohwi


def _wgrad_desc(x, dy, dw, kh, kw, stride, pad, dil, splitk, dbias, dw_oihw, overwrite) -> ConvDesc:
    _need_gpu(x, dy, dw, dbias)
    n, h, w, cin, ldx = _nhwc(x)
    n2, ho, wo, cout, ldy = _nhwc(dy)
    if n2 != n or x.dtype != dy.dtype or dw.dtype != torch.float32 or dw.numel() != cout * kh * kw * cin or not dw.is_contiguous():
        raise _lib.CavpError("conv2d_wgrad: shape / dtype mismatch")
    eho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    ewo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    if (ho, wo) != (eho, ewo):
        raise _lib.CavpError("conv2d_wgrad: dy extent does not match the forward conv")
    if dbias is not None and (dbias.dtype != torch.float32 or dbias.numel() != cout or not dbias.is_contiguous()):
        raise _lib.CavpError("conv2d_wgrad: dbias must be a dense f32 [Cout] tensor")
    return ConvDesc(dtype=dtype_code(x.dtype), N=n, H=h, W=w, Cin=cin, ldx=ldx, Cout=cout, ldy=ldy, KH=kh, KW=kw,
                    stride=stride, pad=pad, dil=dil, ldr=0, act=0, splitk=splitk, tile=0, up=0, Ho=0, Wo=0, stride_w=0,
                    dw_oihw=int(dw_oihw), dw_overwrite=int(overwrite))


def conv2d_wgrad_group(jobs) -> None:
    """Up to _lib.WGRAD_GROUP_MAX independent weight gradients in ONE launch (cavp_conv2d_wgrad_group).  jobs: list of dicts
    with the arguments of conv2d_wgrad (x, dy, dw, kh, kw, stride, pad, dil and optionally splitk, dbias, dw_oihw, overwrite).
    Two jobs must not share a dw / dbias."""
    if not jobs:
        return
    if len(jobs) > _lib.WGRAD_GROUP_MAX:
        raise _lib.CavpError(f"conv2d_wgrad_group: at most {_lib.WGRAD_GROUP_MAX} jobs per launch")
    arr = (_lib.WgradJob * len(jobs))()
    for i, j in enumerate(jobs):
        db = j.get("dbias")
        d = _wgrad_desc(j["x"], j["dy"], j["dw"], j["kh"], j["kw"], j["stride"], j["pad"], j["dil"], j.get("splitk", 0), db,
                        j.get("dw_oihw", False), j.get("overwrite", False))
        arr[i] = _lib.WgradJob(d, j["x"].data_ptr(), j["dy"].data_ptr(), j["dw"].data_ptr(), db.data_ptr() if db is not None else None)
    lib = _lib.load()
    pa = C.cast(arr, C.c_void_p)
    ws = ops.workspace(lib.cavp_conv2d_wgrad_group_workspace_bytes(pa, len(jobs)), jobs[0]["x"].device)
    _check(lib.cavp_conv2d_wgrad_group(pa, len(jobs), _ptr(ws), C.c_size_t(ws.numel() if ws is not None else 0), _s()),
           f"cavp_conv2d_wgrad_group ({len(jobs)} jobs)")


def linear_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, dbias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x: [..., Cin], dy: [..., Cout] -> dw f32 [Cout][Cin] (pre-zeroed) += dy^T x."""
    rx, cin, ldx = _rows(x)
    ry, cout, ldy = _rows(dy)
    xv = torch.as_strided(x, (1, 1, rx, cin), (rx * ldx, rx * ldx, ldx, 1))
    yv = torch.as_strided(dy, (1, 1, ry, cout), (ry * ldy, ry * ldy, ldy, 1))
    return conv2d_wgrad(xv, yv, dw, kh=1, kw=1, stride=1, pad=0, dil=1, dbias=dbias)


CE_SCRATCH_FLOATS = 2 + 2 * 1024   # CAVP_CE_SCRATCH_FLOATS (include/cavp_hip.h)


class PackJob(C.Structure):
    """struct cavp_pack_job (include/cavp_hip.h)."""
    _fields_ = [("w_oihw", C.c_void_p), ("ohwi", C.c_void_p), ("dgrad", C.c_void_p),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32)]


def pack_weights_multi(jobs, dtype: torch.dtype) -> None:
    """jobs: list of (w f32 OIHW / [out, in] contiguous, ohwi_out or None, dgrad_out or None).  One launch per 48 tensors."""
    if not jobs:
        return
    arr = (PackJob * len(jobs))()
    for i, (w, o, g) in enumerate(jobs):
        wd = w.detach()
        _need_gpu(wd)
        if wd.dtype != torch.float32 or not wd.is_contiguous():
            raise _lib.CavpError("pack_weights_multi: contiguous f32 parameters required")
        if wd.dim() == 2:
            cout, cin, kh, kw = wd.shape[0], wd.shape[1], 1, 1
        else:
            cout, cin, kh, kw = wd.shape
        for t in (o, g):
            if t is not None and (t.dtype != dtype or not t.is_contiguous() or t.numel() != wd.numel()):
                raise _lib.CavpError("pack_weights_multi: destination dtype / size mismatch")
        arr[i] = PackJob(wd.data_ptr(), o.data_ptr() if o is not None else None, g.data_ptr() if g is not None else None,
                         cout, cin, kh, kw)
    _check(_lib.load().cavp_pack_weights_multi(dtype_code(dtype), C.cast(arr, C.c_void_p), len(jobs), _s()),
           "cavp_pack_weights_multi")


def pack_weight_dgrad(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    wd = w.detach()
    _need_gpu(wd)
    if wd.dim() == 2:
        cout, cin, kh, kw = wd.shape[0], wd.shape[1], 1, 1
    else:
        cout, cin, kh, kw = wd.shape
    out = torch.empty((cin, kh, kw, cout), dtype=dtype, device=w.device)
    _check(_lib.load().cavp_pack_weight_dgrad(dtype_code(dtype), _ptr(wd.contiguous()), _ptr(out), cout, cin, kh, kw, _s()),
           "cavp_pack_weight_dgrad")
    return out


def unpack_weight_grad(g_ohwi: torch.Tensor, grad_oihw: torch.Tensor, accumulate: bool) -> torch.Tensor:
    _need_gpu(g_ohwi, grad_oihw)
    if grad_oihw.dim() == 2:
        cout, cin, kh, kw = grad_oihw.shape[0], grad_oihw.shape[1], 1, 1
    else:
        cout, cin, kh, kw = grad_oihw.shape
    if g_ohwi.numel() != grad_oihw.numel() or g_ohwi.dtype != torch.float32 or grad_oihw.dtype != torch.float32 \
            or not grad_oihw.is_contiguous():
        raise _lib.CavpError("unpack_weight_grad: f32 tensors of equal size required")
    _check(_lib.load().cavp_unpack_weight_grad(_ptr(g_ohwi), _ptr(grad_oihw), cout, cin, kh, kw, int(accumulate), _s()),
           "cavp_unpack_weight_grad")
    return grad_oihw


def smallcin_wgrad(x_nchw: torch.Tensor, dy: torch.Tensor, dw_oihw: torch.Tensor, stride: int) -> torch.Tensor:
    _need_gpu(x_nchw, dy, dw_oihw)
    n, cin, h, w = x_nchw.shape
    cout = dy.shape[-1]
    if dw_oihw.dtype != torch.float32 or tuple(dw_oihw.shape) != (cout, cin, 3, 3) or not dy.is_contiguous():
        raise _lib.CavpError("smallcin_wgrad: bad shapes")
    lib = _lib.load()
    ws = ops.workspace(lib.cavp_conv3x3_smallcin_wgrad_workspace_bytes(dtype_code(dy.dtype), n, cin, h, w, cout, stride),
                       x_nchw.device)
    if ws is None:
        raise _lib.CavpError("smallcin_wgrad: unsupported shape")
    _check(lib.cavp_conv3x3_smallcin_wgrad(dtype_code(dy.dtype), _ptr(x_nchw), _ptr(dy), _ptr(dw_oihw), n, cin, h, w, cout,
                                           stride, _ptr(ws), C.c_size_t(ws.numel()), _s()), "cavp_conv3x3_smallcin_wgrad")
    return dw_oihw


def colstats(x: torch.Tensor, sums: torch.Tensor, sumsq: torch.Tensor, shift: Optional[torch.Tensor] = None) -> None:
    rows, c, ld = _rows(x)
    _need_gpu(x, sums, sumsq, shift)
    _check(_lib.load().cavp_colstats(dtype_code(x.dtype), _ptr(x), _ptr(shift), rows, c, ld, _ptr(sums), _ptr(sumsq), _s()),
           "cavp_colstats")


def zeros(shape, dtype, device) -> torch.Tensor:
    """torch.zeros without a framework fill kernel on the path (cavp_zero_bytes)."""
    t = torch.empty(shape, dtype=dtype, device=device)
    nb = t.numel() * t.element_size()
    if t.device.type != "cuda":
        return t.zero_()
    if nb % 4:
        return t.zero_()    # (odd byte counts never occur on the CAVP path)
    if nb:
        _check(_lib.load().cavp_zero_bytes(_ptr(t), C.c_size_t(nb), _s()), "cavp_zero_bytes")
    return t


def zero_(t: torch.Tensor) -> torch.Tensor:
    """t.zero_() for a dense tensor without a framework fill kernel (cavp_zero_bytes)."""
    nb = t.numel() * t.element_size()
    if t.device.type != "cuda" or nb % 4 or not t.is_contiguous():
        return t.zero_()
    if nb:
        _check(_lib.load().cavp_zero_bytes(_ptr(t), C.c_size_t(nb), _s()), "cavp_zero_bytes")
    return t


def i64_add_table(table_dev: torch.Tensor, inc: int = 1) -> None:
    """*(int64*)table_dev[i] += inc: BatchNorm's num_batches_tracked counters in one launch (table of device addresses)."""
    _check(_lib.load().cavp_i64_add_table(_ptr(table_dev), table_dev.numel(), inc, _s()), "cavp_i64_add_table")


def scale_f32(src: torch.Tensor, alpha: float, dst: torch.Tensor) -> torch.Tensor:
    _need_gpu(src, dst)
    _check(_lib.load().cavp_scale_f32(_ptr(src), C.c_float(alpha), _ptr(dst), src.numel(), _s()), "cavp_scale_f32")
    return dst


def bn_finalize(sums, sumsq, count: int, gamma, beta, eps: float, momentum: float, running_mean, running_var, scale,
                shift, mean, rstd, stat_shift=None) -> None:
    _need_gpu(sums, sumsq, gamma, beta, scale, shift, mean, rstd)
    _check(_lib.load().cavp_bn_finalize(_ptr(sums), _ptr(sumsq), _ptr(stat_shift), count, _ptr(gamma), _ptr(beta), C.c_float(eps),
                                        C.c_float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(scale), _ptr(shift),
                                        _ptr(mean), _ptr(rstd), gamma.numel(), _s()), "cavp_bn_finalize")


def bn_tiles_to_moments(tile_stats, tiles: int, rows_per_tile: int, count: int, moments) -> None:
    """this rank's per-channel (mean, M2) [C, 2] from its tile statistics (SyncBatchNorm: gathered, then bn_finalize_tiles)."""
    _need_gpu(tile_stats, moments)
    _check(_lib.load().cavp_bn_tiles_to_moments(_ptr(tile_stats), tiles, rows_per_tile, count, _ptr(moments), moments.shape[0], _s()),
           "cavp_bn_tiles_to_moments")


def bn_finalize_tiles(tile_stats, tiles: int, rows_per_tile: int, count: int, gamma, beta, eps: float, momentum: float,
                      running_mean, running_var, scale, shift, mean, rstd) -> None:
    _need_gpu(tile_stats, gamma, beta, scale, shift, mean, rstd)
    _check(_lib.load().cavp_bn_finalize_tiles(_ptr(tile_stats), tiles, rows_per_tile, count, _ptr(gamma), _ptr(beta),
                                              C.c_float(eps), C.c_float(momentum), _ptr(running_mean), _ptr(running_var),
                                              _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), gamma.numel(), _s()),
           "cavp_bn_finalize_tiles")


def scale_shift_act(x, scale, shift, y, act: int, residual=None) -> torch.Tensor:
    rows, c, ldx = _rows(x)
    r2, c2, ldy = _rows(y)
    ldr = 0
    if residual is not None:
        _, _, ldr = _rows(residual)
    _need_gpu(x, y, scale, shift, residual)
    if (rows, c) != (r2, c2) or y.dtype != x.dtype:
        raise _lib.CavpError("scale_shift_act: shape mismatch")
    _check(_lib.load().cavp_scale_shift_act(dtype_code(x.dtype), _ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y),
                                            rows, c, ldx, ldr, ldy, act, _s()), "cavp_scale_shift_act")
    return y


def bn_bwd_sum_tiles(partials, tiles: int, sum_g, sum_gz) -> None:
    """sum_g / sum_gz (f32 [C]) += the per-tile pairs of a conv2d_dgrad(bnb=...) launch, summed over the tiles."""
    _need_gpu(partials, sum_g, sum_gz)
    _check(_lib.load().cavp_bn_bwd_sum_tiles(_ptr(partials), tiles, sum_g.numel(), _ptr(sum_g), _ptr(sum_gz), _s()), "cavp_bn_bwd_sum_tiles")


def bn_act_bwd_reduce(dy, y, z, mean, rstd, act: int, sum_g, sum_gz, fwd_scale=None, fwd_shift=None) -> None:
    """y=None (BN + activation without residual): the activation mask is re-derived from z with the forward's folded
    scale / shift, y is not read."""
    rows, c, ld_dy = _rows(dy)
    ld_y = _rows(y)[2] if y is not None else 0
    _, _, ld_z = _rows(z)
    _need_gpu(dy, y, z, mean, rstd, sum_g, sum_gz, fwd_scale, fwd_shift)
    _check(_lib.load().cavp_bn_act_bwd_reduce(dtype_code(dy.dtype), _ptr(dy), _ptr(y), _ptr(z), _ptr(mean), _ptr(rstd), rows,
                                              c, ld_dy, ld_y, ld_z, act, _ptr(sum_g), _ptr(sum_gz), _ptr(fwd_scale),
                                              _ptr(fwd_shift), _s()),
           "cavp_bn_act_bwd_reduce")


def bn_act_bwd_apply(dy, y, z, mean, rstd, gamma, sum_g, sum_gz, act: int, dz, g_out=None, fwd_scale=None,
                     fwd_shift=None, acc=None) -> torch.Tensor:
    """acc = (dbeta, dgamma): the two sums are also ADDED to these affine-gradient buffers (sums that arrive in scratch memory)."""
    rows, c, ld_dy = _rows(dy)
    ld_y = _rows(y)[2] if y is not None else 0
    _, _, ld_z = _rows(z)
    _, _, ld_dz = _rows(dz)
    ld_g = _rows(g_out)[2] if g_out is not None else 0
    _need_gpu(dy, y, z, dz, g_out)
    _check(_lib.load().cavp_bn_act_bwd_apply_acc(dtype_code(dy.dtype), _ptr(dy), _ptr(y), _ptr(z), _ptr(mean), _ptr(rstd),
                                                 _ptr(gamma), _ptr(sum_g), _ptr(sum_gz), rows, c, ld_dy, ld_y, ld_z, act, _ptr(dz),
                                                 ld_dz, _ptr(g_out), ld_g, _ptr(fwd_scale), _ptr(fwd_shift),
                                                 _ptr(acc[0]) if acc is not None else None, _ptr(acc[1]) if acc is not None else None, _s()),
           "cavp_bn_act_bwd_apply")
    return dz


def act_bwd(dy, ref, dx, act: int) -> torch.Tensor:
    rows, c, ld_dy = _rows(dy)
    _, _, ld_ref = _rows(ref)
    _, _, ld_dx = _rows(dx)
    _need_gpu(dy, ref, dx)
    _check(_lib.load().cavp_act_bwd(dtype_code(dy.dtype), _ptr(dy), _ptr(ref), _ptr(dx), rows, c, ld_dy, ld_ref, ld_dx, act,
                                    _s()), "cavp_act_bwd")
    return dx


def add(a, b, out) -> torch.Tensor:
    _need_gpu(a, b, out)
    if not (a.is_contiguous() and b.is_contiguous() and out.is_contiguous()) or a.numel() != b.numel():
        raise _lib.CavpError("add: dense tensors of equal size required")
    _check(_lib.load().cavp_add(dtype_code(a.dtype), _ptr(a), _ptr(b), _ptr(out), a.numel(), _s()), "cavp_add")
    return out


def colsum(x, out) -> torch.Tensor:
    rows, c, ld = _rows(x)
    _need_gpu(x, out)
    _check(_lib.load().cavp_colsum(dtype_code(x.dtype), _ptr(x), rows, c, ld, _ptr(out), _s()), "cavp_colsum")
    return out


def col_tile_stats(x) -> Tuple[torch.Tensor, int, int]:
    """(tile_stats f32 [tiles, C, 2], tiles, 128): per-tile (mean, M2) of x viewed as [rows][C], one pass."""
    _need_gpu(x)
    rows, c, ld = _rows(x)
    tiles = (rows + 127) // 128
    ts = torch.empty((tiles, c, 2), dtype=torch.float32, device=x.device)
    _check(_lib.load().cavp_col_tile_stats(dtype_code(x.dtype), _ptr(x), rows, c, ld, _ptr(ts), _s()), "cavp_col_tile_stats")
    return ts, tiles, 128


def colsum_groups(x, out) -> torch.Tensor:
    """out[g] += column sums of x[g] for every leading index g in ONE launch (x: [G, ..., C] dense or channel-sliced, out f32 [G, C])."""
    _need_gpu(x, out)
    g = x.shape[0]
    rows, c, ld = _rows(x)
    if rows % g or out.dtype != torch.float32 or tuple(out.shape) != (g, c) or not out.is_contiguous():
        raise _lib.CavpError("colsum_groups: x [G, ..., C] and a contiguous f32 [G, C] output required")
    _check(_lib.load().cavp_colsum_groups(dtype_code(x.dtype), _ptr(x), g, rows // g, c, ld, _ptr(out), _s()), "cavp_colsum_groups")
    return out


def layernorm_bwd(dy, x, gamma, dx, dgamma, dbeta, eps: float, add: Optional[torch.Tensor] = None,
                  scaled: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """add (dx's shape and dtype, may be dx): a gradient the input already holds; dx = layernorm_bwd(dy) + add in one pass.
    scaled + row_scale (f32 [groups]): a second, dense output = dx with the rows of group g (rows / groups consecutive rows)
    multiplied by row_scale[g]."""
    rows, c, ld_dy = _rows(dy)
    _, _, ld_x = _rows(x)
    _, _, ld_dx = _rows(dx)
    _need_gpu(dy, x, gamma, dx, dgamma, dbeta, add, scaled, row_scale)
    ld_add = 0
    if add is not None:
        if add.shape != dx.shape or add.dtype != dx.dtype:
            raise _lib.CavpError("layernorm_bwd: `add` must have dx's shape and dtype")
        _, _, ld_add = _rows(add)
    rpg = 0
    if scaled is not None:
        if row_scale is None or row_scale.dtype != torch.float32 or not row_scale.is_contiguous() or rows % row_scale.numel() \
                or not scaled.is_contiguous() or scaled.numel() != rows * c or scaled.dtype != dx.dtype:
            raise _lib.CavpError("layernorm_bwd: `scaled` needs a dense tensor of dx's size and dtype and an f32 factor per row group")
        rpg = rows // row_scale.numel()
    _check(_lib.load().cavp_layernorm_bwd_add(dtype_code(dy.dtype), _ptr(dy), _ptr(x), _ptr(gamma), _ptr(add), ld_add, _ptr(dx),
                                              _ptr(scaled), _ptr(row_scale if scaled is not None else None), rpg, _ptr(dgamma),
                                              _ptr(dbeta), rows, c, ld_dy, ld_x, ld_dx, C.c_float(eps), _s()),
           "cavp_layernorm_bwd_add")
    return dx


def attn_gate_bwd(dout, q, k, v, attn, dattn, dq, dk, dv, heads: int, scale: float) -> None:
    """q: [q_batch, T, C] (q_batch divides the batch of dout / dq), as in ops.attn_gate."""
    _need_gpu(dout, q, k, v, attn, dq, dk, dv)
    b, t, c = dout.shape
    qb = q.shape[0]
    if b % qb or dq.shape != dout.shape:
        raise _lib.CavpError("attn_gate_bwd: shape mismatch")
    _check(_lib.load().cavp_attn_gate_bwd(dtype_code(q.dtype), _ptr(dout), _ptr(q), _ptr(k), _ptr(v), _ptr(attn), _ptr(dattn),
                                          _ptr(dq), _ptr(dk), _ptr(dv), b, t, heads, c // heads, C.c_float(scale), qb, _s()),
           "cavp_attn_gate_bwd")


def attn1_bwd(dout, x, u, pm, dx, dbp):
    """Backward of ops.attn1_fwd over the tokens: dx [xb, T, C] written; returns (dU, dP) f32 [B, heads, C]; dbp (f32 [C] or None)
    accumulated."""
    _need_gpu(dout, x, u, pm, dx)
    b, t, c = dout.shape
    xb, heads = x.shape[0], u.shape[1]
    if b % xb or tuple(dx.shape) != tuple(x.shape) or dout.dtype != x.dtype or dx.dtype != x.dtype:
        raise _lib.CavpError("attn1_bwd: shape / dtype mismatch")
    for ten in (dout, x, u, pm, dx):
        if not ten.is_contiguous():
            raise _lib.CavpError("attn1_bwd: contiguous tensors required")
    lib = _lib.load()
    nbytes = lib.cavp_attn1_bwd_workspace_bytes(b, xb, t, c, heads)
    ws = ops.workspace(nbytes, x.device)
    du = torch.empty((b, heads, c), dtype=torch.float32, device=x.device)
    dp = torch.empty((b, heads, c), dtype=torch.float32, device=x.device)
    _check(lib.cavp_attn1_bwd(dtype_code(x.dtype), _ptr(dout), _ptr(x), _ptr(u), _ptr(pm), _ptr(dx), _ptr(du), _ptr(dp),
                              _ptr(dbp) if dbp is not None else None, _ptr(ws), C.c_size_t(ws.numel()), b, xb, t, c, heads, _s()),
           "cavp_attn1_bwd")
    return du, dp


def attn1_finish(wq, wp, k, v, du, dp, dwq, dwp, heads: int, scale: float):
    """dwq / dwp (f32 [C, C]) accumulated; returns (dk, dv) f32 [B, C]."""
    _need_gpu(wq, wp, k, v, du, dp, dwq, dwp)
    b, c = k.shape
    dk = torch.empty((b, c), dtype=torch.float32, device=k.device)
    dv = torch.empty((b, c), dtype=torch.float32, device=k.device)
    for ten in (wq, wp, k, v, du, dp, dwq, dwp):
        if not ten.is_contiguous():
            raise _lib.CavpError("attn1_finish: contiguous tensors required")
    if dwq.dtype != torch.float32 or dwp.dtype != torch.float32 or dwq.numel() != c * c or dwp.numel() != c * c:
        raise _lib.CavpError("attn1_finish: dwq / dwp must be f32 [C, C]")
    _check(_lib.load().cavp_attn1_finish(dtype_code(k.dtype), _ptr(wq), _ptr(wp), _ptr(k), _ptr(v), _ptr(du), _ptr(dp), _ptr(dwq),
                                         _ptr(dwp), _ptr(dk), _ptr(dv), b, c, heads, C.c_float(scale), _s()), "cavp_attn1_finish")
    return dk, dv


def maxpool_bwd(argmax, dy, dx, k: int, stride: int, pad: int) -> torch.Tensor:
    """argmax: uint8 [N][Ho][Wo][C] written by ops.maxpool; dy: same shape (dtype); dx: [N][H][W][C]."""
    n, h, w, c, _ = _nhwc(dx)
    _need_gpu(argmax, dy, dx)
    if argmax.dtype != torch.uint8 or argmax.numel() != dy.numel() or not argmax.is_contiguous() or not dy.is_contiguous():
        raise _lib.CavpError("maxpool_bwd: dense uint8 argmax of dy's shape required")
    _check(_lib.load().cavp_maxpool_bwd_nhwc(dtype_code(dx.dtype), _ptr(argmax), _ptr(dy), _ptr(dx), n, h, w, c, k, stride, pad, _s()),
           "cavp_maxpool_bwd_nhwc")
    return dx


def bilinear_bwd(dy, dx, align_corners: bool) -> torch.Tensor:
    n, ho, wo, c, ld_dy = _nhwc(dy)
    _, hi, wi, _, ld_dx = _nhwc(dx)
    _need_gpu(dy, dx)
    _check(_lib.load().cavp_bilinear_bwd_nhwc(dtype_code(dy.dtype), _ptr(dy), _ptr(dx), n, hi, wi, c, ld_dx, ho, wo, ld_dy,
                                              int(align_corners), _s()), "cavp_bilinear_bwd_nhwc")
    return dx


def bilinear_bwd_from_nchw(dy_nchw, dx, n_valid: int, align_corners: bool) -> torch.Tensor:
    n, hi, wi, c, ld_dx = _nhwc(dx)
    _need_gpu(dy_nchw, dx)
    ho, wo = dy_nchw.shape[-2:]
    if dy_nchw.dtype != torch.float32 or not dy_nchw.is_contiguous():
        raise _lib.CavpError("bilinear_bwd_from_nchw: dy must be contiguous f32 NCHW")
    _check(_lib.load().cavp_bilinear_bwd_nchw_to_nhwc(dtype_code(dx.dtype), _ptr(dy_nchw), _ptr(dx), n, n_valid, hi, wi, c, ld_dx,
                                                      ho, wo, int(align_corners), _s()), "cavp_bilinear_bwd_nchw_to_nhwc")
    return dx


def bcast_add(x, v, alpha: float) -> torch.Tensor:
    n, h, w, c, ld = _nhwc(x)
    _need_gpu(x, v)
    _check(_lib.load().cavp_bcast_add_nhwc(dtype_code(x.dtype), _ptr(x), _ptr(v), C.c_float(alpha), n, h * w, c, ld, _s()),
           "cavp_bcast_add_nhwc")
    return x


def ce_loss(logits_nchw, labels, n_img: int, ignore_index: int = 255, grad_scale: float = 1.0, want_grad: bool = True):
    """Returns (loss f32[1], dlogits or None)."""
    _need_gpu(logits_nchw, labels)
    nt, c, h, w = logits_nchw.shape
    if logits_nchw.dtype != torch.float32 or not logits_nchw.is_contiguous() or labels.dtype != torch.int64 \
            or not labels.is_contiguous():
        raise _lib.CavpError("ce_loss: contiguous f32 NCHW logits and int64 labels required")
    loss = torch.empty(1, dtype=torch.float32, device=logits_nchw.device)
    scratch = torch.empty(CE_SCRATCH_FLOATS, dtype=torch.float32, device=logits_nchw.device)
    dl = torch.empty_like(logits_nchw) if want_grad else None
    _check(_lib.load().cavp_ce_loss_nchw(_ptr(logits_nchw), _ptr(labels), n_img, nt, c, h * w, ignore_index,
                                         C.c_float(grad_scale), _ptr(loss), _ptr(dl), _ptr(scratch), _s()), "cavp_ce_loss_nchw")
    return loss, dl


def upsample_ce_head(lo, labels, n_img: int, n_classes: int, ignore_index: int = 255, grad_scale: float = 1.0,
                     align_corners: bool = False, want_grad: bool = True):
    """Fused head: bilinear upsample of the NHWC low-resolution logits `lo` [n_total, h, w, ld] to the label
    resolution + cross entropy on the first `n_img` images + gradient w.r.t. `lo`.  Returns (loss f32[1], dlo or None)."""
    _need_gpu(lo, labels)
    if lo.dim() != 4 or not lo.is_contiguous() or labels.dtype != torch.int64 or not labels.is_contiguous() \
            or labels.dim() != 3 or labels.shape[0] != n_img:
        raise _lib.CavpError("upsample_ce_head: contiguous NHWC logits and int64 [n_img, H, W] labels required")
    nt, h, w, ld = lo.shape
    H, W = labels.shape[1:]
    loss = torch.empty(1, dtype=torch.float32, device=lo.device)
    scratch = torch.empty(CE_SCRATCH_FLOATS, dtype=torch.float32, device=lo.device)
    lse = torch.empty((n_img, H, W), dtype=torch.float32, device=lo.device)
    dlo = torch.empty_like(lo) if want_grad else None
    _check(_lib.load().cavp_upsample_ce_head(dtype_code(lo.dtype), _ptr(lo), _ptr(labels), n_img, nt, n_classes, h, w, ld, H, W,
                                             int(align_corners), ignore_index, C.c_float(grad_scale), _ptr(loss),
                                             _ptr(dlo), _ptr(lse), _ptr(scratch), _s()), "cavp_upsample_ce_head")
    return loss, dlo


# ---- PVTv2-B5 training pass (csrc/pvt_train.hip) ------------------------------------------------------------------------
def sra_attention_bwd(q, kv, dout, dq, dkv, heads: int, scale: float) -> None:
    """dq (q's dtype) and dkv ([B, Nk, 2C], overwritten; f32 or q's dtype - the sum of the query splits' partials is stored in
    the dtype its consumer wants) of ops.sra_attention (pvt.py:120-126)."""
    _need_gpu(q, kv, dout, dq, dkv)
    b, nq, c = q.shape
    nk = kv.shape[1]
    if not all(t.is_contiguous() for t in (q, kv, dout, dq, dkv)) or dkv.dtype not in (torch.float32, q.dtype) \
            or dkv.shape != kv.shape or dout.shape != q.shape or dq.shape != q.shape or dout.dtype != q.dtype or kv.dtype != q.dtype:
        raise _lib.CavpError("sra_attention_bwd: contiguous [B,Nq,C] / [B,Nk,2C] tensors of one dtype (dkv: that dtype or f32) required")
    lib = _lib.load()
    nbytes = lib.cavp_sra_attention_bwd_workspace_bytes(b, nq, heads)
    ws = ops.workspace(nbytes, q.device)
    _check(lib.cavp_sra_attention_bwd_to(dtype_code(q.dtype), _ptr(q), _ptr(kv), _ptr(dout), _ptr(dq), _ptr(dkv),
                                         dtype_code(dkv.dtype), b, nq, nk, heads, c // heads, C.c_float(scale), _ptr(ws),
                                         C.c_size_t(ws.numel()), _s()),
           f"cavp_sra_attention_bwd_to B{b} Nq{nq} Nk{nk} heads{heads}")


def dwconv3x3_wgrad(x, dy, dw, dbias, w9c=None, dx=None) -> None:
    """dw f32 [C,1,3,3] +=, dbias f32 [C] += of ops.dwconv3x3 (pvt.py:320-326).  With w9c (the forward's packed [9][C] taps) and
    dx (dy's shape and dtype, overwritten) the same walk over dy also writes the data gradient (cavp_dwconv3x3_bwd)."""
    _need_gpu(x, dy, dw, dbias, w9c, dx)
    n, h, w, c = x.shape
    if not (x.is_contiguous() and dy.is_contiguous() and dw.is_contiguous()) or dy.shape != x.shape or dw.numel() != 9 * c:
        raise _lib.CavpError("dwconv3x3_wgrad: dense NHWC x / dy and a [C,1,3,3] f32 gradient required")
    if (w9c is None) != (dx is None) or (dx is not None and (dx.shape != dy.shape or dx.dtype != dy.dtype or not dx.is_contiguous()
                                                             or w9c.shape != (9, c) or w9c.dtype != torch.float32)):
        raise _lib.CavpError("dwconv3x3_wgrad: the data gradient needs the packed [9][C] f32 taps and a dense dx of dy's shape")
    _check(_lib.load().cavp_dwconv3x3_bwd(dtype_code(x.dtype), _ptr(x), _ptr(dy), _ptr(w9c), _ptr(dx), _ptr(dw), _ptr(dbias),
                                          n, h, w, c, _s()), "cavp_dwconv3x3_bwd")


def conv_smallcin_kxk_wgrad(x_nchw, dy, dw_oihw, ks: int, stride: int, pad: int) -> None:
    """dw_oihw (f32 [Cout, Cin, ks, ks]) += weight gradient of ops.conv_smallcin_kxk: im2col rows x dy on the MFMA
    weight-gradient kernel (a 1x1 layer with K = Cin*ks*ks padded to a multiple of 8)."""
    _need_gpu(x_nchw, dy, dw_oihw)
    n, cin, h, w = x_nchw.shape
    cout = dy.shape[-1]
    if not (x_nchw.is_contiguous() and dy.is_contiguous() and dw_oihw.is_contiguous()) or x_nchw.dtype != torch.float32 \
            or dw_oihw.dtype != torch.float32 or dw_oihw.numel() != cout * cin * ks * ks:
        raise _lib.CavpError("conv_smallcin_kxk_wgrad: contiguous f32 NCHW input, NHWC dy, OIHW f32 gradient required")
    k = cin * ks * ks
    kpad = (k + 7) // 8 * 8
    rows = dy.numel() // cout
    cols = torch.empty((1, 1, rows, kpad), dtype=dy.dtype, device=dy.device)
    _check(_lib.load().cavp_smallcin_kxk_im2col(dtype_code(dy.dtype), _ptr(x_nchw), _ptr(cols), n, cin, h, w, ks, stride, pad, kpad,
                                                _s()), "cavp_smallcin_kxk_im2col")
    tmp = torch.empty((cout, 1, 1, kpad), dtype=torch.float32, device=dy.device)
    conv2d_wgrad(cols, dy.reshape(1, 1, rows, cout), tmp, kh=1, kw=1, stride=1, pad=0, dil=1, overwrite=True)
    dw_oihw.view(cout, k).add_(tmp.view(cout, kpad)[:, :k])     # 9408 values: gradient accumulation, not compute


def space_to_depth(src, dst, b: int, h: int, w: int, c: int, s: int, inverse: bool = False):
    """[B,H,W,C] -> [B,H/s,W/s,s*s*C] (inverse: back).  Both tensors dense."""
    _need_gpu(src, dst)
    if not (src.is_contiguous() and dst.is_contiguous()) or src.numel() != dst.numel() or src.numel() != b * h * w * c:
        raise _lib.CavpError("space_to_depth: dense tensors of B*H*W*C elements required")
    _check(_lib.load().cavp_space_to_depth(dtype_code(src.dtype), _ptr(src), _ptr(dst), b, h, w, c, s, 1 if inverse else 0, _s()),
           "cavp_space_to_depth")
    return dst


def row_scale_add(x, branch, sample_scale, out):
    """out = x + sample_scale[b] * branch (x None: sample_scale[b] * branch); sample_scale f32 [B]."""
    _need_gpu(x, branch, sample_scale, out)
    b = branch.shape[0]
    if not (branch.is_contiguous() and out.is_contiguous() and (x is None or x.is_contiguous())) or sample_scale.numel() != b:
        raise _lib.CavpError("row_scale_add: dense tensors and one scale per batch item required")
    _check(_lib.load().cavp_row_scale_add(dtype_code(branch.dtype), _ptr(x), _ptr(branch), _ptr(sample_scale), _ptr(out), b,
                                          branch.numel() // b, _s()), "cavp_row_scale_add")
    return out
