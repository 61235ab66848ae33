"""Kernel-level parity: every C-ABI entry point vs a plain PyTorch fp32 CPU reference of the same op.
Runs on the MI355X box only (-m gpu); goes through cavp_amd.ops -> ctypes -> libcavp_hip.so."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from cavp_amd import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _to_nhwc_dev(x_nchw, dtype, ld=None, c0=0):
    """NCHW cpu f32 -> NHWC device view (optionally a channel slice [c0, c0+C) of a wider buffer of width ld)."""
    n, c, h, w = x_nchw.shape
    ld = ld or c
    buf = torch.full((n, h, w, ld), 7.0, dtype=dtype, device=DEV)
    buf[..., c0:c0 + c] = x_nchw.permute(0, 2, 3, 1).to(dtype).to(DEV)
    return buf[..., c0:c0 + c], buf


def _q(t, dtype):
    """value after storage rounding (identity for f32)."""
    return t.to(dtype).to(torch.float32)


def _act(y, act):
    from cavp_amd import ops
    if act == ops.ACT_RELU:
        return F.relu(y)
    if act == ops.ACT_LEAKY:
        return F.leaky_relu(y, 0.01)
    if act == ops.ACT_GELU:
        return F.gelu(y)
    return y


def _check(got, ref, dtype, what):
    got = got.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ": non-finite output"
    scale = max(1.0, float(ref.abs().max()))
    tol = (2e-5 if dtype == torch.float32 else 1.2e-2) * scale
    err = float((got - ref).abs().max())
    assert err <= tol, f"{what}: max err {err:.3e} > {tol:.3e} (ref max {scale:.2f})"


CONV_CASES = [
    # name, N, H, W, Cin, Cout, k, stride, pad, dil
    ("1x1_64_64", 2, 56, 56, 64, 64, 1, 1, 0, 1),
    ("3x3_64_128", 2, 28, 28, 64, 128, 3, 1, 1, 1),
    ("3x3_s2", 2, 56, 56, 128, 128, 3, 2, 1, 1),
    ("1x1_s2_ds", 2, 28, 28, 256, 512, 1, 2, 0, 1),
    ("3x3_d2", 2, 14, 14, 128, 128, 3, 1, 2, 2),
    ("3x3_d12_deadtaps", 1, 14, 14, 256, 256, 3, 1, 12, 12),
    ("3x3_d18_centre_only", 1, 14, 14, 256, 64, 3, 1, 18, 18),
    ("3x3_cin304_ktail", 1, 20, 20, 304, 256, 3, 1, 1, 1),
    ("ragged_13x7", 3, 13, 7, 48, 80, 3, 1, 1, 1),
    ("odd_s2_15x9", 1, 15, 9, 64, 64, 3, 2, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 14])
def test_conv_igemm_tiles(case, tile, dtype):
    ops = _ops()
    name, n, h, w, cin, cout, k, s, p, d = case
    if tile in (5, 6, 7, 12, 13, 14) and name not in ("1x1_64_64", "3x3_d12_deadtaps", "ragged_13x7"):
        pytest.skip("small tiles swept on a subset")
    x = _rand(n, cin, h, w, seed=1)
    wt = _rand(cout, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5)
    xv, _ = _to_nhwc_dev(x, dtype)
    wp = ops.pack_weight(wt.to(DEV), dtype)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype), None, s, p, d)
    out = torch.empty((n, ref.shape[2], ref.shape[3], cout), dtype=dtype, device=DEV)
    ops.conv2d(xv, wp, out, kh=k, kw=k, stride=s, pad=p, dil=d, tile=tile)
    _check(out.permute(0, 3, 1, 2), ref, dtype, f"{name}/tile{tile}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("splitk", [1, 2, 3, 7])
def test_conv_splitk(splitk, dtype):
    ops = _ops()
    n, h, w, cin, cout = 1, 14, 14, 256, 96
    x, wt = _rand(n, cin, h, w, seed=3), _rand(cout, cin, 3, 3, seed=4, scale=0.03)
    bias = _rand(cout, seed=5)
    xv, _ = _to_nhwc_dev(x, dtype)
    ref = F.relu(F.conv2d(_q(x, dtype), _q(wt, dtype), bias, 1, 1, 1))
    out = torch.empty((n, h, w, cout), dtype=dtype, device=DEV)
    ops.conv2d(xv, ops.pack_weight(wt.to(DEV), dtype), out, kh=3, kw=3, pad=1, shift=bias.to(DEV), act=ops.ACT_RELU,
               splitk=splitk)
    _check(out.permute(0, 3, 1, 2), ref, dtype, f"splitk{splitk}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("cout", [2, 22, 71, 24])
def test_conv_classifier_small_cout(cout, dtype):
    """classifier 1x1 256->C with bias (encoder_decoder.py:65), C in {2, 22, 24, 71}: scalar-tail epilogue."""
    ops = _ops()
    x, wt, b = _rand(2, 256, 12, 10, seed=6), _rand(cout, 256, 1, 1, seed=7, scale=0.06), _rand(cout, seed=8)
    xv, _ = _to_nhwc_dev(x, dtype)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype), b)
    out = torch.empty((2, 12, 10, cout), dtype=dtype, device=DEV)
    ops.conv2d(xv, ops.pack_weight(wt.to(DEV), dtype), out, shift=b.to(DEV))
    _check(out.permute(0, 3, 1, 2), ref, dtype, f"cls{cout}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_conv_epilogue_fusions(act, dtype):
    """scale/shift (folded BN) + per-image bias + residual + activation + channel-slice in/out (free concat)."""
    ops = _ops()
    n, h, w, cin, cout = 2, 9, 11, 64, 48
    x, wt = _rand(n, cin, h, w, seed=9), _rand(cout, cin, 1, 1, seed=10, scale=0.12)
    sc, sh = torch.rand(cout, generator=torch.Generator().manual_seed(11)) + 0.5, _rand(cout, seed=12)
    nb, res = _rand(n, cout, seed=13), _rand(n, cout, h, w, seed=14)
    xv, _ = _to_nhwc_dev(x, dtype, ld=96, c0=16)          # read a channel slice
    rv, _ = _to_nhwc_dev(res, dtype, ld=64, c0=8)
    big = torch.full((n, h, w, 304), -3.0, dtype=dtype, device=DEV)
    out = big[..., 256:304]                               # write the [256:304) slice like `reduce` does
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype)) + nb[:, :, None, None]
    ref = _act(ref * sc[None, :, None, None] + sh[None, :, None, None] + _q(res, dtype), act)
    ops.conv2d(xv, ops.pack_weight(wt.to(DEV), dtype), out, scale=sc.to(DEV), shift=sh.to(DEV), nbias=nb.to(DEV),
               residual=rv, act=act)
    _check(out.permute(0, 3, 1, 2), ref, dtype, f"epilogue act{act}")
    assert float((big[..., :256].float() + 3.0).abs().max()) == 0.0, "wrote outside its channel slice"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("rows,cin,cout", [(2, 12288, 4096), (64, 4096, 304), (3, 2048, 256), (6272, 304, 1216)])
def test_linear(rows, cin, cout, dtype):
    ops = _ops()
    x, wt, b = _rand(rows, cin, seed=15), _rand(cout, cin, seed=16, scale=cin ** -0.5), _rand(cout, seed=17)
    ref = F.gelu(F.linear(_q(x, dtype), _q(wt, dtype), b))
    out = torch.empty((rows, cout), dtype=dtype, device=DEV)
    ops.linear(x.to(dtype).to(DEV), ops.pack_weight(wt.to(DEV), dtype), out, bias=b.to(DEV), act=ops.ACT_GELU)
    _check(out, ref, dtype, f"linear {rows}x{cin}->{cout}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("cin,stride,hw", [(3, 2, (32, 40)), (1, 1, (96, 64)), (3, 2, (31, 45)), (3, 1, (6, 300)), (2, 2, (9, 515))])
def test_conv3x3_smallcin(cin, stride, hw, dtype):
    ops = _ops()
    x, wt = _rand(2, cin, *hw, seed=18), _rand(64, cin, 3, 3, seed=19, scale=0.3)
    sc, sh = torch.rand(64, generator=torch.Generator().manual_seed(20)) + 0.5, _rand(64, seed=21)
    ref = F.relu(F.conv2d(x, wt, None, stride, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    out = torch.empty((2, ref.shape[2], ref.shape[3], 64), dtype=dtype, device=DEV)
    ops.conv3x3_smallcin_nchw(x.to(DEV), wt.to(DEV), out, stride=stride, scale=sc.to(DEV), shift=sh.to(DEV), act=ops.ACT_RELU)
    _check(out.permute(0, 3, 1, 2), ref, dtype, "smallcin")


@pytest.mark.parametrize("cin,stride,hw", [(3, 2, (224, 224)), (1, 1, (96, 64)), (3, 2, (31, 45)), (2, 1, (5, 131))])
def test_conv3x3_smallcin_bf16_matrix_core_path_keeps_f32_inputs(cin, stride, hw):
    """The bf16 stem conv (Cout = 64) runs on the MFMA with the f32 image and weights split into bf16 hi + lo pairs (round 6): every output
    must be the f32 convolution rounded ONCE to bf16 - |err| <= 2^-8 |ref| + 2e-4 per element, i.e. no bf16 rounding of the inputs
    (that would be ~27 products x 2^-9 each: ~1e-2 on these inputs; the hi + lo split leaves ~2^-18 per product)."""
    ops = _ops()
    x, wt = _rand(2, cin, *hw, seed=118) * 2.0, _rand(64, cin, 3, 3, seed=119, scale=0.5)
    ref = F.conv2d(x.double(), wt.double(), None, stride, 1).float()
    out = torch.empty((2, ref.shape[2], ref.shape[3], 64), dtype=torch.bfloat16, device=DEV)
    ops.conv3x3_smallcin_nchw(x.to(DEV), wt.to(DEV), out, stride=stride, scale=None, shift=None, act=ops.ACT_NONE)
    got = out.permute(0, 3, 1, 2).float().cpu()
    err = (got - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 2e-4
    assert bool((err <= bound).all()), f"max excess {float((err - bound).max()):.3e} at scale {float(ref.abs().max()):.2f}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("k,s,p,hw", [(3, 2, 1, (112, 112)), (2, 2, 0, (96, 64)), (3, 2, 1, (15, 21))])
def test_maxpool(k, s, p, hw, dtype):
    ops = _ops()
    x = _rand(2, 64, *hw, seed=22)
    ref = F.max_pool2d(_q(x, dtype), k, s, p)
    xv, _ = _to_nhwc_dev(x, dtype)
    out = torch.empty((2, ref.shape[2], ref.shape[3], 64), dtype=dtype, device=DEV)
    ops.maxpool(xv.contiguous(), out, k, s, p)
    assert torch.equal(out.permute(0, 3, 1, 2).float().cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_global_avgpool(dtype):
    ops = _ops()
    x = _rand(3, 200, 14, 14, seed=23)
    xv, _ = _to_nhwc_dev(x, dtype)
    out = torch.empty((3, 200), dtype=torch.float32, device=DEV)
    ops.global_avgpool(xv, out)
    _check(out, _q(x, dtype).flatten(2).mean(-1), torch.float32, "gap")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("hi,ho", [((14, 14), (56, 56)), ((7, 9), (24, 40)), ((28, 28), (56, 56)), ((1, 1), (4, 4))])
def test_bilinear_nhwc(hi, ho, align, dtype):
    ops = _ops()
    x = _rand(2, 32, *hi, seed=24)
    ref = F.interpolate(_q(x, dtype), size=ho, mode="bilinear", align_corners=align)
    xv, _ = _to_nhwc_dev(x, dtype)
    big = torch.zeros((2, ho[0], ho[1], 48), dtype=dtype, device=DEV)
    ops.bilinear(xv, big[..., :32], align)
    _check(big[..., :32].permute(0, 3, 1, 2), ref, dtype, "bilinear")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("C", [2, 22, 71])
def test_bilinear_to_nchw(C, dtype):
    ops = _ops()
    x = _rand(2, C, 56, 56, seed=25)
    ref = F.interpolate(_q(x, dtype), size=(224, 224), mode="bilinear", align_corners=False)
    xv, _ = _to_nhwc_dev(x, dtype)
    out = torch.empty((2, C, 224, 224), dtype=torch.float32, device=DEV)
    ops.bilinear_to_nchw(xv, out, False)
    _check(out, ref, torch.float32 if dtype == torch.float32 else dtype, "bilinear_to_nchw")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("rows,C", [(3136, 304), (2, 304), (17, 1216), (5, 64), (1031, 64), (77, 128), (9, 112), (33, 320), (130, 512)])
def test_layernorm(rows, C, dtype):
    ops = _ops()
    x = _rand(rows, C, seed=26) * 3 + 1
    g, b = torch.rand(C, generator=torch.Generator().manual_seed(27)) + 0.5, _rand(C, seed=28)
    ref = F.layer_norm(_q(x, dtype), (C,), g, b, 1e-5)
    out = torch.empty((rows, C), dtype=dtype, device=DEV)
    ops.layernorm(x.to(dtype).to(DEV), g.to(DEV), b.to(DEV), out, 1e-5)
    _check(out, ref, dtype, "layernorm")


@pytest.mark.parametrize("H,hd", [(4, 76), (4, 28), (4, 128), (2, 64), (8, 32)], ids=["4x76", "4x28", "4x128", "2x64", "8x32"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_attn_gate(dtype, H, hd):
    """4 heads (every CAVP configuration) take the one-DPP-row-per-head kernel, other head counts the masked wave reductions."""
    ops = _ops()
    B, T = 3, 200
    q, k, v = _rand(B, T, H * hd, seed=29), _rand(B, H * hd, seed=30), _rand(B, H * hd, seed=31)
    qq, kq, vq = _q(q, dtype), _q(k, dtype), _q(v, dtype)
    s = torch.sigmoid((qq.view(B, T, H, hd) * kq.view(B, 1, H, hd)).sum(-1) * hd ** -0.5)   # [B,T,H]
    ref_o = (s[..., None] * vq.view(B, 1, H, hd)).reshape(B, T, H * hd)
    out = torch.empty((B, T, H * hd), dtype=dtype, device=DEV)
    attn = torch.empty((B, H, T), dtype=torch.float32, device=DEV)
    ops.attn_gate(q.to(dtype).to(DEV), k.to(dtype).to(DEV), v.to(dtype).to(DEV), out, attn, H, hd ** -0.5)
    _check(out, ref_o, dtype, "attn_gate.o")
    _check(attn, s.permute(0, 2, 1).contiguous(), torch.float32, "attn_gate.attn")


def test_bn_fold_pack_cast():
    ops = _ops()
    C = 300
    g, b, m = _rand(C, seed=32), _rand(C, seed=33), _rand(C, seed=34)
    v = torch.rand(C, generator=torch.Generator().manual_seed(35)) + 0.5
    sc, sh = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_fold(g.to(DEV), b.to(DEV), m.to(DEV), v.to(DEV), 1e-5, sc, sh)
    rs = g / torch.sqrt(v + 1e-5)
    _check(sc, rs, torch.float32, "bn scale")
    _check(sh, b - m * rs, torch.float32, "bn shift")
    w = _rand(10, 24, 3, 3, seed=36)
    for dt in (torch.float32, torch.bfloat16):
        p = ops.pack_weight(w.to(DEV), dt)
        assert torch.equal(p.float().cpu(), w.permute(0, 2, 3, 1).contiguous().to(dt).float())
    x = _rand(1000, seed=37)
    y = ops.cast(x.to(DEV), torch.empty(1000, dtype=torch.bfloat16, device=DEV))
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))


def test_errors_are_loud():
    from cavp_amd import _lib
    ops = _ops()
    x = torch.zeros((1, 4, 4, 6), device=DEV)           # Cin=6 is not a multiple of the 16-byte vector
    w = torch.zeros((8, 1, 1, 6), device=DEV)
    with pytest.raises(_lib.CavpError):
        ops.conv2d(x, w, torch.zeros((1, 4, 4, 8), device=DEV))
    with pytest.raises(_lib.CavpError):
        ops.conv2d(torch.zeros((1, 4, 4, 8)), torch.zeros((8, 1, 1, 8)), torch.zeros((1, 4, 4, 8)))  # CPU tensors


# ---- PVTv2 kernels ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("B,Nq,Nk,heads", [(2, 1024, 256, 5), (1, 4096, 64, 1), (2, 100, 256, 8), (1, 256, 200, 2)])
def test_sra_attention(B, Nq, Nk, heads, dtype):
    ops = _ops()
    C = heads * 64
    q, kv = _rand(B, Nq, C, seed=50), _rand(B, Nk, 2 * C, seed=51)
    qq, kk = _q(q, dtype), _q(kv, dtype)
    qh = qq.view(B, Nq, heads, 64).permute(0, 2, 1, 3)
    kh = kk[..., :C].reshape(B, Nk, heads, 64).permute(0, 2, 1, 3)
    vh = kk[..., C:].reshape(B, Nk, heads, 64).permute(0, 2, 1, 3)
    ref = ((qh @ kh.transpose(-2, -1)) * 0.125).softmax(-1) @ vh
    ref = ref.transpose(1, 2).reshape(B, Nq, C)
    out = torch.empty((B, Nq, C), dtype=dtype, device=DEV)
    ops.sra_attention(q.to(dtype).to(DEV), kv.to(dtype).to(DEV), out, heads, 0.125)
    _check(out, ref, dtype, "sra_attention")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_dwconv_and_patch_embed_and_sr_conv(dtype):
    ops = _ops()
    x, w, b = _rand(2, 64, 13, 17, seed=52), _rand(64, 1, 3, 3, seed=53, scale=0.3), _rand(64, seed=54)
    ref = F.gelu(F.conv2d(_q(x, dtype), w, b, 1, 1, 1, 64))
    xv, _ = _to_nhwc_dev(x, dtype)
    out = torch.empty((2, 13, 17, 64), dtype=dtype, device=DEV)
    ops.dwconv3x3(xv.contiguous(), ops.pack_dwconv_weight(w.to(DEV)), b.to(DEV), out, act=ops.ACT_GELU)
    _check(out.permute(0, 3, 1, 2), ref, dtype, "dwconv3x3+gelu")
    # GELU with its derivative as second output (training pass of pvt.py:46-55), on the strip kernel (W >= 8) and, with a narrow
    # map, on the pixel-per-thread kernel
    for (hh, ww) in ((13, 17), (9, 3)):
        xs = x[:, :, :hh, :ww].contiguous()
        pre = F.conv2d(_q(xs, dtype), w, b, 1, 1, 1, 64).requires_grad_(True)
        F.gelu(pre).sum().backward()
        xv2, _ = _to_nhwc_dev(xs, dtype)
        o2 = torch.empty((2, hh, ww, 64), dtype=dtype, device=DEV)
        aux = torch.empty_like(o2)
        ops.dwconv3x3(xv2.contiguous(), ops.pack_dwconv_weight(w.to(DEV)), b.to(DEV), o2, act=ops.ACT_GELU, aux=aux)
        _check(o2.permute(0, 3, 1, 2), F.gelu(pre.detach()), dtype, f"dwconv3x3+gelu (aux call, W={ww})")
        _check(aux.permute(0, 3, 1, 2), pre.grad, dtype, f"gelu' second output (W={ww})")
        # data gradient from the SAME packed weights (reversed taps), and the weights packed through the multi-tensor pack
        # (a [1][C][3][3] OIHW -> OHWI permutation) equal pack_dwconv_weight's
        xg = _q(xs, dtype).requires_grad_(True)
        dy = _rand(2, 64, hh, ww, seed=61)
        F.conv2d(xg, w, None, 1, 1, 1, 64).backward(_q(dy, dtype))
        dyv, _ = _to_nhwc_dev(dy, dtype)
        from cavp_amd import train_ops as T
        w9c = torch.empty((9, 64), dtype=torch.float32, device=DEV)
        T.pack_weights_multi([(w.to(DEV).view(1, 64, 3, 3), w9c, None)], torch.float32)
        assert torch.equal(w9c, ops.pack_dwconv_weight(w.to(DEV)))
        dx = torch.empty((2, hh, ww, 64), dtype=dtype, device=DEV)
        ops.dwconv3x3_bwd_data(dyv.contiguous(), w9c, dx)
        _check(dx.permute(0, 3, 1, 2), xg.grad, dtype, f"dwconv3x3 data gradient (W={ww})")
    # 7x7 stride-4 overlapping patch embedding (Cin = 3)
    img, w7, b7 = _rand(2, 3, 64, 96, seed=55), _rand(64, 3, 7, 7, seed=56, scale=0.1), _rand(64, seed=57)
    ref = F.conv2d(img, w7, b7, 4, 3)
    out = torch.empty((2, ref.shape[2], ref.shape[3], 64), dtype=dtype, device=DEV)
    ops.conv_smallcin_kxk(img.to(DEV), w7.to(DEV), b7.to(DEV), out, 7, 4, 3)
    _check(out.permute(0, 3, 1, 2), ref, dtype, "patch_embed7x7")
    # spatial-reduction conv k = s = sr as a (KH = sr, KW = 1) conv over [N, H, W/sr, sr*C], stride (sr, 1)
    for sr, C in ((8, 64), (4, 128), (2, 320)):
        xs, ws, bs = _rand(2, C, 16, 32, seed=58), _rand(C, C, sr, sr, seed=59, scale=(C * sr * sr) ** -0.5), _rand(C, seed=60)
        ref = F.conv2d(_q(xs, dtype), _q(ws, dtype), bs, sr)
        xv, _ = _to_nhwc_dev(xs, dtype)
        xin = xv.contiguous().view(2, 16, 32 // sr, sr * C)
        out = torch.empty((2, 16 // sr, 32 // sr, C), dtype=dtype, device=DEV)
        ops.conv2d(xin, ops.pack_weight(ws.to(DEV), dtype), out, kh=sr, kw=1, stride=sr, stride_w=1, shift=bs.to(DEV))
        _check(out.permute(0, 3, 1, 2), ref, dtype, f"sr_conv{sr}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("k,s,p,hw,ch", [(3, 2, 1, (28, 30), 64), (2, 2, 0, (12, 8), 48), (3, 2, 1, (15, 9), 128)])
def test_maxpool_affine_equals_scale_shift_act_then_maxpool(k, s, p, hw, ch, dtype):
    """cavp_maxpool_affine_nhwc (round 6) == cavp_scale_shift_act + cavp_maxpool_nhwc, values AND arg-max, bit for bit; and the values match
    torch on the same quantised input."""
    ops = _ops()
    from cavp_amd import train_ops as T
    x = _q(_rand(2, ch, *hw, seed=61), dtype)
    sc, sh = torch.rand(ch, generator=torch.Generator().manual_seed(62)) + 0.5, _rand(ch, seed=63)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)
    act = torch.empty_like(xd)
    T.scale_shift_act(xd, sc.to(DEV), sh.to(DEV), act, ops.ACT_RELU)
    ho, wo = (hw[0] + 2 * p - k) // s + 1, (hw[1] + 2 * p - k) // s + 1
    y0, y1 = (torch.empty((2, ho, wo, ch), dtype=dtype, device=DEV) for _ in range(2))
    a0, a1 = (torch.empty((2, ho, wo, ch), dtype=torch.uint8, device=DEV) for _ in range(2))
    ops.maxpool(act, y0, k, s, p, argmax=a0)
    ops.maxpool(xd, y1, k, s, p, argmax=a1, scale=sc.to(DEV), shift=sh.to(DEV), act=ops.ACT_RELU)
    assert torch.equal(y0, y1) and torch.equal(a0, a1)
    ref = F.max_pool2d(F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]), k, s, p)
    _check(y1.permute(0, 3, 1, 2), ref, dtype, "maxpool_affine")
