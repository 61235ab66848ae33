"""Parity of the 256x256 implicit-GEMM tile (csrc/conv_igemm_big.hip, tile id 10) against a PyTorch fp32 CPU
reference of the same conv on bf16-rounded operands, plus race screens (bit-identical repeats, a run against the 2-stage
128x128 tile on the same inputs).  MI355X box only; through cavp_amd.ops -> ctypes -> libcavp_hip.so."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_ops import DEV, _act, _check, _ops, _q, _rand, _to_nhwc_dev

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
BIG = 10

CASES = [
    # name, N, H, W, Cin, Cout, k, stride, pad, dil
    ("1x1_one_ktile", 2, 16, 16, 64, 256, 1, 1, 0, 1),           # iters = 1: every phase of the stream is a tile boundary
    ("1x1_304_304_ktail_ctail", 2, 24, 20, 304, 304, 1, 1, 0, 1),  # K tail (304 = 4.75 x 64) and a second, mostly empty, channel tile
    ("3x3_304_256", 1, 40, 36, 304, 256, 3, 1, 1, 1),            # decoder head conv (encoder_decoder.py:62-75) at reduced size
    ("3x3_s2", 2, 30, 30, 128, 128, 3, 2, 1, 1),
    ("3x3_d12_deadtaps", 1, 14, 14, 256, 256, 3, 1, 12, 12),
    ("ragged_13x7_cout40", 3, 13, 7, 48, 40, 3, 1, 1, 1),
    ("1x1_many_tiles", 5, 132, 130, 64, 64, 1, 1, 0, 1),         # 336 pixel tiles on 256 workgroups: the stream crosses output tiles
    ("3x3_many_tiles", 4, 136, 128, 64, 128, 3, 1, 1, 1),        # 272 tiles x 9 K tiles
]


def _run(case, **kw):
    ops = _ops()
    name, n, h, w, cin, cout, k, s, p, d = case
    x = _rand(n, cin, h, w, seed=1)
    wt = _rand(cout, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5)
    xv, _ = _to_nhwc_dev(x, BF)
    wp = ops.pack_weight(wt.to(DEV), BF)
    ref = F.conv2d(_q(x, BF), _q(wt, BF), None, s, p, d)
    out = torch.empty((n, ref.shape[2], ref.shape[3], cout), dtype=BF, device=DEV)
    ops.conv2d(xv, wp, out, kh=k, kw=k, stride=s, pad=p, dil=d, **kw)
    return out, ref, (xv, wp)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_big_tile_matches_reference(case):
    out, ref, _ = _run(case, tile=BIG)
    _check(out.permute(0, 3, 1, 2), ref, BF, case[0] + "/tile10")


@pytest.mark.parametrize("case", [CASES[2], CASES[6], CASES[7]], ids=[CASES[2][0], CASES[6][0], CASES[7][0]])
def test_big_tile_is_race_free(case):
    """Same launch 6 times on the same buffers: bit-identical outputs (a fragment read that overtakes its DMA, or a DMA that
    overtakes a read, shows up as run-to-run differences), and equal to the 128x128 tile up to summation order."""
    ops = _ops()
    name, n, h, w, cin, cout, k, s, p, d = case
    out0, ref, (xv, wp) = _run(case, tile=BIG)
    first = out0.clone()
    for _ in range(5):
        out = torch.empty_like(first)
        ops.conv2d(xv, wp, out, kh=k, kw=k, stride=s, pad=p, dil=d, tile=BIG)
        assert torch.equal(out, first), name + ": repeats differ"
    small = torch.empty_like(first)
    ops.conv2d(xv, wp, small, kh=k, kw=k, stride=s, pad=p, dil=d, tile=1)
    err = float((small.float() - first.float()).abs().max())
    assert err <= 2e-2 * max(1.0, float(ref.abs().max())), (name, err)


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_big_tile_epilogue_fusions(act):
    """scale/shift + per-image bias + residual + activation, reading and writing 16-byte-aligned channel slices."""
    ops = _ops()
    n, h, w, cin, cout = 3, 19, 21, 64, 48
    x, wt = _rand(n, cin, h, w, seed=9), _rand(cout, cin, 1, 1, seed=10, scale=0.12)
    sc, sh = torch.rand(cout, generator=torch.Generator().manual_seed(11)) + 0.5, _rand(cout, seed=12)
    nb, res = _rand(n, cout, seed=13), _rand(n, cout, h, w, seed=14)
    xv, _ = _to_nhwc_dev(x, BF, ld=96, c0=16)
    rv, _ = _to_nhwc_dev(res, BF, ld=64, c0=8)
    big = torch.full((n, h, w, 304), -3.0, dtype=BF, device=DEV)
    out = big[..., 256:304]
    ref = F.conv2d(_q(x, BF), _q(wt, BF)) + nb[:, :, None, None]
    ref = _act(ref * sc[None, :, None, None] + sh[None, :, None, None] + _q(res, BF), act)
    ops.conv2d(xv, ops.pack_weight(wt.to(DEV), BF), out, scale=sc.to(DEV), shift=sh.to(DEV), nbias=nb.to(DEV),
               residual=rv, act=act, tile=BIG)
    _check(out.permute(0, 3, 1, 2), ref, BF, f"tile10 epilogue act{act}")
    assert float((big[..., :256].float() + 3.0).abs().max()) == 0.0, "wrote outside its channel slice"


@pytest.mark.parametrize("case", [CASES[1], CASES[2], CASES[5], CASES[7]], ids=[CASES[1][0], CASES[2][0], CASES[5][0], CASES[7][0]])
def test_big_tile_batchnorm_statistics(case):
    """Per-slab (mean, M2) pairs from the accumulators -> Chan combine == mean / biased variance of the f32 conv output."""
    ops = _ops()
    name, n, h, w, cin, cout, k, s, p, d = case
    x = _rand(n, cin, h, w, seed=21)
    wt = _rand(cout, cin, k, k, seed=22, scale=(cin * k * k) ** -0.5)
    xv, _ = _to_nhwc_dev(x, BF)
    ref = F.conv2d(_q(x, BF), _q(wt, BF), None, s, p, d)
    out = torch.empty((n, ref.shape[2], ref.shape[3], cout), dtype=BF, device=DEV)
    _, stats = ops.conv2d(xv, ops.pack_weight(wt.to(DEV), BF), out, kh=k, kw=k, stride=s, pad=p, dil=d, tile=BIG,
                          want_tile_stats=True)
    assert stats is not None
    ts, tiles, rpt = stats
    rows = out.numel() // cout
    assert rpt == 128 and tiles == (rows + 127) // 128
    ts = ts.cpu().double()
    cnt = torch.tensor([min(rpt, rows - t * rpt) for t in range(tiles)], dtype=torch.float64)
    mean = (ts[:, :, 0] * cnt[:, None]).sum(0) / rows
    m2 = (ts[:, :, 1] + cnt[:, None] * (ts[:, :, 0] - mean[None]) ** 2).sum(0)
    flat = ref.permute(0, 2, 3, 1).reshape(rows, cout).double()
    assert float((mean - flat.mean(0)).abs().max()) <= 1e-4 * max(1.0, float(flat.abs().max()))
    var_ref = flat.var(0, unbiased=False)
    assert float(((m2 / rows) - var_ref).abs().max()) <= 1e-4 * max(1.0, float(var_ref.max()))


def test_big_tile_refuses_what_it_cannot_do():
    ops = _ops()
    from cavp_amd._lib import CavpError
    x, _ = _to_nhwc_dev(_rand(1, 64, 8, 8, seed=1), torch.float32)
    wt = ops.pack_weight(_rand(64, 64, 1, 1, seed=2).to(DEV), torch.float32)
    with pytest.raises(CavpError):   # f32 has no 256x256 tile
        ops.conv2d(x, wt, torch.empty((1, 8, 8, 64), dtype=torch.float32, device=DEV), tile=BIG)
    xb, _ = _to_nhwc_dev(_rand(1, 64, 8, 8, seed=1), BF)
    wb = ops.pack_weight(_rand(20, 64, 1, 1, seed=2).to(DEV), BF)
    with pytest.raises(CavpError):   # Cout not a multiple of the 16-byte vector
        ops.conv2d(xb, wb, torch.empty((1, 8, 8, 20), dtype=BF, device=DEV), tile=BIG)


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("tile", [0, 1, BIG])
def test_token_path_epilogue_fusions(tile, dtype):
    """ABI 6 (cavp_conv2d_nhwc_aux): fc1 + GELU with the derivative as a second output (aux_mode 1), a data-gradient GEMM
    whose result is multiplied by that derivative (aux_mode 2), and a batch-periodic residual (res_rows) - timm Mlp /
    forward_train's duplicated features (attn.py:136-150, cavp_model.py:181) - against plain PyTorch."""
    ops = _ops()
    if tile == BIG and dtype != BF:
        pytest.skip("the 256x256 tile is bf16 only")
    n, t, cin, cout = 4, 512, 96, 256           # 2048 token rows; residual period = 1024 rows
    x = _rand(n, cin, 1, t, seed=31)
    wt = _rand(cout, cin, 1, 1, seed=32, scale=cin ** -0.5)
    bias = _rand(cout, seed=33)
    xv, _ = _to_nhwc_dev(x, dtype)
    wp = ops.pack_weight(wt.to(DEV), dtype)
    pre = F.conv2d(_q(x, dtype), _q(wt, dtype), bias)
    # aux_mode 1
    y = torch.empty((n, 1, t, cout), dtype=dtype, device=DEV)
    d = torch.empty_like(y)
    ops.conv2d(xv, wp, y, shift=bias.to(DEV), act=ops.ACT_GELU, aux=d, aux_mode=1, tile=tile)
    p64 = pre.double()
    dref = (0.5 * (1 + torch.erf(p64 / 2 ** 0.5)) + p64 * torch.exp(-0.5 * p64 * p64) / (2 * torch.pi) ** 0.5).float()
    _check(y.permute(0, 3, 1, 2), F.gelu(pre), dtype, f"gelu out tile{tile}")
    _check(d.permute(0, 3, 1, 2), dref, dtype, f"gelu' out tile{tile}")
    # aux_mode 2 + periodic residual: out = conv * mul + res[p % rows]
    mul = _rand(n, cout, 1, t, seed=34)
    res = _rand(n // 2, cout, 1, t, seed=35)
    mv, _ = _to_nhwc_dev(mul, dtype)
    rv, _ = _to_nhwc_dev(res, dtype)
    out = torch.empty((n, 1, t, cout), dtype=dtype, device=DEV)
    ops.conv2d(xv, wp, out, shift=bias.to(DEV), residual=rv, res_rows=(n // 2) * t, aux=mv, aux_mode=2, tile=tile)
    ref = pre * _q(mul, dtype) + torch.cat((_q(res, dtype), _q(res, dtype)), 0)
    _check(out.permute(0, 3, 1, 2), ref, dtype, f"mul + periodic residual tile{tile}")


def test_tail_split_of_a_nearly_empty_last_round():
    """An automatically planned 256x256-tile launch whose last round of 256 tiles is nearly empty (66 images of 32 x 32 pixels:
    264 tiles) is issued as the 64 images that fill a whole round on the big tile + the 2 remaining images on the small tiles
    (conv_igemm.hip: tail_split_images).  Per-image bias, residual, scale / shift, activation and the accumulated BatchNorm
    sums must all follow the image offset."""
    ops = _ops()
    from cavp_amd import train_ops as T
    n, h, w, cin, cout = 66, 32, 32, 64, 256
    x, wt = _rand(n, cin, h, w, seed=21), _rand(cout, cin, 3, 3, seed=22, scale=(cin * 9) ** -0.5)
    sc, sh = torch.rand(cout, generator=torch.Generator().manual_seed(23)) + 0.5, _rand(cout, seed=24)
    nb, res = _rand(n, cout, seed=25), _rand(n, cout, h, w, seed=26)
    xv, _ = _to_nhwc_dev(x, BF)
    rv, _ = _to_nhwc_dev(res, BF)
    wp = ops.pack_weight(wt.to(DEV), BF)
    conv = F.conv2d(_q(x, BF), _q(wt, BF), None, 1, 1)
    ref = _act((conv + nb[:, :, None, None]) * sc[None, :, None, None] + sh[None, :, None, None] + _q(res, BF), 1)
    out = torch.empty((n, h, w, cout), dtype=BF, device=DEV)
    ops.conv2d(xv, wp, out, kh=3, kw=3, pad=1, scale=sc.to(DEV), shift=sh.to(DEV), nbias=nb.to(DEV), residual=rv, act=1)
    _check(out.permute(0, 3, 1, 2), ref, BF, "tail split / fused epilogue")
    forced = torch.empty_like(out)   # the same conv as ONE launch of the big tile (an explicit tile request never splits)
    ops.conv2d(xv, wp, forced, kh=3, kw=3, pad=1, scale=sc.to(DEV), shift=sh.to(DEV), nbias=nb.to(DEV), residual=rv, act=1, tile=BIG)
    assert torch.equal(out[:64], forced[:64]), "the leading images run on the same tile: bit-identical"
    assert float((out[64:].float() - forced[64:].float()).abs().max()) <= 2e-2 * float(ref.abs().max())
    # per-tile BatchNorm statistics across both launches: the tail continues the 128-row tile sequence of the big tile
    z = torch.empty_like(out)
    _, stats = ops.conv2d(xv, wp, z, kh=3, kw=3, pad=1, want_tile_stats=True)
    assert stats is not None
    ts, tiles, rpt = stats
    assert (tiles, rpt) == (n * h * w // 128, 128)
    g, b = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    f = lambda: torch.empty(cout, device=DEV)
    scale, shift, mean, rstd = f(), f(), f(), f()
    T.bn_finalize_tiles(ts, tiles, rpt, n * h * w, g, b, 1e-5, 0.1, None, None, scale, shift, mean, rstd)
    assert float((mean.cpu() - conv.mean((0, 2, 3))).abs().max()) <= 2e-4, "mean over both launches"
    v_ref = conv.var((0, 2, 3), unbiased=False)
    assert float((rstd.cpu() - 1 / torch.sqrt(v_ref + 1e-5)).abs().max()) <= 2e-4 * float((1 / torch.sqrt(v_ref + 1e-5)).max())
