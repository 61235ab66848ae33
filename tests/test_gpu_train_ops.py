"""Training-side kernels vs torch.autograd on CPU (fp32 reference).  -m gpu."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["f32", "bf16"]


def _mods():
    from cavp_amd import ops, train_ops
    return ops, train_ops


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _q(t, dt):
    return t.to(dt).to(torch.float32)


def _nhwc(x_nchw, dt):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)


def _check(got, ref, dt, what, f32_tol=3e-5, bf16_tol=1.5e-2):
    got = got.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    scale = max(1e-6, float(ref.abs().max()))
    tol = (f32_tol if dt == torch.float32 else bf16_tol) * scale
    err = float((got - ref).abs().max())
    assert err <= tol, f"{what}: max err {err:.3e} > {tol:.3e} (ref max {scale:.3g})"


CONV = [
    # name, N, H, W, Cin, Cout, k, s, p, d
    ("1x1", 2, 14, 14, 64, 128, 1, 1, 0, 1),
    ("3x3", 2, 14, 12, 64, 64, 3, 1, 1, 1),
    ("3x3_s2", 2, 16, 16, 64, 128, 3, 2, 1, 1),
    ("3x3_s2_odd", 1, 15, 13, 64, 64, 3, 2, 1, 1),
    ("1x1_s2", 2, 16, 16, 128, 256, 1, 2, 0, 1),
    # (stride-2 data gradients run on parity-ordered pixel tiles since round 6: odd extents = classes of different sizes, 28 x 28 x 3 =
    # several tiles per class, 1x1 = three of the four classes without any tap)
    ("1x1_s2_odd", 2, 15, 9, 64, 64, 1, 2, 0, 1),
    ("3x3_s2_28", 3, 28, 28, 128, 128, 3, 2, 1, 1),
    ("3x3_s2_d2", 1, 17, 12, 64, 64, 3, 2, 2, 2),
    ("3x3_d2", 1, 14, 14, 128, 128, 3, 1, 2, 2),
    ("3x3_d12", 1, 14, 14, 128, 64, 3, 1, 12, 12),
    ("3x3_d18_dead_taps", 1, 14, 14, 128, 64, 3, 1, 18, 18),   # ASPP rate 18 at 14x14: only the centre tap is live
    ("3x3_304", 1, 12, 12, 304, 256, 3, 1, 1, 1),
    ("1x1_304_48", 2, 9, 7, 256, 48, 1, 1, 0, 1),
]


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", CONV, ids=[c[0] for c in CONV])
def test_conv_dgrad_wgrad(case, dt):
    ops, T = _mods()
    name, n, h, w, cin, cout, k, s, p, d = case
    x = _q(_rand(n, cin, h, w, seed=1), dt).requires_grad_(True)
    wt = _q(_rand(cout, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5), dt).requires_grad_(True)
    y = F.conv2d(x, wt, None, s, p, d)
    dy = _q(_rand(*y.shape, seed=3), dt)
    y.backward(dy)
    dyv = _nhwc(dy, dt)
    # dgrad (+ accumulate into an existing gradient through the residual input)
    prev = _q(_rand(n, cin, h, w, seed=4), dt)
    dx = torch.empty((n, h, w, cin), dtype=dt, device=DEV)
    wT = T.pack_weight_dgrad(wt.detach().to(DEV), dt)
    T.conv2d_dgrad(dyv, wT, dx, kh=k, kw=k, stride=s, pad=p, dil=d, residual=_nhwc(prev, dt))
    _check(dx.permute(0, 3, 1, 2), x.grad + prev, dt, name + ".dgrad")
    # wgrad
    dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device=DEV)
    db = torch.full((cout,), 0.5, dtype=torch.float32, device=DEV)     # the fused bias gradient ACCUMULATES
    T.conv2d_wgrad(_nhwc(x.detach(), dt), dyv, dw, kh=k, kw=k, stride=s, pad=p, dil=d, dbias=db)
    _check(db, dy.sum(dim=(0, 2, 3)) + 0.5, dt, name + ".dbias", f32_tol=5e-5, bf16_tol=2e-2)
    g = torch.zeros((cout, cin, k, k), dtype=torch.float32, device=DEV)
    T.unpack_weight_grad(dw, g, accumulate=False)
    _check(g, wt.grad, dt, name + ".wgrad", f32_tol=5e-5, bf16_tol=2e-2)
    T.unpack_weight_grad(dw, g, accumulate=True)
    _check(g, 2 * wt.grad, dt, name + ".wgrad.acc", f32_tol=5e-5, bf16_tol=2e-2)
    # torch-layout destination (no unpack pass): accumulates into the existing gradient, for every split count
    for sk in (0, 1, 3):
        g2 = torch.full((cout, cin, k, k), 0.25, dtype=torch.float32, device=DEV)
        T.conv2d_wgrad(_nhwc(x.detach(), dt), dyv, g2, kh=k, kw=k, stride=s, pad=p, dil=d, dw_oihw=True, splitk=sk)
        _check(g2, wt.grad + 0.25, dt, name + f".wgrad.oihw.sk{sk}", f32_tol=5e-5, bf16_tol=2e-2)
        # beta = 0: the destination holds garbage (last step's gradient) and is overwritten - bit-identical to
        # accumulating onto zeros, both layouts, dead taps of a dilated kernel included (they become zeros)
        for oihw in (True, False):
            shape = (cout, cin, k, k) if oihw else (cout, k, k, cin)
            ga = torch.zeros(shape, dtype=torch.float32, device=DEV)
            gb = torch.full(shape, 7.5, dtype=torch.float32, device=DEV)
            bb = torch.full((cout,), 0.5, dtype=torch.float32, device=DEV)
            T.conv2d_wgrad(_nhwc(x.detach(), dt), dyv, ga, kh=k, kw=k, stride=s, pad=p, dil=d, dw_oihw=oihw, splitk=sk)
            T.conv2d_wgrad(_nhwc(x.detach(), dt), dyv, gb, kh=k, kw=k, stride=s, pad=p, dil=d, dw_oihw=oihw, splitk=sk,
                           overwrite=True, dbias=bb)
            assert torch.equal(ga, gb), name + f".wgrad.overwrite.sk{sk}.oihw{oihw}"
            _check(bb, dy.sum(dim=(0, 2, 3)) + 0.5, dt, name + ".dbias(overwrite keeps accumulating)", f32_tol=5e-5, bf16_tol=2e-2)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_conv_wgrad_group(dt):
    """cavp_conv2d_wgrad_group: all CONV cases (strides, dilations with dead taps, 304 / 48 channels) as ONE launch, mixed
    destinations (OHWI / torch layout, accumulate / overwrite, with / without bias) - each job against torch.autograd, and
    bit-identical to the per-layer launch when both are given the same split count."""
    ops, T = _mods()
    from cavp_amd._lib import CavpError
    jobs, refs = [], []
    for i, (name, n, h, w, cin, cout, k, s, p, d) in enumerate(CONV):
        x = _q(_rand(n, cin, h, w, seed=10 + i), dt).requires_grad_(True)
        wt = _q(_rand(cout, cin, k, k, seed=30 + i, scale=(cin * k * k) ** -0.5), dt).requires_grad_(True)
        y = F.conv2d(x, wt, None, s, p, d)
        dy = _q(_rand(*y.shape, seed=50 + i), dt)
        y.backward(dy)
        oihw, over, bias = bool(i & 1), bool(i & 2), bool(i % 3 == 0)
        shape = (cout, cin, k, k) if oihw else (cout, k, k, cin)
        dw = torch.full(shape, 7.5 if over else 0.25, dtype=torch.float32, device=DEV)
        db = torch.full((cout,), 0.5, dtype=torch.float32, device=DEV) if bias else None
        jobs.append(dict(x=_nhwc(x.detach(), dt), dy=_nhwc(dy, dt), dw=dw, kh=k, kw=k, stride=s, pad=p, dil=d, dbias=db,
                         dw_oihw=oihw, overwrite=over, splitk=(0, 1, 3)[i % 3]))
        refs.append((name, wt.grad, dy.sum(dim=(0, 2, 3)), oihw, over))
    T.conv2d_wgrad_group(jobs)
    for j, (name, gw, gb, oihw, over) in zip(jobs, refs):
        got = j["dw"] if oihw else j["dw"].permute(0, 3, 1, 2)
        _check(got, gw + (0.0 if over else 0.25), dt, name + ".group.wgrad", f32_tol=5e-5, bf16_tol=2e-2)
        if j["dbias"] is not None:
            _check(j["dbias"], gb + 0.5, dt, name + ".group.dbias", f32_tol=5e-5, bf16_tol=2e-2)
    # same split count -> the same slab sums in the same order as the per-layer launch: bit-identical
    for j in jobs:
        if j["splitk"] == 0:
            continue
        single = torch.full_like(j["dw"], 7.5 if j["overwrite"] else 0.25)
        T.conv2d_wgrad(j["x"], j["dy"], single, kh=j["kh"], kw=j["kw"], stride=j["stride"], pad=j["pad"], dil=j["dil"],
                       dw_oihw=j["dw_oihw"], overwrite=j["overwrite"], splitk=j["splitk"])
        assert torch.equal(single, j["dw"]), "group vs single launch"
    # a group of one, and the limits
    one = dict(jobs[1], dw=torch.zeros_like(jobs[1]["dw"]), overwrite=False, dbias=None)
    T.conv2d_wgrad_group([one])
    name, gw, _, oihw, _ = refs[1]
    _check(one["dw"] if oihw else one["dw"].permute(0, 3, 1, 2), gw, dt, "group of one", f32_tol=5e-5, bf16_tol=2e-2)
    with pytest.raises(CavpError):
        T.conv2d_wgrad_group([jobs[0], dict(jobs[2], dw=jobs[0]["dw"])] if jobs[0]["dw"].numel() == jobs[2]["dw"].numel()
                             else [jobs[0], jobs[0]])          # two jobs adding into one gradient
    with pytest.raises(CavpError):
        T.conv2d_wgrad_group([jobs[0]] * 17)                    # more than CAVP_WGRAD_GROUP_MAX


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("rows,cin,cout", [(6272, 304, 1216), (64, 4096, 304), (4, 12288, 512), (6272, 256, 304)])
def test_linear_wgrad(rows, cin, cout, dt):
    ops, T = _mods()
    x, dy = _q(_rand(rows, cin, seed=5), dt), _q(_rand(rows, cout, seed=6), dt)
    dw = torch.zeros((cout, cin), dtype=torch.float32, device=DEV)
    db = torch.zeros(cout, dtype=torch.float32, device=DEV)
    T.linear_wgrad(x.to(dt).to(DEV), dy.to(dt).to(DEV), dw, dbias=db)
    _check(dw, dy.t() @ x, dt, "linear_wgrad", f32_tol=5e-5, bf16_tol=2e-2)
    _check(db, dy.sum(0), dt, "linear_wgrad.dbias", f32_tol=5e-5, bf16_tol=2e-2)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("act", [1, 2, 0])
def test_bn_train_fwd_bwd(act, dt):
    ops, T = _mods()
    n, c, h, w = 4, 96, 9, 11
    z = _q(_rand(n, c, h, w, seed=7) * 2 + 0.5, dt).requires_grad_(True)
    res = _q(_rand(n, c, h, w, seed=8), dt).requires_grad_(True)
    gamma = (torch.rand(c, generator=torch.Generator().manual_seed(9)) + 0.5).requires_grad_(True)
    beta = _rand(c, seed=10).requires_grad_(True)
    rm, rv = _rand(c, seed=11) * 0.1, torch.rand(c, generator=torch.Generator().manual_seed(12)) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    pre = F.batch_norm(z, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5) + res
    y = F.relu(pre) if act == 1 else (F.leaky_relu(pre, 0.01) if act == 2 else pre)
    dy = _q(_rand(n, c, h, w, seed=13), dt)
    y.backward(dy)
    zv, rvw = _nhwc(z.detach(), dt), _nhwc(res.detach(), dt)
    sums, sq = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    T.colstats(zv, sums, sq)
    f = lambda: torch.empty(c, device=DEV)
    scale, shift, mean, rstd = f(), f(), f(), f()
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    T.bn_finalize(sums, sq, n * h * w, gamma.detach().to(DEV), beta.detach().to(DEV), 1e-5, 0.1, rmd, rvd, scale, shift, mean, rstd)
    yv = torch.empty_like(zv)
    T.scale_shift_act(zv, scale, shift, yv, act, residual=rvw)
    tolf, tolb = 5e-5, 2e-2
    _check(yv.permute(0, 3, 1, 2), y.detach(), dt, "bn fwd", tolf, tolb)
    _check(rmd, rm_ref, torch.float32, "running_mean", 1e-4)
    _check(rvd, rv_ref, torch.float32, "running_var", 1e-4)
    # backward
    dyv = _nhwc(dy, dt)
    sg, sgz = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    T.bn_act_bwd_reduce(dyv, yv, zv, mean, rstd, act, sg, sgz)
    _check(sg, beta.grad, dt, "dbeta", 1e-4, 2e-2)
    _check(sgz, gamma.grad, dt, "dgamma", 1e-4, 2e-2)
    dz, g = torch.empty_like(zv), torch.empty_like(zv)
    T.bn_act_bwd_apply(dyv, yv, zv, mean, rstd, gamma.detach().to(DEV), sg, sgz, act, dz, g_out=g)
    _check(dz.permute(0, 3, 1, 2), z.grad, dt, "dz", 1e-4, 2.5e-2)
    _check(g.permute(0, 3, 1, 2), res.grad, dt, "dres", 1e-5, 1e-2)
    # no residual: the activation mask can be re-derived from z and the folded (scale, shift) - y is not read at all;
    # must give exactly what the y-reading path gives on the same (no-residual) forward
    y2 = torch.empty_like(zv)
    T.scale_shift_act(zv, scale, shift, y2, act)
    sg_y, sgz_y, sg_z, sgz_z = (torch.zeros(c, device=DEV) for _ in range(4))
    T.bn_act_bwd_reduce(dyv, y2, zv, mean, rstd, act, sg_y, sgz_y)
    T.bn_act_bwd_reduce(dyv, None, zv, mean, rstd, act, sg_z, sgz_z, fwd_scale=scale, fwd_shift=shift)
    assert torch.allclose(sg_y, sg_z, rtol=1e-5, atol=1e-5) and torch.allclose(sgz_y, sgz_z, rtol=1e-5, atol=1e-4)
    dz_y, dz_z = torch.empty_like(zv), torch.empty_like(zv)
    T.bn_act_bwd_apply(dyv, y2, zv, mean, rstd, gamma.detach().to(DEV), sg_y, sgz_y, act, dz_y)
    T.bn_act_bwd_apply(dyv, None, zv, mean, rstd, gamma.detach().to(DEV), sg_y, sgz_y, act, dz_z, fwd_scale=scale, fwd_shift=shift)
    assert torch.equal(dz_y, dz_z)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_act_bwd_colsum_add(dt):
    ops, T = _mods()
    x = _q(_rand(333, 304, seed=14), dt).requires_grad_(True)
    dy = _q(_rand(333, 304, seed=15), dt)
    F.gelu(x).backward(dy)
    dx = torch.empty((333, 304), dtype=dt, device=DEV)
    T.act_bwd(dy.to(dt).to(DEV), x.detach().to(dt).to(DEV), dx, ops.ACT_GELU)
    _check(dx, x.grad, dt, "gelu bwd")
    y = F.relu(x.detach())
    T.act_bwd(dy.to(dt).to(DEV), y.to(dt).to(DEV), dx, ops.ACT_RELU)
    _check(dx, dy * (y > 0), dt, "relu bwd")
    cs = torch.zeros(304, device=DEV)
    T.colsum(dy.to(dt).to(DEV), cs)
    _check(cs, dy.sum(0), dt, "colsum", 1e-4, 1e-2)
    o = torch.empty((333, 304), dtype=dt, device=DEV)
    T.add(dy.to(dt).to(DEV), x.detach().to(dt).to(DEV), o)
    _check(o, dy + x.detach(), dt, "add")


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("rows,C", [(3136 * 2, 304), (3, 304), (50, 1216), (1031, 64), (77, 128), (130, 512)])
def test_layernorm_bwd(rows, C, dt):
    ops, T = _mods()
    x = _q(_rand(rows, C, seed=16) * 2 + 0.3, dt).requires_grad_(True)
    g = (torch.rand(C, generator=torch.Generator().manual_seed(17)) + 0.5).requires_grad_(True)
    b = _rand(C, seed=18).requires_grad_(True)
    dy = _q(_rand(rows, C, seed=19), dt)
    F.layer_norm(x, (C,), g, b, 1e-5).backward(dy)
    dx = torch.empty((rows, C), dtype=dt, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    T.layernorm_bwd(dy.to(dt).to(DEV), x.detach().to(dt).to(DEV), g.detach().to(DEV), dx, dg, db, 1e-5)
    _check(dx, x.grad, dt, "ln dx", 5e-5, 2e-2)
    _check(dg, g.grad, dt, "ln dgamma", 1e-4, 1e-2)
    _check(db, b.grad, dt, "ln dbeta", 1e-4, 1e-2)
    # with the gradient the input already holds added in (out of place, and in place: add is dx)
    prior = _q(_rand(rows, C, seed=20), dt)
    pd = prior.to(dt).to(DEV)
    for inplace in (False, True):
        out = pd.clone() if inplace else torch.empty((rows, C), dtype=dt, device=DEV)
        dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        T.layernorm_bwd(dy.to(dt).to(DEV), x.detach().to(dt).to(DEV), g.detach().to(DEV), out, dg2, db2, 1e-5,
                        add=out if inplace else pd)
        _check(out, x.grad + prior, dt, f"ln dx + prior (inplace={inplace})", 5e-5, 2e-2)
        _check(dg2, g.grad, dt, "ln dgamma (add)", 1e-4, 1e-2)
    # second output: dx with the rows of group k (rows / groups consecutive rows) times a factor (DropPath branch gradient)
    groups = 3 if rows % 3 == 0 else 1
    fac = torch.tensor([0.0, 1.25, 1.0][:groups], device=DEV)
    out, sc = torch.empty((rows, C), dtype=dt, device=DEV), torch.empty((rows, C), dtype=dt, device=DEV)
    T.layernorm_bwd(dy.to(dt).to(DEV), x.detach().to(dt).to(DEV), g.detach().to(DEV), out, torch.zeros(C, device=DEV),
                    torch.zeros(C, device=DEV), 1e-5, add=pd, scaled=sc, row_scale=fac)
    want = (out.float().view(groups, -1, C) * fac.view(groups, 1, 1)).view(rows, C)
    assert float((sc.float() - want).abs().max()) <= (1e-6 if dt == torch.float32 else 2e-2) * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("H,hd", [(4, 76), (4, 28), (4, 128), (2, 64), (8, 32)], ids=["4x76", "4x28", "4x128", "2x64", "8x32"])
@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_attn_gate_bwd(dt, H, hd):
    ops, T = _mods()
    B, Tn = 3, 150
    q = _q(_rand(B, Tn, H * hd, seed=20), dt).requires_grad_(True)
    k = _q(_rand(B, H * hd, seed=21), dt).requires_grad_(True)
    v = _q(_rand(B, H * hd, seed=22), dt).requires_grad_(True)
    s = torch.sigmoid((q.view(B, Tn, H, hd) * k.view(B, 1, H, hd)).sum(-1) * hd ** -0.5)
    o = (s[..., None] * v.view(B, 1, H, hd)).reshape(B, Tn, H * hd)
    do = _q(_rand(B, Tn, H * hd, seed=23), dt)
    dattn = _rand(B, H, Tn, seed=24) * 0.1
    (o * do).sum().backward(retain_graph=True)
    (s.permute(0, 2, 1) * dattn).sum().backward()
    attn = s.detach().permute(0, 2, 1).contiguous().to(DEV)
    dq = torch.empty((B, Tn, H * hd), dtype=dt, device=DEV)
    dk, dv = torch.zeros((B, H * hd), device=DEV), torch.zeros((B, H * hd), device=DEV)
    T.attn_gate_bwd(do.to(dt).to(DEV), q.detach().to(dt).to(DEV), k.detach().to(dt).to(DEV), v.detach().to(dt).to(DEV), attn,
                    dattn.to(DEV), dq, dk, dv, H, hd ** -0.5)
    _check(dq, q.grad, dt, "dq", 5e-5, 2e-2)
    _check(dk, k.grad, dt, "dk", 1e-4, 2e-2)
    _check(dv, v.grad, dt, "dv", 1e-4, 2e-2)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("ch", [64, 48], ids=["c64", "c48"])
@pytest.mark.parametrize("k,s,p,hw", [(3, 2, 1, (28, 30)), (2, 2, 0, (12, 8)), (3, 2, 1, (15, 9)), (3, 1, 1, (9, 7))])
def test_maxpool_bwd(k, s, p, hw, dt, ch):
    """(stride 2 takes the row-mapped kernel of round 6 - channel-vector counts that are / are not a power of two; stride 1 the flat one)"""
    ops, T = _mods()
    # post-ReLU style input: exact ties (zeros) inside windows exercise the first-maximum rule
    x = _q(torch.relu(_rand(2, ch, *hw, seed=25)), dt).requires_grad_(True)
    y = F.max_pool2d(x, k, s, p)
    dy = _q(_rand(*y.shape, seed=26), dt)
    y.backward(dy)
    dx = torch.empty((2, hw[0], hw[1], ch), dtype=dt, device=DEV)
    yo = torch.empty((2, y.shape[2], y.shape[3], ch), dtype=dt, device=DEV)
    am = torch.empty(yo.shape, dtype=torch.uint8, device=DEV)
    ops.maxpool(_nhwc(x.detach(), dt), yo, k, s, p, argmax=am)
    _check(yo.permute(0, 3, 1, 2), y.detach(), dt, "maxpool fwd (argmax variant)", 1e-6, 1e-2)
    T.maxpool_bwd(am, _nhwc(dy, dt), dx, k, s, p)
    _check(dx.permute(0, 3, 1, 2), x.grad, dt, "maxpool bwd", 1e-6, 1e-2)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("hi,ho", [((14, 14), (56, 56)), ((7, 9), (24, 40)), ((28, 28), (56, 56))])
def test_bilinear_bwd(hi, ho, align, dt):
    ops, T = _mods()
    x = _rand(2, 32, *hi, seed=27).requires_grad_(True)
    y = F.interpolate(x, size=ho, mode="bilinear", align_corners=align)
    dy = _q(_rand(*y.shape, seed=28), dt)
    y.backward(dy)
    big = torch.zeros((2, ho[0], ho[1], 48), dtype=dt, device=DEV)
    big[..., :32] = _nhwc(dy, dt)
    dx = torch.empty((2, hi[0], hi[1], 32), dtype=dt, device=DEV)
    T.bilinear_bwd(big[..., :32], dx, align)
    _check(dx.permute(0, 3, 1, 2), x.grad, dt, "bilinear bwd", 5e-5, 1.5e-2)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("C", [2, 22])
def test_bilinear_bwd_from_nchw_and_ce(C, dt):
    ops, T = _mods()
    B = 2
    lo = _rand(2 * B, C, 14, 14, seed=29).requires_grad_(True)
    out = F.interpolate(lo, size=(56, 56), mode="bilinear", align_corners=False)
    label = torch.randint(0, C, (B, 56, 56), generator=torch.Generator().manual_seed(30))
    label[torch.rand((B, 56, 56), generator=torch.Generator().manual_seed(31)) < 0.05] = 255
    loss = F.cross_entropy(out[:B] + out[B:] * 0.0, label, ignore_index=255)
    loss.backward()
    l, dl = T.ce_loss(out.detach().contiguous().to(DEV), label.to(DEV), B)
    assert abs(float(l.item()) - float(loss.item())) <= 1e-5 * max(1.0, abs(float(loss.item())))
    dx = torch.empty((2 * B, 14, 14, C), dtype=dt, device=DEV)
    T.bilinear_bwd_from_nchw(dl, dx, n_valid=B, align_corners=False)
    _check(dx.permute(0, 3, 1, 2), lo.grad, dt, "ce + upsample bwd", 5e-5, 1.5e-2)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("C,ld,hw,HW", [(2, 8, (14, 14), (56, 56)), (22, 24, (9, 13), (35, 50)), (71, 72, (8, 8), (32, 32)),
                                        (2, 2, (7, 7), (7, 7)), (71, 72, (20, 18), (80, 72)), (24, 48, (5, 7), (23, 30)),
                                        (9, 16, (6, 6), (50, 48))])
def test_fused_upsample_ce_head(C, ld, hw, HW, dt):
    """one op == F.interpolate(bilinear) + CrossEntropyLoss(ignore_index) on out[:B] + out[B:]*0 and
    its gradient w.r.t. the low-resolution logits (non-integer ratios, padded channel stride, all-ignored rows)."""
    ops, T = _mods()
    B = 2
    lo = _q(_rand(2 * B, C, *hw, seed=40, scale=2.0), dt).requires_grad_(True)
    out = F.interpolate(lo, size=HW, mode="bilinear", align_corners=False)
    label = torch.randint(0, C, (B, *HW), generator=torch.Generator().manual_seed(41))
    label[torch.rand((B, *HW), generator=torch.Generator().manual_seed(42)) < 0.1] = 255
    label[0, :3] = 255
    loss = F.cross_entropy(out[:B] + out[B:] * 0.0, label, ignore_index=255)
    (loss * 3.0).backward()
    x = torch.full((2 * B, *hw, ld), 7.0, dtype=dt, device=DEV)     # padding columns hold junk: must not be read
    x[..., :C] = lo.detach().permute(0, 2, 3, 1).to(dt).to(DEV)
    l, dlo = T.upsample_ce_head(x, label.to(DEV), B, C, 255, grad_scale=3.0)
    assert abs(float(l.item()) - float(loss.item())) <= 2e-5 * max(1.0, abs(float(loss.item())))
    assert dlo.shape == x.shape and dlo.dtype == dt
    assert float(dlo[..., C:].abs().max()) == 0.0 if ld > C else True
    assert float(dlo[B:].abs().max()) == 0.0
    _check(dlo[..., :C].permute(0, 3, 1, 2), lo.grad, dt, "fused head dlo", 5e-5, 1.5e-2)
    # and against the three-op path it replaces
    outp = torch.empty((2 * B, C, *HW), dtype=torch.float32, device=DEV)
    ops.bilinear_to_nchw(x[..., :C], outp, align_corners=False)
    l3, dl = T.ce_loss(outp, label.to(DEV), B, 255, grad_scale=3.0)
    g3 = torch.zeros_like(x)
    T.bilinear_bwd_from_nchw(dl, g3[..., :C], n_valid=B, align_corners=False)
    assert abs(float(l3.item()) - float(l.item())) <= 2e-6 * max(1.0, abs(float(l.item())))
    _check(dlo, g3.float().cpu(), dt, "fused vs three-op", 2e-5, 1.5e-2)
    # loss only
    l2, none = T.upsample_ce_head(x, label.to(DEV), B, C, 255, want_grad=False)
    assert none is None and float(l2.item()) == float(l.item())
    # everything ignored: loss nan (0 / 0, torch's CrossEntropyLoss(reduction="mean") convention), zero gradient
    l0, d0 = T.upsample_ce_head(x, torch.full_like(label, 255).to(DEV), B, C, 255)
    assert torch.isnan(l0).all() and float(d0.abs().max()) == 0.0
    l0b, d0b = T.ce_loss(outp, torch.full_like(label, 255).to(DEV), B, 255)
    assert torch.isnan(l0b).all() and float(d0b.abs().max()) == 0.0
    # a label outside [0, C) that is not the ignore value (torch device-asserts): no out-of-bounds read, the loss is poisoned
    # with nan and the offending pixels get no gradient
    bad = label.clone()
    bad[0, 0, :3] = C + 5
    bad[0, 1, 0] = -7
    lb1, db1 = T.upsample_ce_head(x, bad.to(DEV), B, C, 255)
    lb2, db2 = T.ce_loss(outp, bad.to(DEV), B, 255)
    assert torch.isnan(lb1).all() and torch.isnan(lb2).all()
    assert torch.isfinite(db1.float()).all() and torch.isfinite(db2).all()
    assert float(db2[0, :, 0, :3].abs().max()) == 0.0 and float(db2[0, :, 1, 0].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_bcast_add_and_smallcin_wgrad(dt):
    ops, T = _mods()
    x = _q(_rand(3, 64, 5, 7, seed=32), dt)
    v = _rand(3, 64, seed=33)
    xv = _nhwc(x, dt)
    T.bcast_add(xv, v.to(DEV), 1.0 / 35)
    _check(xv.permute(0, 3, 1, 2), x + v[:, :, None, None] / 35, dt, "bcast_add")
    # (bf16 / 64 channels = the fused matrix-core kernel of round 6: ragged rows, several 128-pixel runs per row, more items than
    # persistent workgroups (224 x 224), accumulation into a non-zero gradient; f32 = the im2col + GEMM route)
    for cin, stride, hw in [(3, 2, (32, 40)), (1, 1, (24, 16)), (3, 2, (31, 45)), (2, 1, (9, 300)), (3, 2, (224, 224))]:
        xi = _rand(2, cin, *hw, seed=34)
        w = _rand(64, cin, 3, 3, seed=35, scale=0.3).requires_grad_(True)
        y = F.conv2d(xi, w, None, stride, 1)
        dy = _q(_rand(*y.shape, seed=36), dt)
        y.backward(dy)
        dw = torch.full((64, cin, 3, 3), 0.5, device=DEV)
        T.smallcin_wgrad(xi.to(DEV), _nhwc(dy, dt), dw, stride)
        _check(dw - 0.5, w.grad, dt, f"smallcin wgrad {cin} s{stride} {hw}", 1e-4, 1e-2)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("shape", [(4, 56, 56, 64, 64, 3), (2, 28, 28, 128, 512, 1), (3, 13, 11, 64, 48, 1)])
def test_fused_bn_statistics_from_conv_epilogue(shape, dt):
    """per-tile (mean, M2) written by the conv epilogue + Chan combine == batch statistics of the conv output."""
    ops, T = _mods()
    n, h, w, cin, cout, k = shape
    x = _q(_rand(n, cin, h, w, seed=40) + 0.7, dt)
    wt = _q(_rand(cout, cin, k, k, seed=41, scale=(cin * k * k) ** -0.5), dt)
    z = F.conv2d(x, wt, None, 1, k // 2)
    out = torch.empty((n, h, w, cout), dtype=dt, device=DEV)
    out, stats = ops.conv2d(_nhwc(x, dt), ops.pack_weight(wt.to(DEV), dt), out, kh=k, kw=k, pad=k // 2, want_tile_stats=True)
    if stats is None:
        pytest.skip("the library chose split-K for this launch: fused statistics not offered (fallback path is tested elsewhere)")
    ts, tiles, rpt = stats
    g, b = torch.rand(cout, generator=torch.Generator().manual_seed(42)) + 0.5, _rand(cout, seed=43)
    rm, rv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    f = lambda: torch.empty(cout, device=DEV)
    scale, shift, mean, rstd = f(), f(), f(), f()
    T.bn_finalize_tiles(ts, tiles, rpt, n * h * w, g.to(DEV), b.to(DEV), 1e-5, 0.1, rm, rv, scale, shift, mean, rstd)
    m_ref, v_ref = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
    _check(mean, m_ref, torch.float32, "mean", 2e-5)
    _check(rstd, 1 / torch.sqrt(v_ref + 1e-5), torch.float32, "rstd", 5e-5)
    _check(scale, g / torch.sqrt(v_ref + 1e-5), torch.float32, "scale", 5e-5)
    _check(shift, b - m_ref * g / torch.sqrt(v_ref + 1e-5), torch.float32, "shift", 1e-4)
    _check(rm, 0.1 * m_ref, torch.float32, "running_mean", 1e-4)
    _check(rv, 0.9 + 0.1 * z.var((0, 2, 3), unbiased=True), torch.float32, "running_var", 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_weights_multi_matches_single(dtype):
    """cavp_pack_weights_multi == cavp_pack_weight_ohwi + cavp_pack_weight_dgrad per tensor, bit for bit (ragged shapes,
    more tensors than one launch holds)."""
    ops, T = _mods()
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 3, 3, 3), (70, 50, 3, 3), (256, 304, 3, 3), (304, 1216), (1216, 304), (8, 256, 1, 1), (130, 66, 1, 1),
              (24, 40, 7, 7)] * 7   # 56 tensors > 48 per launch
    ws = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    jobs, outs = [], []
    for i, w in enumerate(ws):
        if w.dim() == 2:
            co, ci, kh, kw = w.shape[0], w.shape[1], 1, 1
        else:
            co, ci, kh, kw = w.shape
        o = torch.empty((co, kh, kw, ci), dtype=dtype, device=DEV) if i % 3 != 2 else None
        d = torch.empty((ci, kh, kw, co), dtype=dtype, device=DEV) if i % 3 != 1 else None
        jobs.append((w, o, d))
        outs.append((o, d))
    T.pack_weights_multi(jobs, dtype)
    torch.cuda.synchronize()
    for w, (o, d) in zip(ws, outs):
        if o is not None:
            assert torch.equal(o, ops.pack_weight(w, dtype))
        if d is not None:
            assert torch.equal(d, T.pack_weight_dgrad(w, dtype))


def test_training_entry_points_fail_loudly():
    """Error paths of the training-side C-ABI (round-1 review: only the forward ones were covered): too-small workspace,
    misaligned operands, unsupported strides, bad batch periods, misuse of the auxiliary epilogue tensor, and a deterministic
    -mode scratch that is too small - each a CAVP_ERR_* status surfaced as CavpError, never a silent wrong answer."""
    import ctypes as C
    from cavp_amd import _lib
    from cavp_amd._lib import CavpError, ConvDesc
    ops, T = _mods()
    lib = _lib.load()
    dt = torch.bfloat16
    x = torch.zeros((2, 16, 16, 64), dtype=dt, device=DEV)
    dy = torch.zeros((2, 16, 16, 128), dtype=dt, device=DEV)
    dw = torch.zeros((128, 3, 3, 64), dtype=torch.float32, device=DEV)
    d = ConvDesc(dtype=ops.dtype_code(dt), N=2, H=16, W=16, Cin=64, ldx=64, Cout=128, ldy=128, KH=3, KW=3, stride=1, pad=1, dil=1,
                 ldr=0, act=0, splitk=4, tile=0, up=0, Ho=0, Wo=0, stride_w=0)
    need = lib.cavp_conv2d_wgrad_workspace_bytes(C.byref(d))
    assert need > 0, "the split weight-gradient plan of this shape is expected to need slabs"
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.cavp_conv2d_wgrad_nhwc(C.byref(d), p(x), p(dy), p(dw), None, p(ws), C.c_size_t(need), s) == 0
    st = lib.cavp_conv2d_wgrad_nhwc(C.byref(d), p(x), p(dy), p(dw), None, p(ws), C.c_size_t(need // 2), s)      # workspace too small
    assert st != 0 and b"workspace" in lib.cavp_error_string(st).lower()
    st = lib.cavp_conv2d_wgrad_nhwc(C.byref(d), C.c_void_p(x.data_ptr() + 2), p(dy), p(dw), None, p(ws), C.c_size_t(need), s)  # x not 16-byte aligned
    assert st != 0 and b"align" in lib.cavp_error_string(st).lower()
    assert lib.cavp_conv2d_wgrad_nhwc(C.byref(d), None, p(dy), p(dw), None, p(ws), C.c_size_t(need), s) != 0    # null operand
    with pytest.raises(CavpError):   # dy extent does not match the forward conv
        T.conv2d_wgrad(x, dy[:, :8], dw, kh=3, kw=3, stride=1, pad=1, dil=1)
    with pytest.raises(CavpError):   # dgrad: transposed padding would be negative
        T.conv2d_dgrad(dy, torch.zeros((64, 3, 3, 128), dtype=dt, device=DEV), x, kh=3, kw=3, stride=1, pad=5, dil=1)
    # LayerNorm backward: channel stride not a multiple of the 16-byte vector
    xs = torch.zeros((32, 300), dtype=dt, device=DEV)
    with pytest.raises(CavpError):
        T.layernorm_bwd(xs, xs, torch.ones(300, device=DEV), torch.empty_like(xs), torch.zeros(300, device=DEV),
                        torch.zeros(300, device=DEV), 1e-5)
    # attention gate: the query batch must divide the batch
    q = torch.zeros((3, 8, 304), dtype=dt, device=DEV)
    kv = torch.zeros((4, 304), dtype=dt, device=DEV)
    with pytest.raises(CavpError):
        ops.attn_gate(q, kv, kv, torch.empty((4, 8, 304), dtype=dt, device=DEV), torch.empty((4, 4, 8), device=DEV), 4, 0.1)
    # auxiliary epilogue tensor: derivative output without GELU, and a periodic residual that is not a multiple of 256 rows
    w = torch.zeros((128, 1, 1, 64), dtype=dt, device=DEV)
    y = torch.empty((2, 16, 16, 128), dtype=dt, device=DEV)
    with pytest.raises(CavpError):
        ops.conv2d(x, w, y, aux=torch.empty_like(y), aux_mode=1, act=ops.ACT_RELU)
    with pytest.raises(CavpError):
        ops.conv2d(x, w, y, residual=y[:1, :10].contiguous(), res_rows=160)
    # deterministic mode with a scratch buffer smaller than one launch's partials
    small = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    assert lib.cavp_set_deterministic(p(small), C.c_size_t(small.numel())) == 0
    try:
        big = torch.zeros((200704, 304), dtype=dt, device=DEV)
        with pytest.raises(CavpError):
            T.layernorm_bwd(big, big, torch.ones(304, device=DEV), torch.empty_like(big), torch.zeros(304, device=DEV),
                            torch.zeros(304, device=DEV), 1e-5)
    finally:
        assert lib.cavp_set_deterministic(None, C.c_size_t(0)) == 0
    assert lib.cavp_set_deterministic(C.c_void_p(small.data_ptr() + 4), C.c_size_t(1 << 20)) != 0   # misaligned scratch


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_colsum_groups(dt):
    """Per-image column sums in one launch (the ASPP pooled branch's per-image bias gradient), accumulating, on a channel slice."""
    ops, T = _mods()
    G, H, W, C = 5, 14, 13, 256
    x = _q(_rand(G, H, W, C + 16, seed=40), dt)
    xd = x.to(dt).to(DEV)[..., 8:8 + C]
    out = torch.full((G, C), 0.5, device=DEV)
    T.colsum_groups(xd, out)
    ref = x[..., 8:8 + C].double().sum((1, 2)).float() + 0.5
    _check(out, ref, torch.float32, "colsum_groups", 1e-4, 1e-4)


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("rows,C", [(32, 256), (1000, 64), (6272, 1024), (130, 48)], ids=["pooled_B32", "ragged", "aspp_slice", "narrow"])
def test_col_tile_stats(rows, C, dt):
    """One-pass per-tile (mean, M2) of a channel slice + cavp_bn_finalize_tiles == mean / biased variance of the tensor, also for a
    large common offset (the cancellation case) and the M = B rows of the ASPP pooled branch."""
    ops, T = _mods()
    if C % (8 if dt == torch.bfloat16 else 4):
        pytest.skip("channel count must be a multiple of the 16-byte vector")
    x = _q(_rand(rows, C + 16, seed=41) * 0.5 + 30.0, dt)
    xd = x.to(dt).to(DEV)[:, 8:8 + C]
    ts, tiles, rpt = T.col_tile_stats(xd)
    assert rpt == 128 and tiles == (rows + 127) // 128
    scale, shift, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
    T.bn_finalize_tiles(ts, tiles, rpt, rows, torch.ones(C, device=DEV), torch.zeros(C, device=DEV), 1e-5, 0.1, None, None, scale, shift,
                        mean, rstd)
    ref = x[:, 8:8 + C].double()
    assert float((mean.cpu().double() - ref.mean(0)).abs().max()) <= 1e-4
    var = ref.var(0, unbiased=False)
    got_var = 1.0 / rstd.cpu().double() ** 2 - 1e-5
    assert float(((got_var - var).abs() / var).max()) <= 2e-4


BNB_CASES = [
    # name, N, H, W, Cin, Cout, k, s, p, d, with_out (mask from the activation output + accumulated residual: the bottleneck-output case)
    ("1x1_mid", 8, 56, 56, 64, 256, 1, 1, 0, 1, False),
    ("3x3_mid", 8, 56, 56, 64, 64, 3, 1, 1, 1, False),
    ("3x3_s2", 4, 56, 56, 128, 128, 3, 2, 1, 1, False),
    ("1x1_out_res", 8, 56, 56, 256, 64, 1, 1, 0, 1, True),
    ("1x1_s2_out_res", 4, 56, 56, 256, 128, 1, 2, 0, 1, True),
    ("ragged_1x1", 3, 13, 11, 72, 48, 1, 1, 0, 1, False),
    ("3x3_s2_odd", 3, 15, 13, 64, 64, 3, 2, 1, 1, False),
    ("1x1_s2_odd_out_res", 3, 15, 13, 128, 64, 1, 2, 0, 1, True),
    ("wide_14", 8, 14, 14, 1024, 256, 1, 1, 0, 1, True),
]


@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", BNB_CASES, ids=[c[0] for c in BNB_CASES])
def test_conv_dgrad_fused_bn_backward_stats(case, dt):
    """cavp_conv2d_nhwc_bnbwd: the data-gradient launch that produces the gradient of a BatchNorm + ReLU output stores
    g = (dgrad + residual) * relu'(.) and emits the per-tile sums of cavp_bn_act_bwd_reduce.  Checked against the plain data
    gradient followed by torch on the host: g itself, sum g and sum g * zhat (after cavp_bn_bwd_sum_tiles), and - end to end - the
    BatchNorm input gradient cavp_bn_act_bwd_apply makes from them against torch.autograd through conv -> BN(train) -> ReLU
    (resnet.py:75-98)."""
    ops, T = _mods()
    name, n, h, w, cin, cout, k, s, p, d, with_out = case
    if cin % (8 if dt == torch.bfloat16 else 4):
        pytest.skip("channel count must be a multiple of the 16-byte vector")
    # forward chain on the host: a = relu(bn(z) [+ skip]); y = conv(a)
    z = _q(_rand(n, cin, h, w, seed=61) * 0.8 + 0.3, dt)
    gam, bet = _rand(cin, seed=62) * 0.3 + 1.0, _rand(cin, seed=63) * 0.3
    skip = _q(_rand(n, cin, h, w, seed=64), dt) if with_out else None
    mean = z.mean((0, 2, 3))
    var = z.var((0, 2, 3), unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    scale, shift = gam * rstd, bet - mean * gam * rstd
    bn = F.batch_norm(z, None, None, gam, bet, True, 0.1, 1e-5)
    a = F.relu(bn + skip) if with_out else F.relu(bn)
    a_q = _q(a.detach(), dt)
    wt = _q(_rand(cout, cin, k, k, seed=65, scale=(cin * k * k) ** -0.5), dt)
    y = F.conv2d(a, wt, None, s, p, d)
    dy = _q(_rand(*y.shape, seed=66), dt)
    prev = _q(_rand(n, cin, h, w, seed=67) * 0.5, dt) if with_out else None    # the skip path's gradient already in a.g
    # plain launch: gradient of a (+ what the skip path already put there)
    dyv = _nhwc(dy, dt)
    wT = T.pack_weight_dgrad(wt.to(DEV), dt)
    plain = torch.empty((n, h, w, cin), dtype=dt, device=DEV)
    T.conv2d_dgrad(dyv, wT, plain, kh=k, kw=k, stride=s, pad=p, dil=d, residual=_nhwc(prev, dt) if prev is not None else None)
    # fused launch
    zv = _nhwc(z, dt)
    outv = _nhwc(a_q, dt) if with_out else None
    g = torch.empty((n, h, w, cin), dtype=dt, device=DEV)
    mean_d, rstd_d, sc_d, sh_d = (t.float().to(DEV) for t in (mean, rstd, scale, shift))
    r = T.conv2d_dgrad(dyv, wT, g, kh=k, kw=k, stride=s, pad=p, dil=d, residual=_nhwc(prev, dt) if prev is not None else None,
                       bnb=dict(z=zv, out=outv, scale=sc_d, shift=sh_d, mean=mean_d, rstd=rstd_d, act=ops.ACT_RELU))
    if r is None:   # (the planner split this launch over K or chose a tile without the fused epilogue: the caller keeps the separate reduce)
        assert name not in ("3x3_mid", "1x1_out_res"), "the backbone-shaped cases must carry the statistics"
        pytest.skip("launch cannot carry the fused statistics")
    part, tiles = r
    # reference from the plain gradient (same storage rounding): mask from the stored activation / from z*scale + shift in f32
    pl = plain.float().cpu()
    zc = zv.float().cpu()
    if with_out:
        mask = (outv.float().cpu() > 0).float()
    else:
        mask = ((zc * scale.float() + shift.float()) > 0).float()
    g_ref = pl * mask
    got_g = g.float().cpu()
    tol = 1e-5 if dt == torch.float32 else 1e-2
    # (a pixel whose pre-activation rounds to +-0 may flip its mask between the fused multiply-add on the device and the host's two roundings)
    bad = ((got_g - g_ref).abs() > tol * max(1.0, float(g_ref.abs().max()))).float().mean()
    assert float(bad) <= 1e-4, f"{name}: {float(bad):.2e} of the masked gradient elements differ"
    sums = torch.zeros(2, cin, device=DEV)
    sums[0].fill_(0.25)   # cavp_bn_bwd_sum_tiles ADDS
    T.bn_bwd_sum_tiles(part, tiles, sums[0], sums[1])
    gq = got_g.double()   # the sums are taken from the f32 values BEFORE storage rounding; compare with a tolerance that covers it
    zhat = (zc.double() - mean.double()) * rstd.double()
    s0 = gq.sum((0, 1, 2)) + 0.25
    s1 = (gq * zhat).sum((0, 1, 2))
    rows = n * h * w
    stol = (2e-4 if dt == torch.float32 else 6e-3) * (rows ** 0.5) * max(1.0, float(gq.abs().max()))
    assert float((sums[0].cpu().double() - s0).abs().max()) <= stol, (name, float((sums[0].cpu().double() - s0).abs().max()), stol)
    assert float((sums[1].cpu().double() - s1).abs().max()) <= 3 * stol, (name, float((sums[1].cpu().double() - s1).abs().max()), stol)
    # end to end: dz from the fused route == dz from the separate reduce on the plain gradient
    sums_ref = torch.zeros(2, cin, device=DEV)
    T.bn_act_bwd_reduce(plain, outv, zv, mean_d, rstd_d, ops.ACT_RELU, sums_ref[0], sums_ref[1], fwd_scale=sc_d, fwd_shift=sh_d)
    sums[0] -= 0.25
    dz_ref, dz = torch.empty_like(plain), torch.empty_like(plain)
    T.bn_act_bwd_apply(plain, outv, zv, mean_d, rstd_d, gam.to(DEV), sums_ref[0], sums_ref[1], ops.ACT_RELU, dz_ref, fwd_scale=sc_d, fwd_shift=sh_d)
    T.bn_act_bwd_apply(g, None, zv, mean_d, rstd_d, gam.to(DEV), sums[0], sums[1], ops.ACT_NONE, dz)
    _check(dz, dz_ref.float().cpu(), dt, name + ".dz fused vs separate", 2e-4, 2e-2)
