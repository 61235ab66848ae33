"""The 256 x 256 weight-gradient tile (csrc/conv_wgrad_big.hip, cavp_set_wgrad_big) vs torch.autograd on CPU (fp32 reference of
the bf16-rounded operands): the weight gradients of encoder_decoder.py:62-75 / attn.py:136-143 / cavp_model.py:123-128.  -m gpu."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_train_ops import CONV, DEV, _check, _nhwc, _q, _rand

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _mods():
    from cavp_amd import _lib, train_ops
    return _lib.load(), train_ops


@pytest.fixture
def big():
    """set(mode, stagger) -> None; the process-wide switch is restored afterwards"""
    lib, _ = _mods()

    def set_(mode, stagger=2):
        assert lib.cavp_set_wgrad_big(mode, stagger) == 0
    yield set_
    assert lib.cavp_set_wgrad_big(0, 2) == 0


def _case(name, n, h, w, cin, cout, k, s, p, d, seed=0):
    x = _q(_rand(n, cin, h, w, seed=seed + 1), BF).requires_grad_(True)
    wt = _q(_rand(cout, cin, k, k, seed=seed + 2, scale=(cin * k * k) ** -0.5), BF).requires_grad_(True)
    y = F.conv2d(x, wt, None, s, p, d)
    dy = _q(_rand(*y.shape, seed=seed + 3), BF)
    y.backward(dy)
    return _nhwc(x.detach(), BF), _nhwc(dy, BF), wt.grad, dy.sum(dim=(0, 2, 3))


@pytest.mark.parametrize("stagger", [2, 1, 0], ids=["16waves", "8waves_interleaved", "8waves_plain"])
@pytest.mark.parametrize("case", CONV, ids=[c[0] for c in CONV])
def test_big_tile_forced_on_small_shapes(case, stagger, big):
    """every layer shape of the op tests (strides, dilations with dead taps, 304 / 48 channels: tiles with dead 32-channel blocks,
    pixel ranges far shorter than one ring trip) through the big tile: both layouts, accumulate / overwrite, bias, split counts"""
    lib, T = _mods()
    name, n, h, w, cin, cout, k, s, p, d = case
    xv, dyv, gw, gb = _case(*case)
    big(2, stagger)
    for sk in (0, 1, 3):
        for oihw in (False, True):
            shape = (cout, cin, k, k) if oihw else (cout, k, k, cin)
            dw = torch.full(shape, 0.25, dtype=torch.float32, device=DEV)
            db = torch.full((cout,), 0.5, dtype=torch.float32, device=DEV)
            T.conv2d_wgrad(xv, dyv, dw, kh=k, kw=k, stride=s, pad=p, dil=d, dw_oihw=oihw, splitk=sk, dbias=db)
            got = dw if oihw else dw.permute(0, 3, 1, 2)
            _check(got, gw + 0.25, BF, f"{name}.big.sk{sk}.oihw{oihw}", bf16_tol=2e-2)
            _check(db, gb + 0.5, BF, f"{name}.big.dbias.sk{sk}", bf16_tol=2e-2)
            # beta = 0: garbage in the destination is overwritten - bit-identical to accumulating onto zeros
            ga = torch.zeros(shape, dtype=torch.float32, device=DEV)
            ov = torch.full(shape, 7.5, dtype=torch.float32, device=DEV)
            T.conv2d_wgrad(xv, dyv, ga, kh=k, kw=k, stride=s, pad=p, dil=d, dw_oihw=oihw, splitk=sk)
            T.conv2d_wgrad(xv, dyv, ov, kh=k, kw=k, stride=s, pad=p, dil=d, dw_oihw=oihw, splitk=sk, overwrite=True)
            assert torch.equal(ga, ov), f"{name}.big.overwrite.sk{sk}.oihw{oihw}"


BIG = [
    # name, N, H, W, Cin, Cout, k, s, p, d: >= 16384 pixel rows and >= 192 channels -> the automatic choice
    ("head0_like", 2, 96, 96, 304, 256, 3, 1, 1, 1),
    ("token_fc1_like", 1, 1, 20000, 304, 1216, 1, 1, 0, 1),
    ("token_fc2_like", 1, 1, 16500, 1216, 304, 1, 1, 0, 1),
    ("s2_512", 1, 260, 260, 256, 512, 3, 2, 1, 1),
]


@pytest.mark.parametrize("case", BIG, ids=[c[0] for c in BIG])
def test_big_tile_auto_vs_torch_and_small_tile(case, big):
    lib, T = _mods()
    name, n, h, w, cin, cout, k, s, p, d = case
    xv, dyv, gw, gb = _case(*case, seed=40)
    outs = {}
    for mode, stagger in ((0, 2), (0, 1), (0, 0), (1, 1)):
        big(mode, stagger)
        dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device=DEV)
        db = torch.zeros((cout,), dtype=torch.float32, device=DEV)
        T.conv2d_wgrad(xv, dyv, dw, kh=k, kw=k, stride=s, pad=p, dil=d, dbias=db)
        _check(dw.permute(0, 3, 1, 2), gw, BF, f"{name}.mode{mode}.stagger{stagger}", bf16_tol=1e-2)
        _check(db, gb, BF, f"{name}.mode{mode}.dbias", bf16_tol=1e-2)
        outs[(mode, stagger)] = (dw, db)
    # the schedule does not change a single product or the order they are added in
    for other in ((0, 1), (0, 0)):
        assert torch.equal(outs[(0, 2)][0], outs[other][0]) and torch.equal(outs[(0, 2)][1], outs[other][1])
    # the two tiles compute the same f32 sums in a different association: equal to f32 rounding
    a, b = outs[(0, 2)][0], outs[(1, 1)][0]
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


def test_big_tile_group_matches_single_launch_bitwise(big):
    """a grouped launch that mixes jobs of both tiles: each job bit-identical to its single launch at the same split count"""
    lib, T = _mods()
    jobs = []
    for i, (name, n, h, w, cin, cout, k, s, p, d) in enumerate([BIG[0], CONV[0], BIG[1], CONV[8], CONV[2]]):
        xv, dyv, gw, gb = _case(name, n, h, w, cin, cout, k, s, p, d, seed=60 + 5 * i)
        oihw, over = bool(i & 1), bool(i & 2)
        shape = (cout, cin, k, k) if oihw else (cout, k, k, cin)
        dw = torch.full(shape, 7.5 if over else 0.25, dtype=torch.float32, device=DEV)
        db = torch.full((cout,), 0.5, dtype=torch.float32, device=DEV) if i % 2 == 0 else None
        jobs.append(dict(x=xv, dy=dyv, dw=dw, kh=k, kw=k, stride=s, pad=p, dil=d, dbias=db, dw_oihw=oihw, overwrite=over,
                         splitk=(5, 1, 7, 3, 2)[i], _ref=(name, gw, gb)))
    big(0, 2)
    T.conv2d_wgrad_group([{k: v for k, v in j.items() if k != "_ref"} for j in jobs])
    for j in jobs:
        name, gw, gb = j["_ref"]
        got = j["dw"] if j["dw_oihw"] else j["dw"].permute(0, 3, 1, 2)
        _check(got, gw + (0.0 if j["overwrite"] else 0.25), BF, name + ".group", bf16_tol=1e-2)
        if j["dbias"] is not None:
            _check(j["dbias"], gb + 0.5, BF, name + ".group.dbias", bf16_tol=1e-2)
        single = torch.full_like(j["dw"], 7.5 if j["overwrite"] else 0.25)
        T.conv2d_wgrad(j["x"], j["dy"], single, kh=j["kh"], kw=j["kw"], stride=j["stride"], pad=j["pad"], dil=j["dil"],
                       dw_oihw=j["dw_oihw"], overwrite=j["overwrite"], splitk=j["splitk"])
        assert torch.equal(single, j["dw"]), name + ": group vs single launch"


def test_big_tile_is_deterministic(big):
    lib, T = _mods()
    name, n, h, w, cin, cout, k, s, p, d = BIG[0]
    xv, dyv, gw, gb = _case(*BIG[0], seed=90)
    big(0, 2)
    outs = []
    for _ in range(3):
        dw = torch.empty((cout, k, k, cin), dtype=torch.float32, device=DEV)
        T.conv2d_wgrad(xv, dyv, dw, kh=k, kw=k, stride=s, pad=p, dil=d, overwrite=True)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
