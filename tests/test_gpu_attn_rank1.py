"""The one-key attention collapse (csrc/attn_rank1.hip; attn.py:73-106 + the block's residual, attn.py:153-156, as CAVP calls it
with a single audio token per batch item) against a plain PyTorch f32 evaluation of the reference's formulation: q projection,
per-head sigmoid(scale q k^T), gate times v, output projection + bias + residual - forward, and every gradient (tokens summed over
the batch items that share them, Wq, Wp, bp, k, v) through torch.autograd."""
import pytest
import torch

from cavp_amd import ops
from cavp_amd import train_ops as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(x, k, v, wq, wp, bp, heads, scale, reps):
    xb, t, c = x.shape
    b, d = k.shape[0], c // heads
    xr = x.repeat(reps, 1, 1)                                     # cavp_model.py:181: torch.cat((fea_v, fea_v.clone()))
    q = (xr @ wq.t()).view(b, t, heads, d)
    s = (q * k.view(b, 1, heads, d)).sum(-1) * scale              # [B, T, H]: q @ k^T with ONE key
    g = torch.sigmoid(s)
    o = (g.unsqueeze(-1) * v.view(b, 1, heads, d)).reshape(b, t, c)
    return xr + o @ wp.t() + bp, g.permute(0, 2, 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("xb,reps,t,c", [(3, 2, 197, 304), (2, 1, 64, 112), (1, 3, 33, 512), (4, 2, 3136, 304)])
def test_attn_rank1_forward_backward(dtype, xb, reps, t, c):
    heads, b = 4, xb * reps
    scale = (c // heads) ** -0.5
    g = torch.Generator(device="cpu").manual_seed(1234 + c + t)
    x = torch.randn((xb, t, c), generator=g)
    k, v = torch.randn((b, c), generator=g), torch.randn((b, c), generator=g)
    wq, wp = torch.randn((c, c), generator=g) * c ** -0.5, torch.randn((c, c), generator=g) * c ** -0.5
    bp = torch.randn(c, generator=g) * 0.1
    dout = torch.randn((b, t, c), generator=g)
    # what the kernels see: activations in the compute dtype, weights f32
    xd, kd, vd, dd = (z.to(DEV).to(dtype) for z in (x, k, v, dout))
    wqd, wpd, bpd = wq.to(DEV), wp.to(DEV), bp.to(DEV)
    # reference on the SAME (rounded) inputs, f64 accumulate-free f32 math on the GPU through autograd
    rx, rk, rv = (z.float().clone().requires_grad_(True) for z in (xd, kd, vd))
    rwq, rwp, rbp = (z.clone().requires_grad_(True) for z in (wqd, wpd, bpd))
    ref_out, ref_attn = _reference(rx, rk, rv, rwq, rwp, rbp, heads, scale, reps)
    (ref_out * dd.float()).sum().backward()

    assert ops.attn1_supported(c, heads)
    u, pm = ops.attn1_prepare(wqd, wpd, kd, vd, heads, scale)
    out = torch.empty((b, t, c), dtype=dtype, device=DEV)
    attn = torch.empty((b, heads, t), dtype=torch.float32, device=DEV)
    ops.attn1_fwd(xd, u, pm, bpd, out, attn)
    dx = torch.empty_like(xd)
    dbp = torch.zeros(c, dtype=torch.float32, device=DEV)
    du, dp = T.attn1_bwd(dd, xd, u, pm, dx, dbp)
    dwq, dwp = torch.zeros_like(wqd), torch.zeros_like(wpd)
    dk, dv = T.attn1_finish(wqd, wpd, kd, vd, du, dp, dwq, dwp, heads, scale)
    torch.cuda.synchronize()

    def rel(a, r):
        a, r = a.detach().double(), r.detach().double()
        return float((a - r).norm() / max(float(r.norm()), 1e-30))
    tol_o = 2e-6 if dtype == torch.float32 else 6e-3      # outputs / dx are stored in the compute dtype
    tol_g = 2e-5 if dtype == torch.float32 else 2e-5      # parameter / key / value gradients are f32 sums of the same products
    assert rel(out.float(), ref_out) <= tol_o, rel(out.float(), ref_out)
    assert float((attn - ref_attn).abs().max()) <= 2e-6
    assert rel(dx.float(), rx.grad) <= tol_o, rel(dx.float(), rx.grad)
    for name, got, want in (("dWq", dwq, rwq.grad), ("dWp", dwp, rwp.grad), ("dbp", dbp, rbp.grad), ("dk", dk, rk.grad), ("dv", dv, rv.grad)):
        assert rel(got, want) <= tol_g, (name, rel(got, want))
    # accumulate semantics of the parameter gradients (the flat arena is zeroed once per step, contributions add up)
    dk2, dv2 = T.attn1_finish(wqd, wpd, kd, vd, du, dp, dwq, dwp, heads, scale)
    torch.cuda.synchronize()
    assert rel(dwq, 2 * rwq.grad) <= tol_g and rel(dwp, 2 * rwp.grad) <= tol_g and torch.equal(dk2, dk)


def test_attn_rank1_unsupported_shapes_fall_back():
    assert not ops.attn1_supported(304, 8) and not ops.attn1_supported(520, 4) and not ops.attn1_supported(300, 4)
    assert ops.attn1_supported(304, 4) and ops.attn1_supported(112, 4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_model_step_same_with_and_without_the_collapse(dtype, deterministic):
    """The whole training step with the one-key collapse against the same step on the reference's formulation (q GEMM, gate
    kernel, proj GEMM): loss, logits and every parameter gradient."""
    import types

    import cavp_amd.train as TR
    from cavp_amd.cavp_model import CAVP
    from cavp_amd.synth import synth_inputs, synth_state_dict
    C, B, hw = 3, 4, (64, 64)
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=[False, False, False], audio_backbone="vgg",
                                 num_classes=C, batch_size=B, local_rank="cpu")
    image, audio, label = [z.to(DEV) for z in synth_inputs(B, hw, audio_batch=2 * B, num_classes=C, seed=3)]
    res = {}
    for on in (False, True):
        TR._RANK1_ATTN = on
        try:
            m = CAVP(50, None, num_classes=C, args=args)
            m.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1), strict=True)
            m.train().to(DEV).set_compute_dtype(dtype)
            loss = m.train_step(image, audio, label, want_pred=True)
            torch.cuda.synchronize()
            res[on] = (float(loss.item()), m._last_outputs[0].float().clone(),
                       {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
            m.eval()
            with torch.no_grad():
                res[on] += (m(image, audio[:B], eval_mode=True)[0].float().clone(),)
        finally:
            TR._RANK1_ATTN = True
    (l0, o0, g0, e0), (l1, o1, g1, e1) = res[False], res[True]
    f32 = dtype == torch.float32
    assert abs(l0 - l1) <= (1e-5 if f32 else 2e-2) * max(1.0, abs(l0)), (l0, l1)
    assert float((o0 - o1).norm() / o0.norm()) <= (1e-4 if f32 else 0.5)
    assert float((e0 - e1).norm() / e0.norm()) <= (1e-5 if f32 else 5e-2)
    assert g0.keys() == g1.keys()
    if f32:   # (bf16: the batch-statistics trunk amplifies the different rounding of q / o, as everywhere on synthetic weights)
        for k in g0:
            n = float(g0[k].double().norm())
            if n > 0:
                assert float((g0[k].double() - g1[k].double()).norm()) <= 5e-3 * n, (k, float((g0[k] - g1[k]).norm()) / n)
