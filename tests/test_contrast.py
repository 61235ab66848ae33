"""ContrastLoss (loss/contrastive_aud.py): oracle pinned to the reference's golden output (CPU), host-side sampling
equal to the oracle's, and the MI355X path (gpu) against both."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))


def _inputs():
    # same generator as tools/make_golden.py::contrast_inputs (duplicated here: tools/ imports the reference)
    g = torch.Generator().manual_seed(7)
    B, C, hw, full, nc = 2, 304, (56, 56), (224, 224), 4
    em = torch.randn((B, C) + hw, generator=g)
    es = em * 0.5 + torch.randn((B, C) + hw, generator=g)
    gt = torch.zeros((B,) + full, dtype=torch.long)
    for b in range(B):
        for k in range(1, nc):
            h0 = int(torch.randint(0, full[0] - 100, (1,), generator=g)); w0 = int(torch.randint(0, full[1] - 120, (1,), generator=g))
            gt[b, h0:h0 + 60 + 20 * k, w0:w0 + 120] = k
        gt[b, :8, :] = 255
    gs = gt.clone()
    gs[1:] = 0
    return em, gt, es, gs


def _sample(t):
    t = t.detach().float().cpu().contiguous().flatten()
    return t[:: max(1, t.numel() // 4096)][:4096].numpy()


def test_oracle_matches_reference_golden():
    from oracle.contrast_oracle import contrast_loss
    z = np.load(os.path.join(HERE, "golden", "contrast.npz"))
    em, gt, es, gs = _inputs()
    em.requires_grad_(True); es.requires_grad_(True)
    torch.manual_seed(1234)
    loss = contrast_loss(em, gt, es, gs, 0.1, 255, 512)
    loss.backward()
    assert abs(loss.item() - float(z["loss"][0])) <= 1e-6
    assert np.abs(_sample(em.grad) - z["sample/d_match"]).max() <= 1e-9 + 1e-5 * np.abs(z["sample/d_match"]).max()
    assert np.abs(_sample(es.grad) - z["sample/d_shuffle"]).max() <= 1e-9 + 1e-5 * np.abs(z["sample/d_shuffle"]).max()


def test_host_sampling_consumes_rng_like_reference():
    """cavp_amd.contrast.sample_anchors picks the oracle's anchors for the same RNG state (labels, order, count)."""
    from cavp_amd.contrast import downsample_labels, sample_anchors
    import torch.nn.functional as F
    em, gt, es, gs = _inputs()
    torch.manual_seed(99)
    plan = sample_anchors(downsample_labels(gt, (56, 56)), downsample_labels(gs, (56, 56)), 255, 512)
    # oracle-side recomputation of the label sequence with the same seed
    torch.manual_seed(99)
    gm = F.interpolate(gt.unsqueeze(1).float(), size=(56, 56), mode="nearest").squeeze(1).long().flatten(1)
    gsd = F.interpolate(gs.unsqueeze(1).float(), size=(56, 56), mode="nearest").squeeze(1).long().flatten(1)
    fg = (gm > 0) & (gm != 255)
    g_fg = gm[fg]
    labs = []
    for item in torch.unique(g_fg):
        cur = g_fg == item
        if int(cur.sum()) < 512:
            continue
        labs.append(g_fg[cur][torch.randperm(int(cur.sum()))][:512])
    n_bg = int((gm == 0).sum())
    k = min(512, int(fg.sum()), n_bg)
    i1, i2 = torch.randperm(n_bg), torch.randperm(int(fg.sum()))
    labs += [torch.zeros(k, dtype=torch.long), gsd[fg][i2][:k]]
    ref = torch.cat(labs).numpy()
    assert plan.n == ref.shape[0] and plan.n_match == ref.shape[0] - k
    assert np.array_equal(plan.labels, ref.astype(np.int32))
    # anchor coordinates address the right label
    gmn = gm.numpy()
    assert np.array_equal(gmn[plan.b[:plan.n_match], plan.p[:plan.n_match]][: plan.n_match - k] > 0, np.ones(plan.n_match - k, bool))


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["nhwc_view", "nchw"])
def test_gpu_contrast_loss_matches_reference(layout):
    from cavp_amd.contrast import ContrastLoss
    z = np.load(os.path.join(HERE, "golden", "contrast.npz"))
    em, gt, es, gs = _inputs()
    dev = "cuda:0"
    if layout == "nhwc_view":    # what CAVP returns: NCHW-shaped views of NHWC memory
        emd = em.permute(0, 2, 3, 1).contiguous().to(dev).permute(0, 3, 1, 2).requires_grad_(True)
        esd = es.permute(0, 2, 3, 1).contiguous().to(dev).permute(0, 3, 1, 2).requires_grad_(True)
    else:
        emd, esd = em.to(dev).requires_grad_(True), es.to(dev).requires_grad_(True)
    crit = ContrastLoss(temperature=0.1, ignore_idx=255, max_views=512)
    torch.manual_seed(1234)
    loss = crit(emd, gt.to(dev), esd, gs.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - float(z["loss"][0])) <= 2e-5 * max(1.0, abs(float(z["loss"][0])))
    for k, g in (("d_match", emd.grad), ("d_shuffle", esd.grad)):
        ref = z["sample/" + k]
        err = np.abs(_sample(g) - ref).max()
        assert err <= 2e-4 * np.abs(ref).max() + 1e-10, (k, err, np.abs(ref).max())
        nnz = int((g.abs().sum(1) > 0).sum().item())
        assert nnz == int(z["nnz_pixels/" + k][0]), (k, nnz)
