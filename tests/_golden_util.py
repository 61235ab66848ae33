"""Helpers shared by the CPU (oracle) and GPU (product) golden tests."""
import os

import numpy as np
import torch

NSAMP = 4096
HERE = os.path.dirname(os.path.abspath(__file__))


def load_case(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=True)
    C, B, H, W = [int(v) for v in z["cfg/CBHW"]]
    return z, dict(C=C, B=B, hw=(H, W), lds=[bool(v) for v in z["cfg/lds"]], train=bool(z["cfg/train"][0]))


def sample(t):
    """Same fixed strided sample as tools/make_golden.py."""
    t = t.detach().to(torch.float32).cpu().contiguous().flatten()
    n = t.numel()
    stride = max(1, n // NSAMP)
    return t[::stride][:NSAMP].numpy(), np.array([t.double().sum().item(), t.double().abs().sum().item(), n])


def check_tap(z, key, t, atol, rtol=0.0, what=""):
    """Compare tensor `t` (logical NCHW / reference shape) with the golden sample + checksums."""
    assert tuple(t.shape) == tuple(int(v) for v in z["shape/" + key]), (key, t.shape, z["shape/" + key])
    s, c = sample(t)
    ref = z["sample/" + key]
    err = np.abs(s - ref)
    tol = atol + rtol * np.abs(ref)
    assert np.all(err <= tol), f"{what}{key}: max err {err.max():.3e} (tol {atol:g}+{rtol:g}*|ref|), ref max {np.abs(ref).max():.3f}"
    rc = z["cksum/" + key]
    # mean-abs checksum: catches errors outside the strided sample
    mean_abs_err = abs(c[1] - rc[1]) / rc[2]
    assert mean_abs_err <= atol + rtol * rc[1] / rc[2], f"{what}{key}: abs-sum checksum off by {mean_abs_err:.3e}/elem"
    return float(err.max())
