"""Log-mel front-end: oracle/mel_oracle.py pinned on CPU (STFT half against an independent numpy
DFT, filterbank against its closed form and against the tables the product builds), the HIP kernel against the oracle."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mel_oracle  # noqa: E402


def _wave(n, c, a, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(a, dtype=torch.float32) / 16000.0
    tone = 0.3 * torch.sin(2 * math.pi * 440.0 * t) + 0.1 * torch.sin(2 * math.pi * 2500.0 * t)
    return (torch.randn(n, c, a, generator=g) * 0.05 + tone).clamp(-1, 1)


def test_filterbank_closed_form():
    fb = mel_oracle.melscale_fbanks(257, 125.0, 3800.0, 64, 16000)
    assert fb.shape == (257, 64) and fb.dtype == torch.float32
    assert float(fb.min()) >= 0.0 and float(fb.max()) <= 1.0 + 1e-6
    freqs = np.linspace(0, 8000, 257)
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    pts = 700.0 * (10.0 ** (np.linspace(mel(125.0), mel(3800.0), 66) / 2595.0) - 1.0)
    ref = np.zeros((257, 64))
    for m in range(64):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        ref[:, m] = np.maximum(0.0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)))
    np.testing.assert_allclose(fb.numpy(), ref, atol=2e-5)
    assert float(fb[freqs < 125.0].abs().max()) == 0.0 and float(fb[freqs > 3800.0].abs().max()) == 0.0
    # the tables the product builds are the oracle's, bit for bit
    from cavp_amd.audio_frontend import hann_window_padded, mel_filterbank
    assert torch.equal(mel_filterbank(257, 125.0, 3800.0, 64, 16000), fb)
    w = hann_window_padded(400, 512)
    assert float(w[:56].abs().max()) == 0.0 and float(w[456:].abs().max()) == 0.0
    assert torch.equal(w[56:456], torch.hann_window(400))


def test_oracle_against_numpy_dft():
    """independent restatement of the whole chain in float64 numpy (explicit reflect padding, rfft)"""
    wave = _wave(2, 1, 16000, seed=3)
    got = mel_oracle.preprocess_audio(wave).numpy()
    x = wave.reshape(2, -1).double().numpy()
    pad = np.pad(x, ((0, 0), (256, 256)), mode="reflect")
    win = np.zeros(512)
    win[56:456] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 400.0)
    fb = mel_oracle.melscale_fbanks(257, 125.0, 3800.0, 64, 16000).double().numpy()
    out = np.zeros((2, 96, 64))
    for t in range(96):
        fr = pad[:, t * 160:t * 160 + 512] * win
        p = np.abs(np.fft.rfft(fr, axis=-1)) ** 2
        out[:, t] = 2.0 * (20.0 * np.log10(np.maximum(1e-5, p @ fb)) + 100.0) / 200.0 - 1.0
    np.testing.assert_allclose(got[:, 0], out, atol=2e-5)
    assert got.shape == (2, 1, 96, 64)


def test_oracle_tone_lands_in_the_right_bin():
    t = torch.arange(16000, dtype=torch.float32) / 16000.0
    wave = (0.5 * torch.sin(2 * math.pi * 1000.0 * t)).view(1, 1, -1)
    out = mel_oracle.preprocess_audio(wave)[0, 0]
    mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    centres = 700.0 * (10.0 ** (np.linspace(mel(125.0), mel(3800.0), 66)[1:-1] / 2595.0) - 1.0)
    assert int(out[48].argmax()) == int(np.abs(centres - 1000.0).argmin())
    assert float(out.max()) <= 1.0 and float(out.min()) >= -1.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,a,audio_len", [(3, 1, 16000, 1.0), (2, 2, 16000, 1.0), (2, 1, 48000, 3.0), (1, 1, 15400, 1.0)])
def test_mel_frontend_vs_oracle(n, c, a, audio_len):
    from types import SimpleNamespace
    from cavp_amd.audio_frontend import MelFrontEnd
    wave = _wave(n, c, a, seed=11)
    fe = MelFrontEnd(SimpleNamespace(audio_len=audio_len, spec_min=-100, spec_max=100))
    got = fe(wave.to("cuda:0")).cpu()
    ref = mel_oracle.preprocess_audio(wave, n_frames=fe.n_frames)
    assert got.shape == ref.shape == (n, c, fe.n_frames, 64)
    err = float((got - ref).abs().max())
    assert err <= 1e-4, err      # float32 direct DFT vs torch.stft: dB error 1e-2 x this on the +-1 scale


@pytest.mark.gpu
def test_mel_frontend_silence_and_errors():
    from cavp_amd import _lib
    from cavp_amd.audio_frontend import MelFrontEnd
    fe = MelFrontEnd()
    out = fe(torch.zeros(1, 1, 16000, device="cuda:0"))
    # floor: 20 log10(1e-5) = -100 dB -> -1
    assert torch.allclose(out, torch.full_like(out, -1.0))
    with pytest.raises(_lib.CavpError):
        fe(torch.zeros(1, 16000, device="cuda:0"))
    with pytest.raises(_lib.CavpError):
        fe(torch.zeros(1, 1, 8000, device="cuda:0"))   # fewer than 96 frames
