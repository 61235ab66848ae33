"""Waveform -> normalised log-mel tensor on the GPU: the MI355X counterpart of the trainers' `preprocess_audio`
(reference trainer/trainer_cavp_vpo_mono.py:43-52,59-69: torchaudio MelSpectrogram + sourcesep.db_from_amp +
sourcesep.normalize_spec), SURVEY.md §8(f) row f3.  Removes the torchaudio dependency on ROCm and lets the
end-to-end step start from 16 kHz waveforms.  The arithmetic runs in `cavp_mel_frontend` (csrc/mel_frontend.hip);
this module only builds the two constant tables (Hann window padded into the FFT frame, HTK mel filterbank) with
the same float32 operations torchaudio uses."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .ops import _need_gpu, _ptr, _stream


def hann_window_padded(win_length: int, n_fft: int) -> torch.Tensor:
    """torch.stft centres a window shorter than n_fft inside the frame (zero padding on both sides)."""
    w = torch.hann_window(win_length, periodic=True, dtype=torch.float32)
    left = (n_fft - win_length) // 2
    out = torch.zeros(n_fft, dtype=torch.float32)
    out[left:left + win_length] = w
    return out


def mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """HTK-scale triangular filterbank, no area normalisation: [n_freqs, n_mels] float32
    (the published torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk"))."""
    def hz_to_mel(f):
        return 2595.0 * math.log10(1.0 + f / 700.0)
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0).contiguous()


class MelFrontEnd:
    """`MelFrontEnd(args)(waveform[N, C, A]) -> [N, C, T, 64]`, T = 96 for 1-second clips (300 for 3 s), exactly the
    tensor `trainer.preprocess_audio` hands to `CAVP.forward`.  `args` needs .audio_len, .spec_min, .spec_max."""

    def __init__(self, args=None, device="cuda:0", sample_rate: int = 16000, n_fft: int = 512, win_length: int = 400,
                 hop_length: int = 160, n_mels: int = 64, f_min: float = 125.0, f_max: float = 3800.0):
        audio_len = float(getattr(args, "audio_len", 1.0)) if args is not None else 1.0
        self.n_frames = 96 if audio_len == 1.0 else 300          # trainer_cavp_vpo_mono.py:61
        self.spec_min = float(getattr(args, "spec_min", -100)) if args is not None else -100.0
        self.spec_max = float(getattr(args, "spec_max", 100)) if args is not None else 100.0
        self.n_fft, self.hop, self.n_mels = n_fft, hop_length, n_mels
        self.window = hann_window_padded(win_length, n_fft).to(device)
        self.fb = mel_filterbank(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate).to(device)

    def __call__(self, audio: torch.Tensor) -> torch.Tensor:
        _need_gpu(audio)
        if audio.dim() != 3 or audio.dtype != torch.float32:
            raise _lib.CavpError("MelFrontEnd: float32 waveform [N, C, A] required")
        n, c, a = audio.shape
        x = audio.contiguous()
        out = torch.empty((n, c, self.n_frames, self.n_mels), dtype=torch.float32, device=audio.device)
        st = _lib.load().cavp_mel_frontend(_ptr(x), n * c, a, _ptr(self.window), _ptr(self.fb), _ptr(out), self.n_fft,
                                           self.hop, self.n_frames, self.n_mels, C.c_float(1e-5),
                                           C.c_float(self.spec_min), C.c_float(self.spec_max), C.c_void_p(_stream()))
        _lib.check(st, "cavp_mel_frontend")
        return out
