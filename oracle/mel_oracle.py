"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md §3): CPU restatement of the trainers' log-mel front-end,
`preprocess_audio` (reference trainer/trainer_cavp_vpo_mono.py:43-52,59-69) + utils/sourcesep.py:23-47.

The reference builds it from torchaudio.transforms.MelSpectrogram (torchaudio is pinned in the reference's
requirements.txt but NOT installed in this image, and its source is not under /root/reference).  Restated from its
published definition:
  * Spectrogram: torch.stft(n_fft=512, hop_length=160, win_length=400, window=hann_window(400) [periodic],
    center=True, pad_mode="reflect", normalized=False, onesided=True), |.|^2  -> here the real torch.stft is called,
    so the STFT half of this oracle IS the library the reference runs;
  * MelScale(n_mels=64, f_min=125, f_max=3800, n_stft=257, norm=None, mel_scale="htk") with
    torchaudio.functional.melscale_fbanks: triangular filters between mel-equidistant points, evaluated in float32.
PARITY UNPINNED for the filterbank half: no golden vector of the reference exists for it (the reference has no test
and torchaudio cannot be imported here); tests pin the STFT half against torch.stft and the filterbank against its
closed form.
"""
import math

import torch


def hz_to_mel_htk(f: float) -> float:
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") -> [n_freqs, n_mels] float32."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min, m_max = hz_to_mel_htk(f_min), hz_to_mel_htk(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def preprocess_audio(wave: torch.Tensor, n_frames: int = 96, sample_rate: int = 16000, n_fft: int = 512,
                     win_length: int = 400, hop_length: int = 160, n_mels: int = 64, f_min: float = 125.0,
                     f_max: float = 3800.0, spec_min: float = -100.0, spec_max: float = 100.0) -> torch.Tensor:
    """wave [N, C, A] float32 -> [N, C, n_frames, n_mels] (trainer_cavp_vpo_mono.py:59-69)."""
    n, c, a = wave.shape
    x = wave.reshape(n * c, a).float()
    spec = torch.stft(x, n_fft, hop_length=hop_length, win_length=win_length, window=torch.hann_window(win_length),
                      center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.abs() ** 2                                            # [NC, 257, frames]
    fb = melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate)
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)  # MelScale: [NC, n_mels, frames]
    mel = mel[:, :, :n_frames].transpose(-1, -2)                       # [NC, frames, n_mels]
    db = 20.0 * (torch.log(torch.max(torch.tensor(1e-5), mel.float())) / torch.log(torch.tensor(10.0)))
    out = 2.0 * (db - spec_min) / float(spec_max - spec_min) - 1.0
    return out.view(n, c, n_frames, n_mels)
