"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the mel frontend.

Reference call site: osuT5/osuT5/model/spectrogram.py:50-61 (ctor) and :63-83 (forward) which
delegates to `nnAudio.features.MelSpectrogram` (third-party, pinned nnAudio==0.3.4 in the
reference's requirements.txt:3, NOT vendored and NOT installable here).  This file restates
nnAudio 0.3.4's published algorithm:

  * STFT as two conv1d's with (sin, cos) x periodic-hann fp32 kernels of length n_fft, stride hop,
    `center=True` zero padding of n_fft//2 (pad_mode='constant'), magnitude = sqrt(re^2+im^2),
    then `** power` (power=2.0);
  * librosa-style Slaney mel filterbank, area normalised (norm=1), float32, applied as a matmul.

PINNED TO AN INDEPENDENT THIRD-PARTY IMPLEMENTATION (round 5): nnAudio itself is absent from the image (and the
reference ships no mel test vectors), but `transformers.audio_utils` (installed; numpy rfft + its own Slaney / HTK
filterbanks, written independently of nnAudio, librosa and torchaudio) implements the same published definitions.
tests/test_oracle_pinned.py::test_mel_oracle_pinned_to_transformers_audio_utils holds this file to it: filterbank
within 5e-8 (measured 1.0e-8) with every triangle on the same bins and the same single all-zero filter, the full mel
of noise and tones within 3e-6 of the peak (measured 5.8e-7 .. 9.1e-7) for the nnAudio parameterisation and within
2e-6 / 1e-5 (log1p) for the torchaudio one.  The analytic checks (frame count, pure-tone bin energy (N/4)^2,
silence == 0, filter areas, float64 FFT cross-check) stay in test_mel_oracle_analytic.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_t = f >= min_log_hz
        mels = np.where(log_t, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)
    return mels


def mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    log_t = m >= min_log_mel
    freqs = np.where(log_t, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)
    return freqs


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """librosa.filters.mel(htk=False, norm=1) as copied into nnAudio 0.3.4 `get_mel`:
    float64 ramps written into a float32 matrix, then scaled in place by 2/(f[i+2]-f[i])."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    fftfreqs = np.linspace(0, sr / 2.0, n_bins, endpoint=True)
    mel_pts = np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis].astype(np.float32)
    return weights


def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True)."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(np.float32)


def fourier_kernels(n_fft: int):
    """nnAudio `create_fourier_kernels(freq_scale='no')`: fp32 sin/cos tables times the fp32 window."""
    n_bins = n_fft // 2 + 1
    s = np.arange(0, n_fft, 1.0)
    k = np.arange(n_bins, dtype=np.float64)[:, None]
    wsin = np.sin(2 * np.pi * k * s / n_fft).astype(np.float32)
    wcos = np.cos(2 * np.pi * k * s / n_fft).astype(np.float32)
    win = hann_periodic(n_fft)
    return wsin * win[None, :], wcos * win[None, :]


class NnAudioMelSpectrogram(nn.Module):
    """Drop-in for `nnAudio.features.MelSpectrogram(sr, n_fft, n_mels, hop_length, center=True,
    fmin, fmax, pad_mode)` as constructed at spectrogram.py:50-61.  (B, Ns) -> (B, n_mels, frames)."""

    def __init__(self, sr=16000, n_fft=1024, n_mels=388, hop_length=128, center=True,
                 fmin=0.0, fmax=8000.0, pad_mode="constant", power=2.0, **_):
        super().__init__()
        assert center and pad_mode == "constant", "only the configuration the hot path uses"
        self.n_fft, self.hop, self.power = n_fft, hop_length, power
        wsin, wcos = fourier_kernels(n_fft)
        self.register_buffer("wsin", torch.from_numpy(wsin)[:, None, :], persistent=False)
        self.register_buffer("wcos", torch.from_numpy(wcos)[:, None, :], persistent=False)
        self.register_buffer("mel_basis", torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax)),
                             persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 1:
            x = x[None]
        x = x.to(torch.float32)[:, None, :]
        x = F.pad(x, (self.n_fft // 2, self.n_fft // 2), mode="constant", value=0.0)
        im = F.conv1d(x, self.wsin, stride=self.hop)
        re = F.conv1d(x, self.wcos, stride=self.hop)
        mag = torch.sqrt(re.pow(2) + im.pow(2))
        spec = mag ** self.power
        return torch.matmul(self.mel_basis, spec)


def mel_spectrogram(audio: torch.Tensor, sr=16000, n_fft=1024, n_mels=388, hop=128,
                    fmin=0.0, fmax=8000.0, log_scale=False) -> torch.Tensor:
    """`MelSpectrogram.forward` of the reference wrapper (spectrogram.py:63-83):
    (B, Ns) fp32 -> (B, Ns//hop + 1, n_mels) fp32."""
    m = NnAudioMelSpectrogram(sr, n_fft, n_mels, hop, fmin=fmin, fmax=fmax)
    spec = m(audio)
    if log_scale:
        spec = torch.log1p(spec)
    return spec.permute(0, 2, 1).contiguous()


def mel_spectrogram_f64(audio: np.ndarray, sr=16000, n_fft=1024, n_mels=388, hop=128,
                        fmin=0.0, fmax=8000.0) -> np.ndarray:
    """Independent float64 rFFT formulation (accuracy yard-stick for both the fp32 restatement
    above and the HIP kernel)."""
    audio = np.asarray(audio, dtype=np.float64)
    if audio.ndim == 1:
        audio = audio[None]
    B, ns = audio.shape
    pad = n_fft // 2
    xp = np.pad(audio, ((0, 0), (pad, pad)))
    n_frames = ns // hop + 1
    idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)
    frames = xp[:, idx] * win
    spec = np.abs(np.fft.rfft(frames, axis=-1)) ** 2  # (B, L, bins)
    fb = mel_filterbank(sr, n_fft, n_mels, fmin, fmax).astype(np.float64)
    return spec @ fb.T


def mel_spectrogram_torchaudio(audio, n_fft=1024, hop=128, n_mels=128, sr=16000, f_min=20.0, f_max=8000.0, pad_mode="reflect",
                               log_scale=True):
    """The `torchaudio` branch of the reference wrapper (osuT5/osuT5/model/spectrogram.py:38-49, 79-83) restated with
    plain torch: torchaudio.transforms.MelSpectrogram = torch.stft (periodic hann, center=True, `pad_mode`, onesided,
    power 2) followed by `melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk")`.
    torchaudio is not installed here; `torch.stft` is the routine it calls, and the result is pinned to
    `transformers.audio_utils.spectrogram(pad_mode="reflect", mel_scale="htk", norm=None)` (see the module header).
    audio (B, Ns) -> (B, Ns // hop + 1, n_mels) float32."""
    import math
    x = audio.to(torch.float32)
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float32)
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=n_fft, window=win, center=True, pad_mode=pad_mode,
                      normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                                   # (B, bins, frames)
    n_bins = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sr // 2, n_bins, dtype=torch.float64)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    f_pts = 700.0 * (10.0 ** (torch.linspace(m_min, m_max, n_mels + 2, dtype=torch.float64) / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    fb = torch.clamp(torch.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0).to(torch.float32)
    mel = torch.matmul(power.transpose(1, 2), fb)                            # (B, frames, n_mels)
    return torch.log1p(mel) if log_scale else mel
