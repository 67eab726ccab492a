"""Host side of K1: `MelSpectrogram` with the reference's constructor/forward contract
(osuT5/osuT5/model/spectrogram.py:8-83), computed by `mh_mel` (csrc/mel.hip).

Two parameterisations: the nnAudio one of the T5 configs (center=True, pad_mode='constant', hann, power 2, Slaney
area-normalised filterbank; configs/model/default.yaml:29-37) and the torchaudio one of the Whisper-family configs
(reflect padding, HTK filterbank without normalisation, log1p; configs/model/whisper_base_v3.yaml:16-21).  The
host builds three small tables once (window, FFT twiddles, CSR filterbank) in float64 and hands
fp32 copies to the kernel.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    out = f / f_sp
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    big = f >= min_log_hz
    out = np.where(big, min_log_hz / f_sp + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, out)
    return out


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    out = f_sp * m
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    big = m >= min_log_mel
    return np.where(big, min_log_hz * np.exp(logstep * (m - min_log_mel)), out)


def slaney_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """[n_mels, n_fft//2+1] float32 triangular filters, area-normalised (librosa `mel(htk=False, norm=1)`
    semantics, which is what nnAudio 0.3.4 embeds)."""
    n_bins = n_fft // 2 + 1
    fft_f = np.linspace(0.0, sr / 2.0, n_bins)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    width = np.diff(pts)
    ramps = pts[:, None] - fft_f[None, :]
    fb = np.zeros((n_mels, n_bins), dtype=np.float32)
    for i in range(n_mels):
        fb[i] = np.maximum(0.0, np.minimum(-ramps[i] / width[i], ramps[i + 2] / width[i + 1]))
    fb *= (2.0 / (pts[2:] - pts[:-2]))[:, None].astype(np.float32)
    return fb


def htk_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """[n_mels, n_fft//2+1] float32 triangular filters on the HTK mel scale, no area normalisation: what
    `torchaudio.transforms.MelSpectrogram` builds by default (`melscale_fbanks(norm=None, mel_scale="htk")`: mel =
    2595 log10(1 + f / 700), n_mels + 2 points equally spaced in mel, rising / falling slopes over the FFT bin centres)."""
    n_bins = n_fft // 2 + 1
    fft_f = np.linspace(0.0, sr // 2, n_bins)
    m_lo, m_hi = 2595.0 * np.log10(1.0 + fmin / 700.0), 2595.0 * np.log10(1.0 + fmax / 700.0)
    pts = 700.0 * (10.0 ** (np.linspace(m_lo, m_hi, n_mels + 2) / 2595.0) - 1.0)
    width = np.diff(pts)
    slopes = pts[None, :] - fft_f[:, None]                      # (bins, n_mels + 2)
    down = -slopes[:, :-2] / width[:-1]
    up = slopes[:, 2:] / width[1:]
    return np.maximum(0.0, np.minimum(down, up)).T.astype(np.float32)


class MelSpectrogram(torch.nn.Module):
    """Same signature as the reference wrapper (`n_ftt` spelling included).  implementation "nnAudio" (Slaney filterbank,
    area-normalised) or "torchaudio" (HTK filterbank, no normalisation); pad_mode "constant" or "reflect"."""

    def __init__(self, implementation: str = "nnAudio", log_scale: bool = False, sample_rate: int = 16000,
                 n_ftt: int = 1024, n_mels: int = 388, hop_length: int = 128, f_min: int = 0,
                 f_max: int = 8000, pad_mode: str = "constant"):
        super().__init__()
        if implementation not in ("nnAudio", "torchaudio") or pad_mode not in ("constant", "reflect"):
            raise NotImplementedError(f"mel frontend: implementation {implementation!r} / pad_mode {pad_mode!r} is not built")
        if n_ftt != 1024:
            raise NotImplementedError("mh_mel is built for n_fft=1024")
        self.log_scale, self.sample_rate = bool(log_scale), sample_rate
        self.n_fft, self.n_mels, self.hop_length = n_ftt, n_mels, hop_length
        self.reflect = pad_mode == "reflect"
        fb = (slaney_filterbank if implementation == "nnAudio" else htk_filterbank)(sample_rate, n_ftt, n_mels, float(f_min), float(f_max))
        starts, lens, offs, ws = [], [], [], []
        off = 0
        for i in range(n_mels):
            nz = np.nonzero(fb[i])[0]
            if len(nz) == 0:
                s, ln = 0, 0
            else:
                s, ln = int(nz[0]), int(nz[-1] - nz[0] + 1)
            starts.append(s)
            lens.append(ln)
            offs.append(off)
            ws.append(fb[i, s:s + ln])
            off += ln
        n = np.arange(n_ftt, dtype=np.float64)
        window = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_ftt)).astype(np.float32)
        ang = 2.0 * np.pi * n / n_ftt
        tw = np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)
        self.register_buffer("window", torch.from_numpy(window), persistent=False)
        self.register_buffer("twiddle", torch.from_numpy(tw).contiguous(), persistent=False)
        self.register_buffer("fb_start", torch.tensor(starts, dtype=torch.int32), persistent=False)
        self.register_buffer("fb_len", torch.tensor(lens, dtype=torch.int32), persistent=False)
        self.register_buffer("fb_off", torch.tensor(offs, dtype=torch.int32), persistent=False)
        wcat = np.concatenate(ws) if off > 0 else np.zeros(1, np.float32)
        self.register_buffer("fb_w", torch.from_numpy(np.ascontiguousarray(wcat, dtype=np.float32)), persistent=False)

    def n_frames(self, n_samples: int) -> int:
        return n_samples // self.hop_length + 1

    def forward_padded(self, samples: torch.Tensor, ld_out: int, out_dtype: torch.dtype) -> torch.Tensor:
        """(B, Ns) fp32 on the GPU -> (B, frames, ld_out) with zero K-padding columns; dtype fp32 or bf16."""
        if not samples.is_cuda:
            raise RuntimeError("MelSpectrogram (HIP): input must live on the GPU; there is no CPU fallback")
        if self.window.device != samples.device:
            self.to(samples.device)
        x = samples.contiguous().to(torch.float32)
        B, ns = x.shape
        out = torch.empty((B, self.n_frames(ns), ld_out), dtype=out_dtype, device=x.device)
        lib = _lib.load()
        rc = lib.mh_mel(x.data_ptr(), B, ns, self.n_fft, self.hop_length, self.n_mels, self.window.data_ptr(),
                        self.twiddle.data_ptr(), self.fb_start.data_ptr(), self.fb_len.data_ptr(),
                        self.fb_off.data_ptr(), self.fb_w.data_ptr(), int(self.log_scale) | (2 if self.reflect else 0), out.data_ptr(), ld_out,
                        _lib.MH_BF16 if out_dtype == torch.bfloat16 else _lib.MH_F32,
                        torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc, "mh_mel")
        return out

    def forward(self, samples: torch.Tensor) -> torch.Tensor:
        """Reference contract: (B, Ns) -> (B, Ns // hop + 1, n_mels) float32."""
        return self.forward_padded(samples, self.n_mels, torch.float32)
