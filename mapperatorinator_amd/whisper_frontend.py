"""Host side of K2: the Whisper-style conv + position audio front-end.

`WhisperFrontendHIP` replaces the first lines of `WhisperEncoder.forward` / `VarWhisperEncoder.forward`
(HF modeling_whisper.py "inputs_embeds = gelu(conv1(input_features)) ..."; reference fork
osuT5/osuT5/model/custom_transformers/modeling_varwhisper.py:779-780,813-816): two k=3 convolutions with exact
GELU, the second with stride 2, then (HF Whisper only) the fixed sinusoid `embed_positions`.
The transformer layers of the Whisper-family backbones (RoPE / sliding window / nGPT variants) are NOT on the HIP
path; this op is the front-end kernel the north star names.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def _round_up(a, b):
    return (a + b - 1) // b * b


class WhisperFrontendHIP:
    def __init__(self, conv1_weight, conv1_bias, conv2_weight, conv2_bias, embed_positions: Optional[torch.Tensor] = None,
                 dtype: torch.dtype = torch.bfloat16, device="cuda"):
        """conv*_weight: [d, C_in, 3] as in nn.Conv1d; embed_positions: [max_source_positions, d] or None."""
        if not torch.cuda.is_available():
            raise RuntimeError("WhisperFrontendHIP needs a ROCm GPU; there is no CPU fallback")
        assert dtype in (torch.float32, torch.bfloat16)
        self.lib = _lib.load()
        self.device, self.dtype = torch.device(device), dtype
        self.d, self.c_in = conv1_weight.shape[0], conv1_weight.shape[1]
        assert conv1_weight.shape[2] == 3 and tuple(conv2_weight.shape) == (self.d, self.d, 3)

        def pack(w):  # [d, C, 3] -> [d, 3*C] tap-major, K padded to a multiple of 32, storage dtype
            d, c, _ = w.shape
            m = w.detach().float().permute(0, 2, 1).reshape(d, 3 * c)
            m = torch.nn.functional.pad(m, (0, _round_up(3 * c, 32) - 3 * c))
            return m.to(dtype).contiguous().to(self.device)

        def vec(b):
            return b.detach().to(dtype).float().contiguous().to(self.device)

        self.w1, self.w2 = pack(conv1_weight), pack(conv2_weight)
        self.b1, self.b2 = vec(conv1_bias), vec(conv2_bias)
        self.pos = vec(embed_positions) if embed_positions is not None else None
        self._ws = None

    def out_len(self, l_in: int) -> int:
        return (l_in - 1) // 2 + 1

    @torch.no_grad()
    def forward_time_major(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, L, C_in) -> (B, (L-1)//2+1, d) in the storage dtype."""
        B, L, C = x.shape
        if C != self.c_in:
            raise ValueError(f"expected {self.c_in} input channels, got {C}")
        lo = self.out_len(L)
        if self.pos is not None and self.pos.shape[0] != lo:
            raise ValueError(f"Whisper expects {self.pos.shape[0] * 2} input frames (embed_positions has "
                             f"{self.pos.shape[0]} rows), got {L}")
        x = x.to(self.device, self.dtype).contiguous()
        dt = _lib.MH_BF16 if self.dtype == torch.bfloat16 else _lib.MH_F32
        need = self.lib.mh_whisper_frontend_workspace_bytes(B, L, C, self.d, dt)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        out = torch.empty((B, lo, self.d), dtype=self.dtype, device=self.device)
        rc = self.lib.mh_whisper_frontend(x.data_ptr(), B, L, C, self.w1.data_ptr(), self.b1.data_ptr(),
                                          self.w2.data_ptr(), self.b2.data_ptr(), _lib.ptr(self.pos), self.d,
                                          out.data_ptr(), self._ws.data_ptr(), self._ws.numel(), dt,
                                          torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "mh_whisper_frontend")
        return out

    def forward(self, input_features: torch.Tensor) -> torch.Tensor:
        """HF layout: input_features (B, C_in, L) -> hidden states (B, L_out, d)."""
        return self.forward_time_major(input_features.transpose(1, 2))

    __call__ = forward
