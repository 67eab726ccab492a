"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the Whisper-family backbone of the released V30-V32 checkpoints
(`OliBomby/varwhisper-*`, configs/model/varwhisper_{small,base}_v3.yaml) behind the same wrapper, SURVEY.md 8f rank 2.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module; it is the checker,
never the product.  It follows osuT5/osuT5/model/custom_transformers/modeling_varwhisper.py:

  rotary tables        VarWhisperRotaryEmbedding.forward :212-226 (inv_freq theta^(-2i/64), fp32 angles, cos | sin cast
                       to the activation dtype); the default initialiser it looks up is HF's published
                       `_compute_default_rope_parameters` (third-party, transformers 4.57 pin)
  rotate-half RoPE     rotate_half / apply_rotary_pos_emb :229-258 (on q and k of the self-attention only)
  attention            VarWhisperAttention :381-568: fused Wqkv (self) / Wq + Wkv (cross) with optional bias, Wo;
                       softmax(q k^T / sqrt(64) + mask) v as eager_attention_forward :261-291 / sdpa :347-372 compute it.
                       Local layers (layer_idx % global_attn_every_n_layers != 0) attend keys within
                       local_attention // 2 on either side -- on the reference's flash-attention path ONLY (:330
                       `window_size=local_attention`); its eager / sdpa paths receive sliding_window_mask=None (:466)
                       and attend everything.  `local_window=True` restates the flash path; the released configs keep
                       global_attn_every_n_layers = 1 (configs/model/default.yaml:24), i.e. no local layers.
  encoder layer        VarWhisperEncoderLayer :571-630 (pre-norm nn.RMSNorm, fc1 -> gelu(erf) -> fc2 with biases)
  encoder              VarWhisperEncoder :751-852 (conv1 k3 p1 -> gelu -> conv2 k3 s2 p1 -> gelu, no position table,
                       final RMSNorm)
  decoder layer        VarWhisperDecoderLayer :633-741
  decoder              VarWhisperDecoder :938-1150 (inputs_embeds from the WRAPPER's decoder_embedder,
                       modeling_mapperatorinator.py:205-206; position_ids = cache positions, left padding included)
  head                 VarWhisperForConditionalGeneration.proj_out :1354 (no bias, untied)
  nn.RMSNorm           eps=None -> torch.finfo(x.dtype).eps: 1.1920929e-07 in fp32, 0.0078125 in bf16 (third-party torch)
  generation loop, processors, guidance: inherited from oracle/t5.py (HF `_sample` under osuT5/osuT5/inference/server.py)

PINNING: tests/golden/vw_*.npz come from the imported reference (oracle/make_golden.py: its encoder states, greedy ids and
per-step scores); tests/test_oracle_pinned.py checks this restatement against them.  The torchaudio log-mel front-end in
front of it is oracle/mel.py::mel_spectrogram_torchaudio (parity unpinned for that one stage: torchaudio is not installed).
"""
from __future__ import annotations

import math

import torch

from . import mel as omel
from .t5 import Rounding, T5Oracle


def rms_norm(x, w, eps):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def rope_tables(positions, theta=10000.0, dim=64, dtype=torch.float32):
    """cos, sin (len(positions), dim) as VarWhisperRotaryEmbedding.forward builds them: fp32 angles, halves duplicated,
    cast to the activation dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float32) / dim))
    freqs = torch.as_tensor(positions, dtype=torch.float32)[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype).to(torch.float32), emb.sin().to(dtype).to(torch.float32)


def apply_rope(x, cos, sin):
    """x (B, H, T, 64); cos / sin (T, 64): x cos + rotate_half(x) sin"""
    x1, x2 = x[..., :32], x[..., 32:]
    return x * cos + torch.cat((-x2, x1), dim=-1) * sin


class VarWhisperOracle(T5Oracle):
    """Stateless math over a reference-named state_dict of `Mapperatorinator` with a VarWhisper backbone."""

    def __init__(self, sd: dict, d_model, n_heads, n_enc, n_dec, theta=10000.0, rounding=None, every_n=1, local_attention=128,
                 local_theta=10000.0, local_window=False):
        self.r = Rounding(rounding)
        self.sd = {k: self.r.w(v.detach().to(torch.float32)) for k, v in sd.items() if v.dtype.is_floating_point}
        assert d_model == n_heads * 64, "whisper-family heads are 64 wide"
        self.d, self.H, self.ne, self.nd = d_model, n_heads, n_enc, n_dec
        self.act_dtype = torch.bfloat16 if rounding == "bf16" else torch.float32
        self.eps = float(torch.finfo(self.act_dtype).eps)
        self.theta, self.every_n, self.local, self.local_theta, self.local_window = theta, every_n, local_attention, local_theta, local_window

    # ---- pieces ----------------------------------------------------------------------------
    def _lin(self, x, name):
        y = x @ self.sd[name + ".weight"].t()
        return y + self.sd[name + ".bias"] if name + ".bias" in self.sd else y

    def _is_local(self, layer):
        return layer % self.every_n != 0

    def _rope(self, layer, positions):
        return rope_tables(positions, self.local_theta if self._is_local(layer) else self.theta, 64, self.act_dtype)

    def _attn_scaled(self, q, k, v, mask=None):
        r = self.r
        scores = torch.matmul(r(q), r(k).transpose(-1, -2)) * 0.125
        if mask is not None:
            scores = scores.masked_fill(~mask, torch.finfo(torch.float32).min)
        p = torch.softmax(scores, dim=-1)
        out = torch.matmul(r(p), r(v))
        B, H, T, _ = out.shape
        return out.transpose(1, 2).reshape(B, T, H * 64)

    def _window(self, layer, q_pos, k_pos):
        """(len(q_pos), len(k_pos)) bool, True = attend, for a local layer under the flash-attention window semantics
        (keys within local_attention // 2 on either side); None for global layers or when the eager path is restated"""
        if not (self.local_window and self._is_local(layer)):
            return None
        qp, kp = torch.as_tensor(q_pos)[:, None], torch.as_tensor(k_pos)[None, :]
        return (kp >= qp - self.local // 2) & (kp <= qp + self.local // 2)

    def _mlp(self, n, pre):
        r = self.r
        return self._lin(r(gelu_erf(self._lin(n, pre + "fc1"))), pre + "fc2")

    # ---- encoder ---------------------------------------------------------------------------
    def frontend(self, mel):
        """mel (B, L, n_mels) log-mel frames -> (B, ceil(L / 2), d): the two convolutions of VarWhisperEncoder.forward"""
        sd, r = self.sd, self.r
        p = "transformer.model.encoder."
        x = torch.nn.functional.conv1d(r(mel.transpose(1, 2)), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
        x = torch.nn.functional.conv1d(r(gelu_erf(x)), sd[p + "conv2.weight"], sd[p + "conv2.bias"], stride=2, padding=1)
        return r(gelu_erf(x)).transpose(1, 2)     # (the residual stream starts from the storage-typed conv output)

    def encoder(self, h):
        sd, r = self.sd, self.r
        B, L, _ = h.shape
        pos = list(range(L))
        for l in range(self.ne):
            b = f"transformer.model.encoder.layers.{l}."
            n = r(rms_norm(h, sd[b + "self_attn_layer_norm.weight"], self.eps))
            qkv = r(self._lin(n, b + "self_attn.Wqkv")).view(B, L, 3, self.H, 64)
            q, k, v = qkv.transpose(1, 3).unbind(dim=2)                       # (B, H, L, 64) each
            cos, sin = self._rope(l, pos)
            q, k = r(apply_rope(q, cos, sin)), r(apply_rope(k, cos, sin))
            w = self._window(l, pos, pos)
            h = h + self._lin(r(self._attn_scaled(q, k, v, None if w is None else w[None, None])), b + "self_attn.Wo")
            n = r(rms_norm(h, sd[b + "final_layer_norm.weight"], self.eps))
            h = h + self._mlp(n, b)
        return rms_norm(h, sd["transformer.model.encoder.layer_norm.weight"], self.eps)

    def log_mel(self, audio, n_mels=128):
        return omel.mel_spectrogram_torchaudio(audio, n_mels=n_mels, log_scale=True)

    def encode_audio(self, audio, n_mels=128, cond=None):
        assert cond is None, "the Whisper-family configs carry no conditioning embedders"
        return self.encoder(self.frontend(self.log_mel(audio, n_mels)))

    # ---- decoder ---------------------------------------------------------------------------
    def cross_kv(self, enc):
        r = self.r
        e = r(enc)
        B, L, _ = e.shape
        out = []
        for l in range(self.nd):
            kv = r(self._lin(e, f"transformer.model.decoder.layers.{l}.cross_attn.Wkv")).view(B, L, 2, self.H, 64)
            k, v = kv.transpose(1, 3).unbind(dim=2)
            out.append((k, v))
        return out

    def decoder_step(self, tok, pos, cache, ckv, key_mask):
        """tok (B,) ids fed at position `pos`; cache: list of (K, V) (B, H, Tmax, 64), K stored ROTATED (the reference's
        cache.update receives the rotated keys, :549-552); key_mask (B, Tmax) bool.  Returns fp32 logits (B, V)."""
        sd, r = self.sd, self.r
        B = tok.shape[0]
        h = sd["decoder_embedder.weight"][tok][:, None, :]
        m = key_mask[:, None, None, :pos + 1]
        for l in range(self.nd):
            b = f"transformer.model.decoder.layers.{l}."
            n = r(rms_norm(h, sd[b + "self_attn_layer_norm.weight"], self.eps))
            qkv = r(self._lin(n, b + "self_attn.Wqkv")).view(B, 1, 3, self.H, 64)
            q, k, v = qkv.transpose(1, 3).unbind(dim=2)
            cos, sin = self._rope(l, [pos])
            q, k = r(apply_rope(q, cos, sin)), r(apply_rope(k, cos, sin))
            K, V = cache[l]
            K[:, :, pos] = k[:, :, 0]
            V[:, :, pos] = v[:, :, 0]
            w = self._window(l, [pos], list(range(pos + 1)))
            mm = m if w is None else m & w[None, None]
            h = h + self._lin(r(self._attn_scaled(q, K[:, :, :pos + 1], V[:, :, :pos + 1], mm)), b + "self_attn.Wo")
            n = r(rms_norm(h, sd[b + "cross_attn_layer_norm.weight"], self.eps))
            q = r(self._lin(n, b + "cross_attn.Wq")).view(B, 1, self.H, 64).transpose(1, 2)
            h = h + self._lin(r(self._attn_scaled(q, ckv[l][0], ckv[l][1])), b + "cross_attn.Wo")
            n = r(rms_norm(h, sd[b + "final_layer_norm.weight"], self.eps))
            h = h + self._mlp(n, b)
        n = r(rms_norm(h, sd["transformer.model.decoder.layer_norm.weight"], self.eps))
        return (n @ sd["transformer.proj_out.weight"].t())[:, 0, :]

    def decoder_forward(self, ids, ckv, key_mask=None):
        raise NotImplementedError("teacher-forced batch form: use generate(..., forced=ids, return_logits=True)")
