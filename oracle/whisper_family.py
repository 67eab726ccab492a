"""TEST INFRASTRUCTURE ONLY -- CPU restatements of the two remaining Whisper-family backbones behind the `Mapperatorinator`
wrapper: 'Tiger14n/ropewhisper-*' (the V30 / V31 releases, configs/model/whisper_small_v2.yaml) and
stock 'openai/whisper-*' (the V28 / V29 releases, configs/model/whisper_{base,small}.yaml).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module; it is the checker, never
the product.

RoPEWhisperOracle follows osuT5/osuT5/model/custom_transformers/modeling_ropewhisper.py:
  rotary tables        LlamaRotaryEmbedding.forward :312-332 (fp32 inv_freq = 10000^(-2i/64), fp32 angles, cos | sin cast to the
                       activation dtype).  rope_type "dynamic" with factor 1.0 (configs/model/default.yaml:15-17): the NTK
                       re-scaling only triggers for positions beyond max_position_embeddings (:299-305), which the encoder
                       (max_source_positions) and the decoder (max_target_positions = the StaticCache length) never reach
  rotate-half RoPE     rotate_half / apply_rotary_pos_emb :228-259, on q and k of the self-attentions only (:886 passes
                       position_embeddings=None to the cross-attention)
  attention            RoPEWhisperAttention :346-515: separate q_proj (bias) / k_proj (NO bias) / v_proj (bias) / out_proj (bias);
                       the query is scaled by head_dim^-0.5 BEFORE the rotation (:408), scores = q k^T + mask, softmax, @ v
  encoder layer        RoPEWhisperEncoderLayer :757-826 (pre-norm nn.RMSNorm, fc1 -> gelu(erf) -> fc2)
  encoder              RoPEWhisperEncoder :1116-1278 (conv1 k3 p1 -> gelu -> conv2 k3 s2 p1 -> gelu, positions 0 .. L-1, final RMSNorm)
  decoder layer        RoPEWhisperDecoderLayer :829-946
  decoder              RoPEWhisperDecoder :1280-1560; position_ids = `decoder_position_ids` of the fork's own
                       prepare_inputs_for_generation (:2015-2018: (decoder_attention_mask.cumsum(-1) - 1).clamp(min=0)), i.e. a
                       left-padded row counts its positions from its first real token
  wrapper              osuT5/osuT5/model/modeling_mapperatorinator.py:174-209: with project_encoder_input = false the
                       conditioning vectors are repeated over the frames and concatenated to the log-mel CHANNELS (:197-202),
                       conv1 has n_mels + cond_size input channels (configuration_mapperatorinator.py:104)

HFWhisperOracle follows the third-party transformers models/whisper/modeling_whisper.py (pin 4.57.3; the installed 5.15 was
read: WhisperAttention "query_states = self.q_proj(hidden_states) * self.scaling", k_proj without bias, nn.LayerNorm pre-norm
blocks, `hidden_states = inputs_embeds + self.embed_positions(all_positions)` in the encoder, `inputs_embeds + positions` with
the learned WhisperPositionalEmbedding in the decoder) behind the wrapper's encoder_embedder (project_encoder_input = true,
modeling_mapperatorinator.py:204-205,211) and decoder_embedder (:217-219).
  VERSION-SKEW HAZARD: transformers 4.57's Whisper `prepare_inputs_for_generation` derives decoder_position_ids from the
  decoder attention mask (the code the RoPEWhisper fork copied, modeling_ropewhisper.py:2015-2018); the installed 5.x hands
  the decoder no position ids, so it uses cache positions (`torch.arange(...) + past_key_values_length`).  The two differ for
  left-padded rows only.  `positions="cache"` (default: what the imported reference does HERE and what the goldens pin) or
  "mask" (the 4.57 behaviour: pinned by goldens `ids_mask_positions`, which the imported reference produces when it is
  handed those position ids explicitly -- ref_harness.reference_generate_whisper_family(positions_from_mask=True)).

PINNING: tests/golden/rw_*.npz and hfw_*.npz come from the imported reference (oracle/make_golden.py: encoder states, greedy
ids, per-step scores through its own `model_generate`); tests/test_oracle_pinned.py checks these restatements against them.
"""
from __future__ import annotations

import torch

from . import mel as omel
from .varwhisper import VarWhisperOracle, apply_rope, gelu_erf, rms_norm, rope_tables


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


class RoPEWhisperOracle(VarWhisperOracle):
    """Stateless math over a reference-named state_dict of `Mapperatorinator` with a RoPEWhisper backbone."""

    E, D = "transformer.model.encoder.", "transformer.model.decoder."

    def __init__(self, sd: dict, d_model, n_heads, n_enc, n_dec, rounding=None, n_mels=80):
        super().__init__(sd, d_model, n_heads, n_enc, n_dec, rounding=rounding)
        self.n_mels = n_mels

    # ---- pieces -------------------------------------------------------------------------------------------------
    def _norm(self, x, name):
        return rms_norm(x, self.sd[name + ".weight"], self.eps)

    def _heads(self, x):
        B, T, _ = x.shape
        return x.view(B, T, self.H, 64).transpose(1, 2)

    def _attn_plain(self, q, k, v, mask=None):
        """scores = q k^T (the query arrives scaled, :408) + mask, softmax, @ v"""
        r = self.r
        scores = torch.matmul(r(q), r(k).transpose(-1, -2))
        if mask is not None:
            scores = scores.masked_fill(~mask, torch.finfo(torch.float32).min)
        out = torch.matmul(r(torch.softmax(scores, dim=-1)), r(v))
        B, H, T, _ = out.shape
        return out.transpose(1, 2).reshape(B, T, H * 64)

    def _q(self, n, pre):
        return self._heads(self.r(self._lin(n, pre + "q_proj") * 0.125))

    def _rope_rows(self, x, positions):
        """x (B, H, T, 64), positions (B, T) long: per-row rotary tables"""
        B, _, T, _ = x.shape
        cos, sin = rope_tables(positions.reshape(-1).tolist(), self.theta, 64, self.act_dtype)
        cos, sin = cos.view(B, 1, T, 64), sin.view(B, 1, T, 64)
        x1, x2 = x[..., :32], x[..., 32:]
        return x * cos + torch.cat((-x2, x1), dim=-1) * sin

    # ---- encoder ------------------------------------------------------------------------------------------------
    def log_mel(self, audio, n_mels=None):
        return omel.mel_spectrogram_torchaudio(audio, n_mels=n_mels or self.n_mels, log_scale=True)

    def with_cond(self, mel, cond):
        """(B, L, n_mels) | (B, cond_size) repeated over the frames (modeling_mapperatorinator.py:201-202)"""
        if cond is None:
            return mel
        return torch.cat([mel, self.r(cond)[:, None, :].expand(-1, mel.shape[1], -1)], -1)

    def encoder(self, h):
        r = self.r
        B, L, _ = h.shape
        cos, sin = rope_tables(list(range(L)), self.theta, 64, self.act_dtype)
        for l in range(self.ne):
            b = f"{self.E}layers.{l}."
            n = r(self._norm(h, b + "self_attn_layer_norm"))
            q = r(apply_rope(self._q(n, b + "self_attn."), cos, sin))
            k = r(apply_rope(self._heads(r(self._lin(n, b + "self_attn.k_proj"))), cos, sin))
            v = self._heads(r(self._lin(n, b + "self_attn.v_proj")))
            h = h + self._lin(r(self._attn_plain(q, k, v)), b + "self_attn.out_proj")
            n = r(self._norm(h, b + "final_layer_norm"))
            h = h + self._mlp(n, b)
        return self._norm(h, self.E + "layer_norm")

    def encode_audio(self, audio, n_mels=None, cond=None):
        return self.encoder(self.frontend(self.with_cond(self.log_mel(audio, n_mels), cond)))

    # ---- decoder ------------------------------------------------------------------------------------------------
    def cross_kv(self, enc):
        r = self.r
        e = r(enc)
        out = []
        for l in range(self.nd):
            b = f"{self.D}layers.{l}.encoder_attn."
            out.append((self._heads(r(self._lin(e, b + "k_proj"))), self._heads(r(self._lin(e, b + "v_proj")))))
        return out

    def _positions(self, pos, key_mask):
        """decoder_position_ids of column `pos` per row: (cumsum(mask) - 1).clamp(min=0) (:2015-2018)"""
        return (key_mask[:, :pos + 1].long().sum(-1) - 1).clamp(min=0)[:, None]

    def _embed(self, tok, pos, key_mask):
        return self.sd["decoder_embedder.weight"][tok][:, None, :]

    def decoder_step(self, tok, pos, cache, ckv, key_mask):
        r = self.r
        h = self._embed(tok, pos, key_mask)
        m = key_mask[:, None, None, :pos + 1]
        pids = self._positions(pos, key_mask)
        for l in range(self.nd):
            b = f"{self.D}layers.{l}."
            n = r(self._norm(h, b + "self_attn_layer_norm"))
            q = r(self._rotate(self._q(n, b + "self_attn."), pids))
            k = r(self._rotate(self._heads(r(self._lin(n, b + "self_attn.k_proj"))), pids))
            v = self._heads(r(self._lin(n, b + "self_attn.v_proj")))
            K, V = cache[l]
            K[:, :, pos] = k[:, :, 0]
            V[:, :, pos] = v[:, :, 0]
            h = h + self._lin(r(self._attn_plain(q, K[:, :, :pos + 1], V[:, :, :pos + 1], m)), b + "self_attn.out_proj")
            n = r(self._norm(h, b + "encoder_attn_layer_norm"))
            h = h + self._lin(r(self._attn_plain(self._q(n, b + "encoder_attn."), ckv[l][0], ckv[l][1])), b + "encoder_attn.out_proj")
            n = r(self._norm(h, b + "final_layer_norm"))
            h = h + self._mlp(n, b)
        n = r(self._norm(h, self.D + "layer_norm"))
        return (n @ self.sd["transformer.proj_out.weight"].t())[:, 0, :]

    def _rotate(self, x, pids):
        return self._rope_rows(x, pids)


class HFWhisperOracle(RoPEWhisperOracle):
    """Stock HF Whisper behind the wrapper (V28 / V29): affine LayerNorm, absolute positions, no rotary embedding, the
    wrapper's encoder_embedder in front of conv1."""

    def __init__(self, sd: dict, d_model, n_heads, n_enc, n_dec, rounding=None, n_mels=388, positions="cache"):
        super().__init__(sd, d_model, n_heads, n_enc, n_dec, rounding=rounding, n_mels=n_mels)
        assert positions in ("cache", "mask")
        self.positions = positions

    def _norm(self, x, name):
        return layer_norm(x, self.sd[name + ".weight"], self.sd[name + ".bias"], 1e-5)

    def log_mel(self, audio, n_mels=None):
        # configs/model/whisper_{base,small}.yaml inherit the nnAudio front-end of default.yaml:29-37 (power mel, no log)
        return omel.mel_spectrogram(audio, n_mels=n_mels or self.n_mels)

    def frontend(self, mel):
        """encoder_embedder (modeling_mapperatorinator.py:204-205) -> conv1 -> gelu -> conv2 -> gelu -> + embed_positions"""
        sd, r = self.sd, self.r
        x = r(r(mel) @ sd["encoder_embedder.weight"].t() + sd["encoder_embedder.bias"])
        h = super().frontend(x)
        return h + sd[self.E + "embed_positions.weight"][None, :h.shape[1]]

    def encoder(self, h):
        r = self.r
        for l in range(self.ne):
            b = f"{self.E}layers.{l}."
            n = r(self._norm(h, b + "self_attn_layer_norm"))
            q = self._q(n, b + "self_attn.")
            k = self._heads(r(self._lin(n, b + "self_attn.k_proj")))
            v = self._heads(r(self._lin(n, b + "self_attn.v_proj")))
            h = h + self._lin(r(self._attn_plain(q, k, v)), b + "self_attn.out_proj")
            n = r(self._norm(h, b + "final_layer_norm"))
            h = h + self._mlp(n, b)
        return self._norm(h, self.E + "layer_norm")

    def _embed(self, tok, pos, key_mask):
        p = self._positions(pos, key_mask)[:, 0] if self.positions == "mask" else torch.full((tok.shape[0],), pos, dtype=torch.long)
        return (self.sd["decoder_embedder.weight"][tok] + self.sd[self.D + "embed_positions.weight"][p])[:, None, :]

    def _rotate(self, x, pids):
        return x
