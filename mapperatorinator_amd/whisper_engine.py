"""Host side of the Whisper-family backbone: the released V30-V32 checkpoints run
'OliBomby/varwhisper-*' (osuT5/osuT5/model/custom_transformers/modeling_varwhisper.py; configs/model/
varwhisper_{small,base}_v3.yaml: torchaudio log-mel with 128 mels -> `input_features`, no encoder projection, the
wrapper's own decoder embedding, untied head) behind the same `Mapperatorinator` wrapper and the same `model_generate`.

`VarWhisperEngine` IS a `T5Engine` for every caller (mel -> encode -> cross-K/V -> KV-cached decode with the
reference's logits processors; `model_generate`, the window scheduler and the sharding code are unchanged): the
library's entry points take `MhT5Config.arch = 1` and run
  K1  log-mel (reflect padding, HTK filterbank, log1p: spectrogram.py:38-49),
  K2  conv1 k3 + GELU, conv2 k3 stride 2 + GELU (modeling_varwhisper.py:779-780,813-816),
  encoder / decoder layers: pre-norm nn.RMSNorm (eps = finfo(dtype).eps), fused Wqkv / Wq + Wkv / Wo with optional
  biases, rotate-half RoPE on q / k of the self-attentions (tables built HERE with VarWhisperRotaryEmbedding's formulas,
  :212-226), softmax(q k^T / 8), fc1 -> gelu(erf) -> fc2, final RMSNorm, `proj_out`.
Local layers (`global_attn_every_n_layers > 1`): keys within `local_attention // 2` on either side, the window the
reference applies on its flash-attention path (:330); its eager / sdpa paths ignore it (:466), the released configs keep
every layer global (configs/model/default.yaml:24).

Round 6 -- the other two released backbone families run through the same engine (`whisper_kind`, read off the parameter names):
  "rope"  'Tiger14n/ropewhisper-*' (V30 / V31; custom_transformers/modeling_ropewhisper.py): separate q_proj (bias) / k_proj (no
          bias) / v_proj (bias) / out_proj, packed HERE into the fused Wqkv / Wq + Wkv slots with a zero k-bias; the query is
          scaled by 1/8 before the rotation there (:408) and the scores after it here -- a power of two commutes with the
          rotation and with every rounding, so the products are the same bits; its conditioning vectors enter conv1 as extra
          input channels (mh_cond_channels);
  "hf"    stock 'openai/whisper-*' (V28 / V29; transformers models/whisper/modeling_whisper.py) = library arch 2: the wrapper's
          encoder_embedder in front of conv1, + encoder.embed_positions, affine nn.LayerNorm blocks, no rotary embedding,
          decoder_embedder[id] + decoder.embed_positions[position], the nnAudio mel front-end of configs/model/default.yaml.

Numerics contract: as t5_engine.py (bf16 storage = bf16 parameters and GEMM operands, fp32 accumulation / residual
stream / norms / softmax / GELU / logits).  The prompt goes through the batched prefill (RoPE on q and on the cached keys, biased GEMMs; local layers: their own rotary table + the causal band).
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import torch

from . import _lib
from .mel import MelSpectrogram
from .t5_engine import T5Engine, _round_up


@dataclasses.dataclass
class VarWhisperDims:
    d_model: int
    n_heads: int
    n_enc_layers: int
    n_dec_layers: int
    d_ff: int
    d_kv: int = 64

    @property
    def inner(self) -> int:
        return self.n_heads * 64


# openai/whisper-{tiny,base,small} dims, which 'OliBomby/varwhisper-<size>' inherits
# (configuration_mapperatorinator.py:77-78); "test" is a test-only size
VARWHISPER_PRESETS = {
    "test": VarWhisperDims(128, 2, 2, 2, 256),
    "tiny": VarWhisperDims(384, 6, 4, 4, 1536),
    "base": VarWhisperDims(512, 8, 6, 6, 2048),
    "small": VarWhisperDims(768, 12, 12, 12, 3072),
}


def rope_table(n_pos: int, theta: float, dtype: torch.dtype) -> torch.Tensor:
    """fp32 [n_pos][64] = cos(32) | sin(32), computed as VarWhisperRotaryEmbedding.forward does (fp32 inv_freq =
    theta^(-2i/64), fp32 angles, cos / sin cast to the activation dtype), the duplicated halves stored once."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, 64, 2, dtype=torch.int64).to(torch.float32) / 64))
    freqs = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv_freq[None, :]
    return torch.cat([freqs.cos().to(dtype).to(torch.float32), freqs.sin().to(dtype).to(torch.float32)], 1).contiguous()


def whisper_kind(sd: dict) -> str:
    """"var" (fused Wqkv: modeling_varwhisper.py), "rope" (split projections + rotary: modeling_ropewhisper.py) or "hf" (split
    projections + absolute position tables: transformers' Whisper), read off the parameter names."""
    pe = "transformer.model.encoder."
    if pe + "layers.0.self_attn.Wqkv.weight" in sd:
        return "var"
    if pe + "layers.0.self_attn.q_proj.weight" in sd:
        return "hf" if pe + "embed_positions.weight" in sd else "rope"
    raise NotImplementedError("not a Whisper-family state dict (neither self_attn.Wqkv nor self_attn.q_proj under transformer.model.encoder)")


def fuse_split_projections(sd: dict) -> dict:
    """RoPEWhisperAttention / WhisperAttention parameters (q_proj with bias, k_proj WITHOUT, v_proj with bias, out_proj;
    modeling_ropewhisper.py:385-388) under the fused names the packer reads: self_attn.Wqkv = [q; k; v] rows (bias with a zero
    k part), cross_attn.Wq / Wkv = [k; v] / Wo from encoder_attn.*, cross_attn_layer_norm from encoder_attn_layer_norm.  Every
    other entry is passed through."""
    out = {}
    done = set()
    for key in sd:
        if not key.endswith("q_proj.weight"):
            continue
        base = key[:-len("q_proj.weight")]                   # "...layers.N.self_attn." or "...layers.N.encoder_attn."
        q, k, v = sd[base + "q_proj.weight"], sd[base + "k_proj.weight"], sd[base + "v_proj.weight"]
        zeros = lambda w: torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        qb = sd.get(base + "q_proj.bias")
        kb = sd.get(base + "k_proj.bias")
        vb = sd.get(base + "v_proj.bias")
        any_bias = qb is not None or kb is not None or vb is not None
        qb = qb if qb is not None else zeros(q)
        kb = kb if kb is not None else zeros(k)
        vb = vb if vb is not None else zeros(v)
        if base.endswith("self_attn."):
            out[base + "Wqkv.weight"] = torch.cat([q, k, v], 0)
            if any_bias:
                out[base + "Wqkv.bias"] = torch.cat([qb, kb, vb], 0)
            tgt = base
        else:
            tgt = base[:-len("encoder_attn.")] + "cross_attn."
            out[tgt + "Wq.weight"] = q
            out[tgt + "Wkv.weight"] = torch.cat([k, v], 0)
            if any_bias:
                out[tgt + "Wq.bias"] = qb
                out[tgt + "Wkv.bias"] = torch.cat([kb, vb], 0)
        out[tgt + "Wo.weight"] = sd[base + "out_proj.weight"]
        if base + "out_proj.bias" in sd:
            out[tgt + "Wo.bias"] = sd[base + "out_proj.bias"]
        done.update(base + n for n in ("q_proj.weight", "q_proj.bias", "k_proj.weight", "k_proj.bias", "v_proj.weight", "v_proj.bias",
                                       "out_proj.weight", "out_proj.bias"))
    for key, val in sd.items():
        if key not in done:
            out[key.replace("encoder_attn_layer_norm", "cross_attn_layer_norm")] = val
    return out


class PackedVarWhisper:
    """Device-resident packed weights + the MhT5Config (arch 1) / MhT5Weights structs that describe them."""

    def __init__(self, sd: dict, dims: VarWhisperDims, vocab_in: int, vocab_out: int, n_mels: int, in_frames: int, tgt_len: int,
                 dtype: torch.dtype, device, global_rope_theta: float = 10000.0, local_rope_theta: float = 10000.0,
                 global_attn_every_n_layers: int = 1, local_attention: int = 128, decoder_positions: str = "cache"):
        assert dtype in (torch.float32, torch.bfloat16)
        if decoder_positions not in ("cache", "mask"):
            raise ValueError("decoder_positions: 'cache' (transformers 5.x) or 'mask' (transformers 4.57's Whisper)")
        if dims.d_model != dims.n_heads * 64 or dims.d_model % 128 or dims.d_model > 1024:
            raise NotImplementedError("the HIP path of the Whisper family needs 64-wide heads and d_model a multiple of 128 <= 1024")
        self.dims, self.dtype, self.device = dims, dtype, torch.device(device)
        self.kind = whisper_kind(sd)
        if self.kind != "var":
            if global_attn_every_n_layers != 1:
                raise NotImplementedError("local attention layers exist in the VarWhisper fork only")
            sd = fuse_split_projections(sd)
        hf = self.kind == "hf"
        pe, pd = "transformer.model.encoder.", "transformer.model.decoder."
        c1 = sd[pe + "conv1.weight"]
        # what the front-end's first GEMM reads per frame: the mel channels (+ the conditioning channels when the model does not
        # project its encoder input: modeling_mapperatorinator.py:201-202); "hf": the columns of encoder_embedder
        self.n_mels = n_mels
        self.cond_channels = 0 if hf else int(c1.shape[1]) - n_mels
        self.cond_cols = int(sd["encoder_embedder.weight"].shape[1]) - n_mels if hf else 0    # (-> row_bias, as the T5 engine)
        if self.cond_channels < 0 or self.cond_cols < 0:
            raise ValueError(f"the model's input width ({int(c1.shape[1])} conv1 channels) is smaller than n_mels = {n_mels}")
        self.vocab_in, self.vocab_out = vocab_in, vocab_out
        self.n_mels_pad = _round_up(n_mels + self.cond_channels, 32)
        self.in_frames, self.src_len, self.tgt_len = in_frames, (in_frames - 1) // 2 + 1, tgt_len
        self._keep = []
        dev, d = self.device, dims.d_model

        def mat(t, kpad=None):
            t = t.detach().to(torch.float32)
            if kpad is not None and t.shape[1] != kpad:
                t = torch.nn.functional.pad(t, (0, kpad - t.shape[1]))
            t = t.to(dtype).contiguous().to(dev)
            self._keep.append(t)
            return t

        def vec(t):
            t = t.detach().to(dtype).to(torch.float32).contiguous().to(dev)
            self._keep.append(t)
            return t

        def bias(name):
            return vec(sd[name]).data_ptr() if name in sd else None

        def conv(wt, c_pad):   # [d, C, 3] -> [d, 3 * c_pad] tap-major (mh_whisper_frontend layout), K padded to 32
            o, c, _ = wt.shape
            m = torch.nn.functional.pad(wt.detach().float(), (0, 0, 0, c_pad - c)).permute(0, 2, 1).reshape(o, 3 * c_pad)
            return mat(m, _round_up(3 * c_pad, 32))

        eps = 1e-5 if hf else float(torch.finfo(dtype).eps)      # nn.LayerNorm default / nn.RMSNorm(eps=None)
        cfg = _lib.MhT5Config(d, 64, dims.d_ff, dims.n_heads, dims.n_enc_layers, dims.n_dec_layers, vocab_in, vocab_out,
                              n_mels + self.cond_channels, self.n_mels_pad, self.src_len, tgt_len,
                              _lib.MH_BF16 if dtype == torch.bfloat16 else _lib.MH_F32, eps,
                              2 if hf else 1, 0.125, in_frames, int(global_attn_every_n_layers), int(local_attention) // 2)
        cfg.dec_pos_from_mask = 1 if (hf and decoder_positions == "mask") else 0
        w = _lib.MhT5Weights()
        c2 = sd[pe + "conv2.weight"]
        c_in = d if hf else n_mels + self.cond_channels
        if tuple(c1.shape) != (d, c_in, 3) or tuple(c2.shape) != (d, d, 3):
            raise NotImplementedError(f"conv front-end of shape conv1 {tuple(c1.shape)} / conv2 {tuple(c2.shape)} is not on the HIP path: "
                                      f"expected conv1 ({d}, {c_in}, 3) and conv2 ({d}, {d}, 3)")
        if hf:
            # the wrapper's encoder_embedder (modeling_mapperatorinator.py:204-205): [d, n_mels (+ cond)] -> the mel columns, K-padded;
            # the conditioning columns reach the device as a per-chunk row bias (conditioning.ConditioningEmbedders.row_bias)
            w.enc_embed_w = mat(sd["encoder_embedder.weight"][:, :n_mels], self.n_mels_pad).data_ptr()
            w.enc_embed_b = vec(sd["encoder_embedder.bias"]).data_ptr()
            ep, dp = sd[pe + "embed_positions.weight"], sd[pd + "embed_positions.weight"]
            if ep.shape[0] != self.src_len or dp.shape[0] < tgt_len:
                raise ValueError(f"embed_positions hold {ep.shape[0]} encoder / {dp.shape[0]} decoder rows; the engine was asked for "
                                 f"{self.src_len} encoder positions and a target length of {tgt_len}")
            w.enc_pos = vec(ep).data_ptr()
            w.dec_pos = vec(dp[:tgt_len]).data_ptr()
        w.conv1_w = conv(sd[pe + "conv1.weight"], _round_up(c_in, 32) if hf else self.n_mels_pad).data_ptr()
        w.conv1_b = vec(sd[pe + "conv1.bias"]).data_ptr()
        w.conv2_w = conv(sd[pe + "conv2.weight"], d).data_ptr()
        w.conv2_b = vec(sd[pe + "conv2.bias"]).data_ptr()
        w.dec_embed = mat(sd["decoder_embedder.weight"]).data_ptr()
        for l in range(dims.n_enc_layers):
            b = f"{pe}layers.{l}."
            w.enc_ln1[l] = vec(sd[b + "self_attn_layer_norm.weight"]).data_ptr()
            w.enc_qkv[l] = mat(sd[b + "self_attn.Wqkv.weight"]).data_ptr()
            w.enc_qkv_b[l] = bias(b + "self_attn.Wqkv.bias")
            w.enc_o[l] = mat(sd[b + "self_attn.Wo.weight"]).data_ptr()
            w.enc_o_b[l] = bias(b + "self_attn.Wo.bias")
            w.enc_ln2[l] = vec(sd[b + "final_layer_norm.weight"]).data_ptr()
            w.enc_wi[l] = mat(sd[b + "fc1.weight"]).data_ptr()
            w.enc_fc1_b[l] = bias(b + "fc1.bias")
            w.enc_wo[l] = mat(sd[b + "fc2.weight"]).data_ptr()
            w.enc_fc2_b[l] = bias(b + "fc2.bias")
            if hf:
                w.enc_ln1_b[l] = vec(sd[b + "self_attn_layer_norm.bias"]).data_ptr()
                w.enc_ln2_b[l] = vec(sd[b + "final_layer_norm.bias"]).data_ptr()
        w.enc_final_ln = vec(sd[pe + "layer_norm.weight"]).data_ptr()
        if hf:
            w.enc_final_ln_b = vec(sd[pe + "layer_norm.bias"]).data_ptr()
        ckv, ckv_b = [], []
        for l in range(dims.n_dec_layers):
            b = f"{pd}layers.{l}."
            w.dec_ln1[l] = vec(sd[b + "self_attn_layer_norm.weight"]).data_ptr()
            w.dec_qkv[l] = mat(sd[b + "self_attn.Wqkv.weight"]).data_ptr()
            w.dec_qkv_b[l] = bias(b + "self_attn.Wqkv.bias")
            w.dec_o[l] = mat(sd[b + "self_attn.Wo.weight"]).data_ptr()
            w.dec_o_b[l] = bias(b + "self_attn.Wo.bias")
            w.dec_ln2[l] = vec(sd[b + "cross_attn_layer_norm.weight"]).data_ptr()
            w.dec_cq[l] = mat(sd[b + "cross_attn.Wq.weight"]).data_ptr()
            w.dec_cq_b[l] = bias(b + "cross_attn.Wq.bias")
            ckv.append(sd[b + "cross_attn.Wkv.weight"])
            if b + "cross_attn.Wkv.bias" in sd:
                ckv_b.append(sd[b + "cross_attn.Wkv.bias"])
            w.dec_co[l] = mat(sd[b + "cross_attn.Wo.weight"]).data_ptr()
            w.dec_co_b[l] = bias(b + "cross_attn.Wo.bias")
            w.dec_ln3[l] = vec(sd[b + "final_layer_norm.weight"]).data_ptr()
            w.dec_wi[l] = mat(sd[b + "fc1.weight"]).data_ptr()
            w.dec_fc1_b[l] = bias(b + "fc1.bias")
            w.dec_wo[l] = mat(sd[b + "fc2.weight"]).data_ptr()
            w.dec_fc2_b[l] = bias(b + "fc2.bias")
            if hf:
                w.dec_ln1_b[l] = vec(sd[b + "self_attn_layer_norm.bias"]).data_ptr()
                w.dec_ln2_b[l] = vec(sd[b + "cross_attn_layer_norm.bias"]).data_ptr()
                w.dec_ln3_b[l] = vec(sd[b + "final_layer_norm.bias"]).data_ptr()
        w.dec_ckv_all = mat(torch.cat(ckv, 0)).data_ptr()
        if ckv_b:
            assert len(ckv_b) == dims.n_dec_layers
            w.dec_ckv_b_all = vec(torch.cat(ckv_b, 0)).data_ptr()
        w.dec_final_ln = vec(sd[pd + "layer_norm.weight"]).data_ptr()
        if hf:
            w.dec_final_ln_b = vec(sd[pd + "layer_norm.bias"]).data_ptr()
        w.lm_head = mat(sd["transformer.proj_out.weight"]).data_ptr()
        tabs = {}
        for key, n_pos in (("enc", self.src_len), ("dec", tgt_len)):
            for kind, theta in (("", global_rope_theta), ("_local", local_rope_theta if local_rope_theta is not None else global_rope_theta)):
                # "hf" has no rotary embedding: the identity rotation (cos 1, sin 0 -> x * 1 - y * 0 = x bit for bit) through the same kernels
                t = (torch.cat([torch.ones(n_pos, 32), torch.zeros(n_pos, 32)], 1) if hf else rope_table(n_pos, float(theta), dtype)).to(dev)
                self._keep.append(t)
                tabs[key + kind] = t.data_ptr()
        w.enc_rope, w.enc_rope_local, w.dec_rope, w.dec_rope_local = tabs["enc"], tabs["enc_local"], tabs["dec"], tabs["dec_local"]
        self.cfg, self.w = cfg, w

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep)


class VarWhisperEngine(T5Engine):
    """(log-)mel -> conv front-end -> encoder -> cross-K/V -> KV-cached AR decode on one GPU for the three Whisper-family backbones
    (library arch 1: 'OliBomby/varwhisper-*' and 'Tiger14n/ropewhisper-*'; arch 2: 'openai/whisper-*'; `self.kind`)."""

    def __init__(self, state_dict: dict, dims: VarWhisperDims, vocab_in: int, vocab_out: int, n_mels: int = 128,
                 src_len: int = 2048, tgt_len: int = 2560, dtype: torch.dtype = torch.bfloat16, device="cuda",
                 sample_rate: int = 16000, n_fft: int = 1024, hop_length: int = 128, f_min: int = 20, f_max: int = 8000,
                 global_rope_theta: float = 10000.0, local_rope_theta: float = 10000.0, global_attn_every_n_layers: int = 1,
                 local_attention: int = 128, decoder_positions: str = "cache", spectrogram: Optional[dict] = None,
                 options: Optional[dict] = None):
        """`src_len` = mel frames per chunk as in the reference's config (data.src_seq_len); the encoder (and the
        cross-attention) sees (src_len - 1) // 2 + 1 positions.  `spectrogram`: dict(implementation, log_scale, pad_mode)
        overriding the family's default front-end (torchaudio log-mel with reflect padding; "hf": the nnAudio mel of
        configs/model/default.yaml:29-37).  `decoder_positions` ("hf" only): "cache" or "mask" (PackedVarWhisper)."""
        if not torch.cuda.is_available():
            raise RuntimeError("VarWhisperEngine needs a ROCm GPU; there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.dims, self.dtype = dims, dtype
        self.packed = PackedVarWhisper(state_dict, dims, vocab_in, vocab_out, n_mels, src_len, tgt_len, dtype, self.device,
                                       global_rope_theta, local_rope_theta, global_attn_every_n_layers, local_attention,
                                       decoder_positions=decoder_positions)
        self.kind = self.packed.kind
        sp = dict(implementation="nnAudio", log_scale=False, pad_mode="constant") if self.kind == "hf" else \
            dict(implementation="torchaudio", log_scale=True, pad_mode="reflect")
        sp.update(spectrogram or {})
        self.spectrogram = MelSpectrogram(sp["implementation"], sp["log_scale"], sample_rate, n_fft, n_mels, hop_length, f_min, f_max,
                                          sp["pad_mode"]).to(self.device)
        self.hop_length, self.in_frames, self.src_len, self.tgt_len = hop_length, src_len, self.packed.src_len, tgt_len
        self.stream = torch.cuda.Stream(self.device)
        self._ws = {}
        self._own_options(options)

    def mel(self, audio: torch.Tensor) -> torch.Tensor:
        """(B, Ns) fp32 -> (B, in_frames, n_mels_pad) (log-)mel frames in the storage dtype: the time-major transpose of the
        `input_features` the wrapper hands its backbone (modeling_mapperatorinator.py:199-200); columns beyond n_mels are
        zero (K padding, and the slots `encode_mel` fills with the conditioning channels)."""
        p = self.packed
        if audio.shape[1] // self.hop_length + 1 != p.in_frames:
            raise ValueError(f"audio of {audio.shape[1]} samples gives {audio.shape[1] // self.hop_length + 1} frames; "
                             f"this engine was built for src_seq_len={p.in_frames}")
        return self.spectrogram.forward_padded(audio, p.n_mels_pad, self.dtype)

    def encode_mel(self, mel: torch.Tensor, want_f32: bool = False, row_bias: Optional[torch.Tensor] = None):
        """`row_bias`: the conditioning of this batch in the form the model takes it --
        kind "rope" / "var" with conditioning embedders: (B, cond_channels) fp32 values of the conv1 input channels behind the mel
        channels (conditioning.ConditioningEmbedders.channels), written into `mel` by mh_cond_channels;
        kind "hf": (B, d_model) fp32 row bias of encoder_embedder, as in the T5 engine."""
        p = self.packed
        if p.cond_channels > 0:
            if row_bias is None:
                raise ValueError(f"this model's conv1 takes {p.cond_channels} conditioning channels: pass them "
                                 "(conditioning.ConditioningEmbedders.channels; model_generate / MapperatorinatorHIP do)")
            cv = row_bias.to(self.device, torch.float32).contiguous()
            B = mel.shape[0]
            if cv.shape != (B, p.cond_channels):
                raise ValueError(f"conditioning channels must be ({B}, {p.cond_channels}), got {tuple(cv.shape)}")
            rc = self.lib.mh_cond_channels(mel.data_ptr(), B, p.in_frames, p.n_mels_pad, p.n_mels, cv.data_ptr(), p.cond_channels,
                                           _lib.MH_BF16 if self.dtype == torch.bfloat16 else _lib.MH_F32, self._s())
            _lib.check(rc, "mh_cond_channels")
            row_bias = None
        return super().encode_mel(mel, want_f32, row_bias)
