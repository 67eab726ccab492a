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

Numerics contract: as t5_engine.py (bf16 storage = bf16 parameters and GEMM operands, fp32 accumulation / residual
stream / norms / softmax / GELU / logits).  The prompt goes through the batched prefill (RoPE on q and on the cached keys, biased GEMMs; local layers: their own rotary table + the causal band).
"""
from __future__ import annotations

import dataclasses

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


class PackedVarWhisper:
    """Device-resident packed weights + the MhT5Config (arch 1) / MhT5Weights structs that describe them."""

    def __init__(self, sd: dict, dims: VarWhisperDims, vocab_in: int, vocab_out: int, n_mels: int, in_frames: int, tgt_len: int,
                 dtype: torch.dtype, device, global_rope_theta: float = 10000.0, local_rope_theta: float = 10000.0,
                 global_attn_every_n_layers: int = 1, local_attention: int = 128):
        assert dtype in (torch.float32, torch.bfloat16)
        if dims.d_model != dims.n_heads * 64 or dims.d_model % 128 or dims.d_model > 1024:
            raise NotImplementedError("the HIP path of the Whisper family needs 64-wide heads and d_model a multiple of 128 <= 1024")
        self.dims, self.dtype, self.device = dims, dtype, torch.device(device)
        self.vocab_in, self.vocab_out, self.n_mels = vocab_in, vocab_out, n_mels
        self.n_mels_pad = _round_up(n_mels, 32)
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

        eps = float(torch.finfo(dtype).eps)      # nn.RMSNorm(eps=None)
        cfg = _lib.MhT5Config(d, 64, dims.d_ff, dims.n_heads, dims.n_enc_layers, dims.n_dec_layers, vocab_in, vocab_out,
                              n_mels, self.n_mels_pad, self.src_len, tgt_len,
                              _lib.MH_BF16 if dtype == torch.bfloat16 else _lib.MH_F32, eps,
                              1, 0.125, in_frames, int(global_attn_every_n_layers), int(local_attention) // 2)
        w = _lib.MhT5Weights()
        pe, pd = "transformer.model.encoder.", "transformer.model.decoder."
        c1, c2 = sd[pe + "conv1.weight"], sd[pe + "conv2.weight"]
        if c1.shape[1] != n_mels or tuple(c2.shape[:2]) != (d, d) or c1.shape[0] != d or c1.shape[2] != 3 or c2.shape[2] != 3:
            # a checkpoint whose conv1 takes n_mels + conditioning channels (input_features with conditioning embedders,
            # modeling_mapperatorinator.py:104-128) would have its extra channels CROPPED by the padding below
            raise NotImplementedError(f"conv front-end of shape conv1 {tuple(c1.shape)} / conv2 {tuple(c2.shape)} is not on the HIP path: "
                                      f"expected conv1 ({d}, n_mels = {n_mels}, 3) and conv2 ({d}, {d}, 3)")
        w.conv1_w = conv(sd[pe + "conv1.weight"], self.n_mels_pad).data_ptr()
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
        w.enc_final_ln = vec(sd[pe + "layer_norm.weight"]).data_ptr()
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
        w.dec_ckv_all = mat(torch.cat(ckv, 0)).data_ptr()
        if ckv_b:
            assert len(ckv_b) == dims.n_dec_layers
            w.dec_ckv_b_all = vec(torch.cat(ckv_b, 0)).data_ptr()
        w.dec_final_ln = vec(sd[pd + "layer_norm.weight"]).data_ptr()
        w.lm_head = mat(sd["transformer.proj_out.weight"]).data_ptr()
        tabs = {}
        for key, n_pos in (("enc", self.src_len), ("dec", tgt_len)):
            for kind, theta in (("", global_rope_theta), ("_local", local_rope_theta if local_rope_theta is not None else global_rope_theta)):
                t = rope_table(n_pos, float(theta), dtype).to(dev)
                self._keep.append(t)
                tabs[key + kind] = t.data_ptr()
        w.enc_rope, w.enc_rope_local, w.dec_rope, w.dec_rope_local = tabs["enc"], tabs["enc_local"], tabs["dec"], tabs["dec_local"]
        self.cfg, self.w = cfg, w

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep)


class VarWhisperEngine(T5Engine):
    """log-mel -> conv front-end -> RoPE encoder -> cross-K/V -> KV-cached AR decode on one GPU (arch 1 of the library)."""

    def __init__(self, state_dict: dict, dims: VarWhisperDims, vocab_in: int, vocab_out: int, n_mels: int = 128,
                 src_len: int = 2048, tgt_len: int = 2560, dtype: torch.dtype = torch.bfloat16, device="cuda",
                 sample_rate: int = 16000, n_fft: int = 1024, hop_length: int = 128, f_min: int = 20, f_max: int = 8000,
                 global_rope_theta: float = 10000.0, local_rope_theta: float = 10000.0, global_attn_every_n_layers: int = 1,
                 local_attention: int = 128, options: Optional[dict] = None):
        """`src_len` = log-mel frames per chunk as in the reference's config (data.src_seq_len); the encoder (and the
        cross-attention) sees (src_len - 1) // 2 + 1 positions."""
        if not torch.cuda.is_available():
            raise RuntimeError("VarWhisperEngine needs a ROCm GPU; there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.dims, self.dtype = dims, dtype
        self.packed = PackedVarWhisper(state_dict, dims, vocab_in, vocab_out, n_mels, src_len, tgt_len, dtype, self.device,
                                       global_rope_theta, local_rope_theta, global_attn_every_n_layers, local_attention)
        self.spectrogram = MelSpectrogram("torchaudio", True, sample_rate, n_fft, n_mels, hop_length, f_min, f_max,
                                          "reflect").to(self.device)
        self.hop_length, self.in_frames, self.src_len, self.tgt_len = hop_length, src_len, self.packed.src_len, tgt_len
        self.stream = torch.cuda.Stream(self.device)
        self._ws = {}
        self._own_options(options)

    def mel(self, audio: torch.Tensor) -> torch.Tensor:
        """(B, Ns) fp32 -> (B, in_frames, n_mels_pad) log-mel frames in the storage dtype: the time-major transpose of the
        `input_features` the wrapper hands its backbone (modeling_mapperatorinator.py:199-200)."""
        p = self.packed
        if audio.shape[1] // self.hop_length + 1 != p.in_frames:
            raise ValueError(f"audio of {audio.shape[1]} samples gives {audio.shape[1] // self.hop_length + 1} frames; "
                             f"this engine was built for src_seq_len={p.in_frames}")
        return self.spectrogram.forward_padded(audio, p.n_mels_pad, self.dtype)

