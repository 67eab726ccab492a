"""Host side of the osuT5 hot path: weight ingestion from the reference `state_dict()` names into
packed device buffers, relative-bias lookup tables, and the mel -> encode -> cross-KV -> AR-decode
driver around libmapperhip (csrc/t5.hip).

Numerics contract ("rounding points") of the two storage modes:
  fp32 : everything fp32; GEMMs on the exact-f32 MFMA atom (bitwise an fmaf chain).
  bf16 : parameters and GEMM operands are bf16 (`model.to(bfloat16)` semantics for the weights,
         reference osuT5/osuT5/utils/model_utils.py:375-376); accumulation, the residual stream,
         RMSNorm, softmax, GELU and logits stay fp32; activations are rounded to bf16 exactly where
         they become a GEMM operand (after each RMSNorm, q/k/v, attention output, gated FFN hidden).
         This keeps MORE precision than the reference's all-bf16 module (which also rounds the
         residual stream); `oracle/t5.py` restates the same contract on the CPU.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
from typing import Optional

import torch

from . import _lib
from .mel import MelSpectrogram


@dataclasses.dataclass
class T5Dims:
    d_model: int
    d_ff: int
    n_heads: int
    n_enc_layers: int
    n_dec_layers: int
    d_kv: int = 64
    n_buckets: int = 32
    max_distance: int = 128
    eps: float = 1e-6

    @property
    def inner(self) -> int:
        return self.n_heads * self.d_kv


# google/t5-v1_1-{small,base,large} (the backbones `get_backbone_model` accepts under "google/t5",
# modeling_mapperatorinator.py:20-24); "tiny" is a test-only size.
T5_PRESETS = {
    "tiny": T5Dims(d_model=128, d_ff=256, n_heads=2, n_enc_layers=2, n_dec_layers=2),
    "small": T5Dims(d_model=512, d_ff=1024, n_heads=6, n_enc_layers=8, n_dec_layers=8),
    "base": T5Dims(d_model=768, d_ff=2048, n_heads=12, n_enc_layers=12, n_dec_layers=12),
    "large": T5Dims(d_model=1024, d_ff=2816, n_heads=16, n_enc_layers=24, n_dec_layers=24),
}


def relative_position_bucket(rel: torch.Tensor, bidirectional: bool, num_buckets: int, max_distance: int):
    """The published T5 bucketing of `rel = key_pos - query_pos` (int64 tensor), evaluated with the
    same fp32 tensor ops as HF `T5Attention._relative_position_bucket` / the restatement at
    osuT5/osuT5/model/custom_transformers/t5.py:88-141, so bucket boundaries agree bit for bit."""
    ret = torch.zeros_like(rel)
    n = num_buckets
    if bidirectional:
        n //= 2
        ret = ret + (rel > 0).to(torch.long) * n
        rp = rel.abs()
    else:
        rp = -torch.min(rel, torch.zeros_like(rel))
    max_exact = n // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact)
                         * (n - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, n - 1))
    return ret + torch.where(is_small, rp, large)


def rel_bias_tables(enc_table: torch.Tensor, dec_table: torch.Tensor, src_len: int, tgt_len: int, dims: T5Dims):
    """enc [H, 2L-1] indexed by (k - q) + L - 1; dec [H, tgt_len] indexed by distance q - k >= 0."""
    rel = torch.arange(-(src_len - 1), src_len, dtype=torch.long)
    eb = relative_position_bucket(rel, True, dims.n_buckets, dims.max_distance)
    enc = enc_table.float()[eb].t().contiguous()           # [H, 2L-1]
    dist = torch.arange(0, tgt_len, dtype=torch.long)
    db = relative_position_bucket(-dist, False, dims.n_buckets, dims.max_distance)
    dec = dec_table.float()[db].t().contiguous()           # [H, tgt_len]
    return enc, dec


def _round_up(a, b):
    return (a + b - 1) // b * b


class PackedT5:
    """Device-resident packed weights + the MhT5Config / MhT5Weights structs that describe them."""

    def __init__(self, sd: dict, dims: T5Dims, vocab_in: int, vocab_out: int, n_mels: int, src_len: int,
                 tgt_len: int, dtype: torch.dtype, device, enc_operand_dtype: Optional[str] = None):
        """`enc_operand_dtype="mx8"` (BASELINE configs[4] "fp8 MFMA"; bf16 storage only): the encoder blocks' projections and
        the cross-K/V projection also get MX-fp8 copies (OCP e4m3 + E8M0 per 32 k, mx8.quantize_mx8 -- the rule the device
        applies to the activations) and MhT5Config.enc_operand_dtype selects the MX GEMMs.  A reduced-precision mode of its
        own, never the default; the decoder is untouched (its GEMVs are latency-bound)."""
        assert dtype in (torch.float32, torch.bfloat16)
        if enc_operand_dtype not in (None, "mx8"):
            raise ValueError(f"enc_operand_dtype must be None or 'mx8', got {enc_operand_dtype!r}")
        if enc_operand_dtype == "mx8" and (dtype != torch.bfloat16 or dims.d_model % 128 or dims.d_ff % 128):
            raise ValueError("MX-fp8 encoder operands need bf16 storage and d_model / d_ff multiples of 128")
        self.enc_operand_dtype = enc_operand_dtype
        self.dims, self.dtype, self.device = dims, dtype, torch.device(device)
        self.vocab_in, self.vocab_out, self.n_mels = vocab_in, vocab_out, n_mels
        self.n_mels_pad = _round_up(n_mels, 32)
        self.src_len, self.tgt_len = src_len, tgt_len
        self._keep = []
        dev = self.device

        def mat(t, kpad=None):  # GEMM operand: storage dtype, K padded with zeros
            t = t.detach().to(torch.float32)
            if kpad is not None and t.shape[1] != kpad:
                t = torch.nn.functional.pad(t, (0, kpad - t.shape[1]))
            t = t.to(dtype).contiguous().to(dev)
            self._keep.append(t)
            return t

        def mxmat(t):  # MX-fp8 copy of a GEMM operand: the weights as the bf16 path holds them, then quantised
            from .mx8 import quantize_mx8
            q, sc = quantize_mx8(t.detach().to(torch.float32).to(dtype).to(torch.float32).to(dev))
            self._keep += [q, sc]
            return q.data_ptr(), sc.data_ptr()

        def vec(t):  # fp32 vector holding the storage-dtype-rounded values
            t = t.detach().to(dtype).to(torch.float32).contiguous().to(dev)
            self._keep.append(t)
            return t

        def interleave16(a, b):  # wi_0 / wi_1 -> alternating 16-row blocks (MH_EPI_GEGLU layout)
            dff, d = a.shape
            assert dff % 16 == 0
            return torch.stack([a.reshape(dff // 16, 16, d), b.reshape(dff // 16, 16, d)], dim=1).reshape(2 * dff, d)

        pe, pd = "transformer.encoder.", "transformer.decoder."
        cfg = _lib.MhT5Config(dims.d_model, dims.d_kv, dims.d_ff, dims.n_heads, dims.n_enc_layers, dims.n_dec_layers,
                              vocab_in, vocab_out, n_mels, self.n_mels_pad, src_len, tgt_len,
                              _lib.MH_BF16 if dtype == torch.bfloat16 else _lib.MH_F32, dims.eps,
                              0, 1.0, src_len, 0, 0,      # arch 0 = T5
                              _lib.MH_MX8 if enc_operand_dtype == "mx8" else 0)
        w = _lib.MhT5Weights()
        # columns beyond n_mels belong to the conditioning vectors: they reach the device as a per-chunk row bias
        # (conditioning.ConditioningEmbedders.row_bias, mh_t5_encode_cond)
        self.cond_cols = int(sd["encoder_embedder.weight"].shape[1]) - n_mels     # > 0: the model carries conditioning embedders
        w.enc_embed_w = mat(sd["encoder_embedder.weight"][:, :n_mels], self.n_mels_pad).data_ptr()
        w.enc_embed_b = vec(sd["encoder_embedder.bias"]).data_ptr()
        w.dec_embed = mat(sd["decoder_embedder.weight"]).data_ptr()
        enc_tab = sd[pe + "block.0.layer.0.SelfAttention.relative_attention_bias.weight"].detach().to(dtype).float().cpu()
        dec_tab = sd[pd + "block.0.layer.0.SelfAttention.relative_attention_bias.weight"].detach().to(dtype).float().cpu()
        eb, db = rel_bias_tables(enc_tab, dec_tab, src_len, tgt_len, dims)
        eb, db = eb.to(dev), db.to(dev)
        self._keep += [eb, db]
        w.enc_rel_bias, w.dec_rel_bias = eb.data_ptr(), db.data_ptr()
        for l in range(dims.n_enc_layers):
            b = f"{pe}block.{l}."
            a = b + "layer.0.SelfAttention."
            w.enc_ln1[l] = vec(sd[b + "layer.0.layer_norm.weight"]).data_ptr()
            w.enc_qkv[l] = mat(torch.cat([sd[a + "q.weight"], sd[a + "k.weight"], sd[a + "v.weight"]], 0)).data_ptr()
            w.enc_o[l] = mat(sd[a + "o.weight"]).data_ptr()
            w.enc_ln2[l] = vec(sd[b + "layer.1.layer_norm.weight"]).data_ptr()
            f = b + "layer.1.DenseReluDense."
            w.enc_wi[l] = mat(interleave16(sd[f + "wi_0.weight"], sd[f + "wi_1.weight"])).data_ptr()
            w.enc_wo[l] = mat(sd[f + "wo.weight"]).data_ptr()
            if enc_operand_dtype == "mx8":
                w.enc_qkv_mx[l], w.enc_qkv_mxs[l] = mxmat(torch.cat([sd[a + "q.weight"], sd[a + "k.weight"], sd[a + "v.weight"]], 0))
                w.enc_o_mx[l], w.enc_o_mxs[l] = mxmat(sd[a + "o.weight"])
                w.enc_wi_mx[l], w.enc_wi_mxs[l] = mxmat(interleave16(sd[f + "wi_0.weight"], sd[f + "wi_1.weight"]))
                w.enc_wo_mx[l], w.enc_wo_mxs[l] = mxmat(sd[f + "wo.weight"])
        w.enc_final_ln = vec(sd[pe + "final_layer_norm.weight"]).data_ptr()
        ckv = []
        for l in range(dims.n_dec_layers):
            b = f"{pd}block.{l}."
            a = b + "layer.0.SelfAttention."
            x = b + "layer.1.EncDecAttention."
            w.dec_ln1[l] = vec(sd[b + "layer.0.layer_norm.weight"]).data_ptr()
            w.dec_qkv[l] = mat(torch.cat([sd[a + "q.weight"], sd[a + "k.weight"], sd[a + "v.weight"]], 0)).data_ptr()
            w.dec_o[l] = mat(sd[a + "o.weight"]).data_ptr()
            w.dec_ln2[l] = vec(sd[b + "layer.1.layer_norm.weight"]).data_ptr()
            w.dec_cq[l] = mat(sd[x + "q.weight"]).data_ptr()
            ckv += [sd[x + "k.weight"], sd[x + "v.weight"]]
            w.dec_co[l] = mat(sd[x + "o.weight"]).data_ptr()
            w.dec_ln3[l] = vec(sd[b + "layer.2.layer_norm.weight"]).data_ptr()
            f = b + "layer.2.DenseReluDense."
            w.dec_wi[l] = mat(interleave16(sd[f + "wi_0.weight"], sd[f + "wi_1.weight"])).data_ptr()
            w.dec_wo[l] = mat(sd[f + "wo.weight"]).data_ptr()
        w.dec_ckv_all = mat(torch.cat(ckv, 0)).data_ptr()
        if enc_operand_dtype == "mx8":
            w.dec_ckv_all_mx, w.dec_ckv_all_mxs = mxmat(torch.cat(ckv, 0))
        w.dec_final_ln = vec(sd[pd + "final_layer_norm.weight"]).data_ptr()
        w.lm_head = mat(sd["transformer.lm_head.weight"]).data_ptr()
        self.cfg, self.w = cfg, w

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep)


class HostStager:
    """Host -> HBM copies that overlap the kernels of the batch before: the host tensor goes through a pinned buffer and
    a copy stream of its own, the consumer's stream waits on the returned event (nothing overlapped the H2D
    of batch i + 1 with the decode of batch i).  Used by the window scheduler's encode loop and by RequestBatcher."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self._live = []                               # (event, pinned buffer): the buffer outlives its copy

    def stage(self, t: torch.Tensor, dtype=None):
        """-> (device tensor, event or None).  Returns at once; device tensors pass through."""
        if t.device.type == "cuda":
            return (t if dtype is None else t.to(dtype)), None
        self._live = [(e, b) for e, b in self._live if not e.query()]
        pin = torch.empty(t.shape, dtype=dtype or t.dtype, pin_memory=True)
        pin.copy_(t)
        with torch.cuda.stream(self.stream):
            d = pin.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._live.append((ev, pin))
        return d, ev


class T5Engine:
    """mel -> encoder -> cross-KV -> KV-cached AR decode on one GPU."""

    def __init__(self, state_dict: dict, dims: T5Dims, vocab_in: int, vocab_out: int, n_mels: int = 388,
                 src_len: int = 1251, tgt_len: int = 512, dtype: torch.dtype = torch.bfloat16, device="cuda",
                 sample_rate: int = 16000, n_fft: int = 1024, hop_length: int = 128, f_min: int = 0,
                 f_max: int = 8000, log_scale: bool = False, enc_operand_dtype: Optional[str] = None,
                 options: Optional[dict] = None):
        """`options`: this engine's own overrides of the library's tuning options ({"decode_chains": 1, ...}; names in
        include/mapperhip.h) -- other engines in the process keep theirs.  `self.options[name] = v` changes one later."""
        if not torch.cuda.is_available():
            raise RuntimeError("T5Engine needs a ROCm GPU; there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.dims, self.dtype = dims, dtype
        self.packed = PackedT5(state_dict, dims, vocab_in, vocab_out, n_mels, src_len, tgt_len, dtype, self.device,
                               enc_operand_dtype=enc_operand_dtype)
        self.spectrogram = MelSpectrogram("nnAudio", log_scale, sample_rate, n_fft, n_mels, hop_length, f_min,
                                          f_max, "constant").to(self.device)
        self.hop_length, self.src_len, self.tgt_len = hop_length, src_len, tgt_len
        self.stream = torch.cuda.Stream(self.device)
        self._ws = {}
        self._own_options(options)

    def _own_options(self, options: Optional[dict]):
        self.options = _lib.OptionSet(options)

    # `engine.options = OptionSet(...)` (or a dict) REPLACES the set: the packed config must follow, or it would keep the handle
    # of a set that the old object's __del__ has destroyed
    @property
    def options(self) -> "_lib.OptionSet":
        return self._options

    @options.setter
    def options(self, value):
        new = value if isinstance(value, _lib.OptionSet) else _lib.OptionSet(value)
        self.packed.cfg.options = new.handle      # first the config, then drop the old set
        self._options = new

    # ---- buffers ---------------------------------------------------------------------------------
    def _workspace(self, kind: str, nbytes: int) -> torch.Tensor:
        t = self._ws.get(kind)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self._ws[kind] = t
        return t

    def _s(self) -> int:
        return self.stream.cuda_stream

    # ---- stages (all on self.stream; callers bracket with `_enter` / `_leave`) ----------------------
    def _enter(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _leave(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def on_stream(self):
        """Context manager: the engine's stream is current (what callers outside this module use around stage calls)."""
        return torch.cuda.stream(self.stream)

    def synchronize(self):
        torch.cuda.current_stream(self.device).synchronize()

    def mel(self, audio: torch.Tensor) -> torch.Tensor:
        """(B, Ns) fp32 -> (B, L, n_mels_pad) storage dtype (K-padded GEMM operand)."""
        p = self.packed
        if audio.shape[1] // self.hop_length + 1 != p.src_len:
            raise ValueError(f"audio of {audio.shape[1]} samples gives {audio.shape[1] // self.hop_length + 1} frames; "
                             f"this engine was built for src_len={p.src_len}")
        return self.spectrogram.forward_padded(audio, p.n_mels_pad, self.dtype)

    def encode_mel(self, mel: torch.Tensor, want_f32: bool = False, row_bias: Optional[torch.Tensor] = None):
        """`row_bias` (B, d_model) fp32: the conditioning embedders' contribution incl. the bias of encoder_embedder
        (conditioning.ConditioningEmbedders.row_bias); None = the plain projection."""
        p = self.packed
        B = mel.shape[0]
        rb = None
        if row_bias is None and getattr(p, "cond_cols", 0) > 0:
            # a conditioned model without its conditioning would neither raise nor reproduce the reference: refuse
            raise ValueError(f"this model's encoder_embedder has {p.cond_cols} conditioning columns: pass row_bias "
                             "(conditioning.ConditioningEmbedders.row_bias; model_generate / MapperatorinatorHIP do)")
        if row_bias is not None:
            rb = row_bias.to(self.device, torch.float32).contiguous()
            if rb.shape != (B, self.dims.d_model):
                raise ValueError(f"row_bias must be ({B}, {self.dims.d_model}), got {tuple(rb.shape)}")
        need = self.lib.mh_t5_encode_workspace_bytes(C.byref(p.cfg), B)
        ws = self._workspace("enc", need)
        enc = torch.empty((B, p.src_len, self.dims.d_model), dtype=self.dtype, device=self.device)
        enc32 = torch.empty((B, p.src_len, self.dims.d_model), dtype=torch.float32, device=self.device) if want_f32 else None
        rc = self.lib.mh_t5_encode_cond(C.byref(p.cfg), C.byref(p.w), mel.data_ptr(), B, _lib.ptr(rb), enc.data_ptr(),
                                        _lib.ptr(enc32), ws.data_ptr(), ws.numel(), self._s())
        _lib.check(rc, "mh_t5_encode_cond")
        return (enc, enc32) if want_f32 else enc

    def cross_kv(self, enc: torch.Tensor) -> torch.Tensor:
        p = self.packed
        B = enc.shape[0]
        kv = torch.empty((self.dims.n_dec_layers, 2, B, self.dims.n_heads, p.src_len, 64), dtype=self.dtype,
                         device=self.device)
        need = self.lib.mh_t5_cross_kv_workspace_bytes(C.byref(p.cfg), B)     # (> 0 only with MX-fp8 operands: the quantised enc)
        ws = self._workspace("ckv", max(int(need), 256))
        rc = self.lib.mh_t5_cross_kv_ws(C.byref(p.cfg), C.byref(p.w), enc.data_ptr(), B, kv.data_ptr(), ws.data_ptr(), ws.numel(), self._s())
        _lib.check(rc, "mh_t5_cross_kv_ws")
        return kv

    def cross_kv_fp8(self, kv: torch.Tensor) -> torch.Tensor:
        """The packed OCP e4m3 copy of `cross_kv(enc)` (one byte per element + one fp32 scale per (layer, k|v, row,
        head)) that the token steps stream when `MhSampling.cross_kv_fp8` points at it: half the HBM bytes of the
        dominant decode kernel (BASELINE configs[4]).  bf16 storage only; not a parity mode."""
        if self.dtype != torch.bfloat16:
            raise ValueError("the fp8 cross K/V copy needs bf16 storage")
        p = self.packed
        B = kv.shape[2]
        out = torch.empty(self.lib.mh_t5_cross_kv_fp8_bytes(C.byref(p.cfg), B), dtype=torch.uint8, device=self.device)
        rc = self.lib.mh_t5_quantize_cross_kv(C.byref(p.cfg), kv.data_ptr(), B, out.data_ptr(), self._s())
        _lib.check(rc, "mh_t5_quantize_cross_kv")
        return out

    def encode(self, audio: torch.Tensor, want_f32: bool = False, row_bias: Optional[torch.Tensor] = None):
        """audio (B, Ns) on the GPU -> encoder last_hidden_state (final RMSNorm applied)."""
        self._enter()
        with torch.cuda.stream(self.stream):
            out = self.encode_mel(self.mel(audio), want_f32, row_bias)
        self._leave()
        return out

    def decoder_forward(self, cross_kv: torch.Tensor, ids: torch.Tensor, mask: Optional[torch.Tensor] = None):
        """Teacher-forced logits of every position: ids int32 (B, T) on the device -> fp32 (B, T, vocab_out)."""
        p = self.packed
        B, T = ids.shape
        if cross_kv.shape[2] != B:
            raise ValueError(f"cross_kv holds {cross_kv.shape[2]} rows for {B} sequences")
        need = self.lib.mh_t5_forward_workspace_bytes(C.byref(p.cfg), B, T)
        ws = self._workspace("fwd", need)
        logits = torch.empty((B, T, p.vocab_out), dtype=torch.float32, device=self.device)
        rc = self.lib.mh_t5_decoder_forward(C.byref(p.cfg), C.byref(p.w), cross_kv.data_ptr(), B, ids.data_ptr(),
                                            _lib.ptr(mask), T, logits.data_ptr(), ws.data_ptr(), ws.numel(), self._s())
        _lib.check(rc, "mh_t5_decoder_forward")
        return logits

    def decode(self, cross_kv: torch.Tensor, prompt: torch.Tensor, prompt_mask: Optional[torch.Tensor],
               eos_table: torch.Tensor, sampling: _lib.MhSampling, forced: Optional[torch.Tensor] = None,
               dump_logits: bool = False, poll_every: int = 16, kv_fp8: Optional[torch.Tensor] = None):
        """prompt int32 (B, P) on device.  Returns (tokens int32 (B, max_length) device, n_cols int, logits|None).
        Under CFG (sampling.cfg_scale > 1) the B rows are [negative-prompt rows | prompt rows], `cross_kv` holds
        B/2 rows and the logits dump has B/2 rows (the guided scores)."""
        p = self.packed
        B, P = prompt.shape
        cfg = sampling.cfg_scale > 1.0
        if cross_kv.shape[2] != (B // 2 if cfg else B):
            raise ValueError(f"cross_kv holds {cross_kv.shape[2]} rows for a decode batch of {B} (cfg={cfg})")
        flags = getattr(sampling, "host_tok_flags", None)
        if flags is not None:   # keep the device copy alive for the duration of the call
            flags_d = torch.as_tensor(flags, dtype=torch.uint8).to(self.device)
            sampling.tok_flags = flags_d.data_ptr()
        elif sampling.n_cond or sampling.lookback_types_first:
            raise ValueError("sampling needs tok_flags (build it with server.build_sampling)")
        sampling.cross_kv_fp8 = kv_fp8.data_ptr() if kv_fp8 is not None else None
        need = self.lib.mh_t5_decode_workspace_bytes(C.byref(p.cfg), B)
        ws = self._workspace("dec", need)
        maxlen = sampling.max_length
        tokens = torch.full((B, maxlen), int(sampling.pad_id), dtype=torch.int32, device=self.device)
        n_out = torch.zeros(1, dtype=torch.int32, device=self.device)
        logits = (torch.zeros((maxlen, B // 2 if cfg else B, p.vocab_out), dtype=torch.float32, device=self.device)
                  if dump_logits else None)
        rc = self.lib.mh_t5_generate(C.byref(p.cfg), C.byref(p.w), cross_kv.data_ptr(), B, prompt.data_ptr(),
                                     _lib.ptr(prompt_mask), P, eos_table.data_ptr(), C.byref(sampling),
                                     tokens.data_ptr(), n_out.data_ptr(), _lib.ptr(logits), _lib.ptr(forced),
                                     ws.data_ptr(), ws.numel(), poll_every, self._s())
        _lib.check(rc, "mh_t5_generate")
        return tokens, n_out, logits

    def generate_beam(self, audio: torch.Tensor, prompt: torch.Tensor, prompt_mask: Optional[torch.Tensor], eos_ids,
                      sampling: _lib.MhSampling, num_beams: int, row_bias: Optional[torch.Tensor] = None,
                      length_penalty: float = 1.0, early_stopping=False, negative_prompt: Optional[torch.Tensor] = None,
                      sample_fn=None, use_kernel: Optional[bool] = None):
        """mel -> encoder -> cross K/V, then HF-style beam search over the step-wise decode entry (beam.py).  Returns
        dict(tokens=int64 CPU (B, n_cols), n_cols, logits=None) like `generate`.  `negative_prompt` with sampling.cfg_scale > 1:
        classifier-free guidance under beams (the doubled batch of modeling_mapperatorinator.py:243-254; beam.py).  With
        `sampling.do_sample` the continuations are drawn (beam-sample): `sample_fn(probs, k)` or torch.multinomial on the device."""
        from .beam import beam_search
        audio = audio.to(self.device, torch.float32)
        if (negative_prompt is not None) != (sampling.cfg_scale > 1.0):
            raise ValueError("negative_prompt and sampling.cfg_scale > 1 go together")
        if negative_prompt is not None:
            n = negative_prompt.shape[1]
            if n > prompt.shape[1]:
                raise ValueError("negative prompt longer than the prompt")
            neg = prompt.clone()
            neg[:, :n] = negative_prompt.to(prompt.dtype)
            prompt = torch.cat([neg, prompt], 0)
            if prompt_mask is not None:      # (the negative rows attend under the prompt's mask: see `generate`)
                prompt_mask = torch.cat([prompt_mask, prompt_mask], 0)
        self._enter()
        with torch.cuda.stream(self.stream):
            kv = self.cross_kv(self.encode_mel(self.mel(audio), row_bias=row_bias))
        self._leave()
        out = beam_search(self, kv, prompt, prompt_mask, eos_ids, sampling, num_beams, length_penalty, early_stopping,
                          sample_fn=sample_fn, use_kernel=use_kernel)
        return dict(tokens=out.cpu(), n_cols=int(out.shape[1]), logits=None)

    def generate(self, audio: torch.Tensor, prompt: torch.Tensor, prompt_mask: Optional[torch.Tensor],
                 eos_ids, sampling: _lib.MhSampling, forced: Optional[torch.Tensor] = None,
                 dump_logits: bool = False, poll_every: int = 16, negative_prompt: Optional[torch.Tensor] = None,
                 negative_mask: Optional[torch.Tensor] = None, cross_kv_fp8: bool = False,
                 row_bias: Optional[torch.Tensor] = None, encoder_states: Optional[torch.Tensor] = None):
        """Full hot path for one batch of chunks.  `encoder_states` (B, src_len, d_model): the encoder's last_hidden_state given by
        the caller (`generate(encoder_outputs=...)` of the reference's signature) -- mel and encoder are skipped, `audio` may be None.  `cross_kv_fp8`: the token steps stream the e4m3 copy of the
        cross-attention K / V (see `cross_kv_fp8()`).  Inputs may be CPU tensors (copied like
        server.py:86-87 does).  Returns dict(tokens=int64 CPU (B, n_cols), logits=..., n_cols=int).
        With `negative_prompt` (classifier-free guidance) the decode batch is doubled the way the reference's
        prepare_inputs_for_generation does it (modeling_mapperatorinator.py:243-254): the first half carries the
        negative prompt over the first columns of the prompt; the returned rows are the prompt rows."""
        dev = self.device
        if encoder_states is None:
            audio = audio.to(dev, torch.float32)
        else:
            encoder_states = encoder_states.to(dev, self.dtype).contiguous()
            if tuple(encoder_states.shape) != (prompt.shape[0], self.packed.src_len, self.dims.d_model):
                raise ValueError(f"encoder_states must be ({prompt.shape[0]}, {self.packed.src_len}, {self.dims.d_model}), got {tuple(encoder_states.shape)}")
        G = prompt.shape[0]
        cfg = negative_prompt is not None
        if cfg != (sampling.cfg_scale > 1.0):
            raise ValueError("negative_prompt and sampling.cfg_scale > 1 go together")
        if cfg:
            n = negative_prompt.shape[1]
            if n > prompt.shape[1]:
                raise ValueError("negative prompt longer than the prompt")
            neg = prompt.clone()
            neg[:, :n] = negative_prompt.to(prompt.dtype)
            prompt = torch.cat([neg, prompt], 0)
            if prompt_mask is not None:
                # `negative_mask` is accepted for signature parity and deliberately unused: in the reference the
                # kwarg `negative_prompt_attention_mask` is swallowed by HF `generate()` (a named parameter of its
                # own) before prepare_inputs_for_generation runs, so the negative rows attend under the prompt's mask
                prompt_mask = torch.cat([prompt_mask, prompt_mask], 0)
            if forced is not None:
                forced = torch.cat([forced, forced], 0)
        prompt_d = prompt.to(dev, torch.int32).contiguous()
        mask_d = prompt_mask.to(dev).to(torch.uint8).contiguous() if prompt_mask is not None else None
        forced_d = forced.to(dev, torch.int32).contiguous() if forced is not None else None
        eos_table = torch.zeros(self.packed.vocab_out, dtype=torch.uint8)
        eos_table[torch.as_tensor(sorted(set(int(e) for e in eos_ids if 0 <= int(e) < self.packed.vocab_out)),
                                  dtype=torch.long)] = 1
        eos_table = eos_table.to(dev)
        self._enter()
        with torch.cuda.stream(self.stream):
            enc = encoder_states if encoder_states is not None else self.encode_mel(self.mel(audio), row_bias=row_bias)
            kv = self.cross_kv(enc)
            kv8 = self.cross_kv_fp8(kv) if cross_kv_fp8 else None
            tokens, n_out, logits = self.decode(kv, prompt_d, mask_d, eos_table, sampling, forced_d, dump_logits,
                                                poll_every, kv_fp8=kv8)
        self._leave()
        torch.cuda.current_stream(dev).synchronize()
        n_cols = int(n_out.item()) if forced is None else sampling.max_length
        if cfg:
            tokens = tokens[G:]
        return dict(tokens=tokens[:, :n_cols].to(torch.int64).cpu(), n_cols=n_cols,
                    logits=None if logits is None else logits[:n_cols])
