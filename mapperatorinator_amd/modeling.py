"""`MapperatorinatorHIP`: the boundary object B2.

Stands where the reference's `Mapperatorinator` (osuT5/osuT5/model/modeling_mapperatorinator.py:60-353)
stands for the *generate* path: same attributes the callers read (`.device`, `.dtype`,
`.config.{max_target_positions, max_source_positions, hidden_size, vocab_size, ...}`,
`.spectrogram`, `.generate(**kwargs)`), but the arithmetic underneath is libmapperhip.

Four backbones are wired (round 6: + 'Tiger14n/ropewhisper-*' = V30 / V31 and stock 'openai/whisper-*' = V28 / V29, both through
whisper_engine.py): the T5 configuration of the north star (input_features=False, project_encoder_input=True,
embed_decoder_input=True: configs/model/default.yaml:1-4 + t5_small_v9.yaml:5) and the Whisper-family one of the released
V30-V32 checkpoints ('OliBomby/varwhisper-*': input_features=True, project_encoder_input=False, torchaudio log-mel --
configs/model/varwhisper_{small,base}_v3.yaml; whisper_engine.py); anything else raises NotImplementedError instead of
silently running something different.
"""
from __future__ import annotations

import types
from typing import Optional

import torch

from . import _lib
from .conditioning import ConditioningEmbedders
from .t5_engine import T5Dims, T5Engine, T5_PRESETS
from .whisper_engine import VARWHISPER_PRESETS, VarWhisperDims, VarWhisperEngine


def dims_from_backbone_config(bc) -> T5Dims:
    """HF T5Config-like object (d_model, d_ff, d_kv, num_heads, num_layers, num_decoder_layers, ...)."""
    if getattr(bc, "d_kv", 64) != 64:
        raise NotImplementedError("HIP attention kernels are built for d_kv = 64")
    if not getattr(bc, "is_gated_act", True):
        raise NotImplementedError("only the gated-gelu (T5 v1.1) FFN is built")
    return T5Dims(d_model=bc.d_model, d_ff=bc.d_ff, n_heads=bc.num_heads, n_enc_layers=bc.num_layers,
                  n_dec_layers=bc.num_decoder_layers, d_kv=64,
                  n_buckets=bc.relative_attention_num_buckets,
                  max_distance=bc.relative_attention_max_distance, eps=bc.layer_norm_epsilon)


class HIPDecodeCache:
    """`past_key_values` of the HIP path: the decoder's self-attention K / V of every position fed so far (inside the engine's decode
    workspace, owned by this object), the cross K / V of the call's audio and the count of positions seen -- the role
    `MapperatorinatorCache` (osuT5/osuT5/inference/cache_utils.py:8-20, around HF's EncoderDecoderCache) plays between two
    `forward` calls of the reference.  `reorder_cache(beam_idx)` is that class's method (:16-20)."""

    def __init__(self, engine, cross_kv: torch.Tensor, rows: int, prompt_mask: Optional[torch.Tensor]):
        import ctypes as C
        p = engine.packed
        if rows > 64:
            raise ValueError(f"{rows} rows exceed the engine's 64-row decode batch")
        if cross_kv.shape[2] != rows:
            raise ValueError(f"cross K/V holds {cross_kv.shape[2]} rows, the decoder ids {rows}")
        self.engine, self.cross_kv, self.rows, self.seen = engine, cross_kv, int(rows), 0
        self.max_positions = int(p.tgt_len)
        dev = engine.device
        self.ws = torch.empty(int(engine.lib.mh_t5_decode_workspace_bytes(C.byref(p.cfg), self.rows)), dtype=torch.uint8, device=dev)
        self._scratch = None
        # the prompt's attention mask is fixed by the FIRST call (HF appends ones behind it)
        self.mask = None if prompt_mask is None else prompt_mask.to(dev).to(torch.uint8).contiguous()
        self.P = 0 if self.mask is None else int(self.mask.shape[1])

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.seen

    def append(self, ids: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
        import ctypes as C
        eng = self.engine
        lib, p, dev = eng.lib, eng.packed, eng.device
        if ids.shape[0] != self.rows:
            raise ValueError(f"{ids.shape[0]} rows given, the cache holds {self.rows}")
        T = int(ids.shape[1])
        if self.seen + T > self.max_positions:
            raise ValueError(f"{self.seen} + {T} positions exceed the cache ({self.max_positions})")
        if attention_mask is not None and self.seen > 0:
            m = attention_mask.to(dev).to(torch.uint8)
            if m.shape[1] != self.seen + T:
                raise ValueError("decoder_attention_mask must cover the cached positions and the new ones")
            if self.P and not torch.equal(m[:, :self.P], self.mask):
                raise ValueError("the mask of the cached prompt positions changed between calls")
            if not bool(m[:, max(self.P, 0):].all()):
                raise NotImplementedError("masked positions behind the first call's prompt are not built (HF's loop appends ones)")
        P = self.P if self.mask is not None else max(self.seen + T, 1)
        ids = ids.to(dev, torch.int32)
        out = torch.empty((T, self.rows, p.vocab_out), dtype=torch.float32, device=dev)
        eng._enter()
        with torch.cuda.stream(eng.stream):
            for t in range(T):
                rc = lib.mh_t5_step(C.byref(p.cfg), C.byref(p.w), self.cross_kv.data_ptr(), self.rows, 1, ids[:, t].contiguous().data_ptr(),
                                    self.seen + t, _lib.ptr(self.mask), P, out[t].data_ptr(), self.ws.data_ptr(), self.ws.numel(), eng._s())
                _lib.check(rc, "mh_t5_step")
        eng._leave()
        eng.synchronize()
        self.seen += T
        return out.transpose(0, 1).contiguous()

    def reorder_cache(self, beam_idx: torch.Tensor):
        """Rows follow `beam_idx` (cache_utils.py:16-20: `index_select(0, beam_idx)` on every self-attention tensor; the cross K / V
        rows are per row here, so they are gathered too)."""
        import ctypes as C
        eng = self.engine
        lib, p, dev = eng.lib, eng.packed, eng.device
        idx = torch.as_tensor(beam_idx).to(dev, torch.int32).contiguous()
        if idx.numel() != self.rows:
            raise ValueError("beam_idx must name one source row per cache row")
        if self._scratch is None:
            self._scratch = torch.empty(int(lib.mh_t5_reorder_cache_scratch_bytes(C.byref(p.cfg), self.rows, self.max_positions)),
                                        dtype=torch.uint8, device=dev)
        eng._enter()
        with torch.cuda.stream(eng.stream):
            if self.seen > 0:
                rc = lib.mh_t5_reorder_cache(C.byref(p.cfg), self.rows, idx.data_ptr(), self.seen, self.ws.data_ptr(), self.ws.numel(),
                                             self._scratch.data_ptr(), self._scratch.numel(), eng._s())
                _lib.check(rc, "mh_t5_reorder_cache")
            self.cross_kv = self.cross_kv.index_select(2, idx.to(torch.int64))
            if self.mask is not None:
                self.mask = self.mask.index_select(0, idx.to(torch.int64)).contiguous()
        eng._leave()
        eng.synchronize()
        return self


class MapperatorinatorHIP:
    main_input_name = "frames"

    def __init__(self, state_dict: dict, dims: T5Dims, *, vocab_size_in: int, vocab_size_out: int,
                 n_mels: int = 388, src_seq_len: int = 1251, tgt_seq_len: int = 512,
                 dtype: torch.dtype = torch.bfloat16, device="cuda", sample_rate: int = 16000, n_fft: int = 1024,
                 hop_length: int = 128, f_min: int = 0, f_max: int = 8000, spectrogram_log_scale: bool = False,
                 pad_token_id: int = 0, bos_token_id: int = 1, eos_token_id: int = 2, backbone_options: Optional[dict] = None,
                 enc_operand_dtype: Optional[str] = None, options: Optional[dict] = None):
        """`options`: engine-owned overrides of the library's tuning options (T5Engine).
        `dims`: T5Dims (google/t5-v1_1-*) or VarWhisperDims (OliBomby/varwhisper-*; `src_seq_len` then counts log-mel
        frames as the reference's data.src_seq_len does, `backbone_options` = global_rope_theta / local_rope_theta /
        global_attn_every_n_layers / local_attention)."""
        self.is_whisper = isinstance(dims, VarWhisperDims)
        if self.is_whisper and enc_operand_dtype is not None:
            raise NotImplementedError("MX-fp8 encoder operands are built for the T5 backbone only")
        if self.is_whisper:
            self.engine = VarWhisperEngine(state_dict, dims, vocab_size_in, vocab_size_out, n_mels, src_seq_len, tgt_seq_len,
                                           dtype, device, sample_rate, n_fft, hop_length, f_min, f_max, options=options, **(backbone_options or {}))
        else:
            self.engine = T5Engine(state_dict, dims, vocab_size_in, vocab_size_out, n_mels, src_seq_len, tgt_seq_len,
                                   dtype, device, sample_rate, n_fft, hop_length, f_min, f_max, spectrogram_log_scale,
                                   enc_operand_dtype=enc_operand_dtype, options=options)
        self._source_state_dict = state_dict      # caller-owned tensors under the reference's parameter names (not copied)
        # difficulty / mapper / song-position / style embedders, if the state dict carries them (host side; their output
        # reaches the device as a per-chunk row bias of the encoder input projection)
        self.cond = ConditioningEmbedders(state_dict, n_mels)
        self.device = self.engine.device
        self.dtype = dtype
        self.spectrogram = self.engine.spectrogram
        self.config = types.SimpleNamespace(
            hidden_size=dims.d_model, num_attention_heads=dims.n_heads, num_hidden_layers=dims.n_enc_layers,
            # (Whisper family: config.max_source_positions = src_seq_len // 2, configuration_mapperatorinator.py:107)
            max_source_positions=src_seq_len // 2 if self.is_whisper else src_seq_len, max_target_positions=tgt_seq_len, vocab_size=vocab_size_out,
            vocab_size_in=vocab_size_in, n_mels=n_mels, hop_length=hop_length, sample_rate=sample_rate,
            pad_token_id=pad_token_id, bos_token_id=bos_token_id, eos_token_id=eos_token_id,
            is_encoder_decoder=True,
            backbone_model_name={"var": "OliBomby/varwhisper(hip)", "rope": "Tiger14n/ropewhisper(hip)", "hf": "openai/whisper(hip)"}[self.engine.kind]
            if self.is_whisper else "google/t5-v1_1(hip)",
            # the fields `get_cache` / `MapperatorinatorCache` read (inference/cache_utils.py:23-35): the engine owns its
            # caches, they are here so that code inspecting the reference config finds them
            num_hidden_layers_decoder=dims.n_dec_layers, d_model=dims.d_model, d_kv=dims.d_kv, num_heads=dims.n_heads,
            num_layers=dims.n_enc_layers, num_decoder_layers=dims.n_dec_layers, d_ff=dims.d_ff,
            torch_dtype=dtype, input_features=self.is_whisper,
            project_encoder_input=(not self.is_whisper) or self.engine.kind == "hf", embed_decoder_input=True)

    # ---- construction from the reference object ---------------------------------------------------
    @classmethod
    def from_reference(cls, model, dtype: Optional[torch.dtype] = None, device="cuda"):
        """`model`: a reference `Mapperatorinator` with a google/t5 backbone (possibly on the CPU)."""
        cfg = model.config
        name = str(cfg.backbone_model_name)
        if name.startswith(("OliBomby/varwhisper", "Tiger14n/ropewhisper", "openai/whisper")):
            hf = name.startswith("openai/whisper")
            if not cfg.input_features or not cfg.embed_decoder_input or cfg.input_raw_wave or cfg.do_style_embed:
                raise NotImplementedError("HIP path of the Whisper family implements input_features=True, embed_decoder_input=True, "
                                          "do_style_embed=False (configs/model/whisper_*.yaml, varwhisper_*_v3.yaml)")
            if bool(cfg.project_encoder_input) != hf:
                # the fork configs feed the (log-)mel channels straight into conv1 (whisper_small_v2.yaml:7, varwhisper_*_v3.yaml:7),
                # the stock-Whisper ones (whisper_{base,small}.yaml) keep default.yaml's encoder projection
                raise NotImplementedError(f"{name} with project_encoder_input={cfg.project_encoder_input} is not a released wiring")
            bc = cfg.backbone_config
            dims = VarWhisperDims(bc.d_model, bc.encoder_attention_heads, bc.encoder_layers, bc.decoder_layers, bc.encoder_ffn_dim)
            if bc.decoder_attention_heads != bc.encoder_attention_heads or bc.decoder_ffn_dim != bc.encoder_ffn_dim or bc.activation_function != "gelu":
                raise NotImplementedError("encoder and decoder of the HIP Whisper path share heads / ffn width; activation gelu")
            if getattr(bc, "scale_embedding", False):
                raise NotImplementedError("scale_embedding=True is not built")
            opts = dict(spectrogram=dict(implementation=cfg.spectrogram_implementation, log_scale=bool(cfg.spectrogram_log_scale),
                                         pad_mode=cfg.pad_mode))
            if name.startswith("OliBomby/varwhisper"):
                opts.update(global_rope_theta=bc.global_rope_theta, local_rope_theta=bc.local_rope_theta,
                            global_attn_every_n_layers=bc.global_attn_every_n_layers, local_attention=bc.local_attention)
            elif hf:
                # decoder positions of left-padded rows follow the transformers the reference RUNS UNDER: 4.x (its pin, 4.57.3) derives
                # them from the attention mask in Whisper's own prepare_inputs_for_generation, 5.x uses cache positions
                try:
                    import transformers
                    major = int(str(transformers.__version__).split(".")[0])
                except Exception:      # no transformers next to us: the reference's pin
                    major = 4
                opts.update(decoder_positions="mask" if major < 5 else "cache")
            if name.startswith("Tiger14n/ropewhisper"):
                # rope_type "dynamic" (NTK): the base only changes for positions beyond max_position_embeddings
                # (modeling_ropewhisper.py:299-305), which a StaticCache of max_target_positions never reaches; factor 1.0
                if float(getattr(bc, "rope_encoder_scaling_factor", 1.0)) != 1.0 or float(getattr(bc, "rope_decoder_scaling_factor", 1.0)) != 1.0 \
                        or str(getattr(bc, "rope_type", "dynamic")) not in ("dynamic", "default"):
                    raise NotImplementedError("RoPEWhisper with a rope scaling factor != 1 or a rope_type other than dynamic / default")
            return cls(model.state_dict(), dims, vocab_size_in=cfg.vocab_size_in, vocab_size_out=cfg.vocab_size, n_mels=cfg.n_mels,
                       src_seq_len=2 * cfg.max_source_positions, tgt_seq_len=cfg.max_target_positions, dtype=dtype or model.dtype,
                       device=device, sample_rate=cfg.sample_rate, n_fft=cfg.n_fft, hop_length=cfg.hop_length, f_min=cfg.f_min,
                       f_max=cfg.f_max, spectrogram_log_scale=bool(cfg.spectrogram_log_scale), pad_token_id=cfg.pad_token_id,
                       bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id, backbone_options=opts)
        if not str(cfg.backbone_model_name).startswith("google/t5"):
            raise NotImplementedError("google/t5, OliBomby/varwhisper, Tiger14n/ropewhisper and openai/whisper backbones run on the HIP "
                                      f"path; {name} (nwhisper / moonshine) does not")
        if cfg.input_features or not cfg.project_encoder_input or not cfg.embed_decoder_input or cfg.input_raw_wave:
            raise NotImplementedError("HIP path implements input_features=False, project_encoder_input=True, "
                                      "embed_decoder_input=True")
        bc = cfg.backbone_config
        return cls(model.state_dict(), dims_from_backbone_config(bc), vocab_size_in=cfg.vocab_size_in,
                   vocab_size_out=cfg.vocab_size, n_mels=cfg.n_mels, src_seq_len=cfg.max_source_positions,
                   tgt_seq_len=cfg.max_target_positions, dtype=dtype or model.dtype, device=device,
                   sample_rate=cfg.sample_rate, n_fft=cfg.n_fft, hop_length=cfg.hop_length, f_min=cfg.f_min,
                   f_max=cfg.f_max, spectrogram_log_scale=cfg.spectrogram_log_scale,
                   pad_token_id=cfg.pad_token_id, bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id)

    @classmethod
    def from_preset(cls, name: str, state_dict: dict, **kw):
        """name: a T5 preset (tiny / small / base / large) or "varwhisper-<test|tiny|base|small>"."""
        if name.startswith("varwhisper-"):
            return cls(state_dict, VARWHISPER_PRESETS[name[len("varwhisper-"):]], **kw)
        return cls(state_dict, T5_PRESETS[name], **kw)

    # nn.Module-ish conveniences the reference callers use
    def eval(self):
        return self

    def state_dict(self):
        """The parameters under the reference's names (`Mapperatorinator.state_dict()` keys: encoder_embedder.*,
        decoder_embedder.weight, transformer.*), i.e. what this object was built from -- the packed device copies are
        derived data (t5_engine.PackedT5)."""
        return dict(self._source_state_dict)

    def get_encoder(self):
        """`Mapperatorinator.get_encoder()` (modeling_mapperatorinator.py:333-353): a callable taking the raw audio
        (`frames`) and returning an object with `.last_hidden_state` -- spectrogram, input projection and the T5 encoder
        stack with its final RMSNorm, on the HIP engine."""
        eng = self.engine

        class _Encoder:
            main_input_name = "frames"

            def __call__(self_inner, frames=None, **kw):
                if frames is None:
                    raise ValueError("frames (raw audio, (B, samples)) is required")
                enc = eng.encode(frames.to(eng.device, torch.float32), row_bias=self._row_bias(frames.shape[0], kw))
                return types.SimpleNamespace(last_hidden_state=enc, hidden_states=None, attentions=None)
        return _Encoder()

    def prepare_inputs_for_generation(self, decoder_input_ids, past_key_values=None, use_cache=None, encoder_outputs=None,
                                      decoder_attention_mask=None, cache_position=None, negative_prompt=None,
                                      negative_prompt_attention_mask=None, **kwargs):
        """The batch layout the reference builds for classifier-free guidance (modeling_mapperatorinator.py:230-253):
        the batch is doubled, the NEGATIVE prompt overwrites the first columns of the first half, masks and encoder
        states follow.  `generate()` here applies exactly this layout inside the engine; the method exists so that code
        written against the reference object (and the parity test) can see it."""
        if negative_prompt is not None:
            decoder_input_ids = decoder_input_ids.repeat((2, 1))
            decoder_input_ids[:decoder_input_ids.shape[0] // 2, :negative_prompt.shape[1]] = negative_prompt
            if decoder_attention_mask is not None:
                decoder_attention_mask = decoder_attention_mask.repeat((2, 1))
                if negative_prompt_attention_mask is not None:
                    half = decoder_attention_mask.shape[0] // 2
                    decoder_attention_mask[:half, :negative_prompt_attention_mask.shape[1]] = negative_prompt_attention_mask
            if encoder_outputs is not None:
                enc = getattr(encoder_outputs, "last_hidden_state", encoder_outputs)
                encoder_outputs = types.SimpleNamespace(last_hidden_state=enc.repeat((2, 1, 1)))
        out = dict(input_ids=decoder_input_ids, decoder_input_ids=decoder_input_ids, past_key_values=past_key_values,
                   use_cache=use_cache, encoder_outputs=encoder_outputs, decoder_attention_mask=decoder_attention_mask,
                   cache_position=cache_position)
        for k, v in kwargs.items():
            out.setdefault(k, v)
        return out

    def to(self, *a, **k):
        return self

    def _row_bias(self, batch, kw):
        """The conditioning embedders' per-chunk bias of the encoder input projection (None without embedders); `kw` may
        carry beatmap_idx / difficulty / mapper_idx / song_position as the reference's forward takes them."""
        if not self.cond.active:
            return None
        vec = self.cond.vectors(batch, beatmap_idx=kw.get("beatmap_idx"), difficulty=kw.get("difficulty"),
                                mapper_idx=kw.get("mapper_idx"), song_position=kw.get("song_position"))
        # project_encoder_input = false ('Tiger14n/ropewhisper-*'): the vectors are conv1 input channels, else a row bias of the projection
        return self.cond.channels(vec, self.dtype) if self.cond.as_channels else self.cond.row_bias(vec, self.dtype)

    # ---- B2: the two calls the reference makes on the model object -----------------------------------------
    def _cross_kv(self, frames, encoder_outputs, row_bias=None):
        eng = self.engine
        if encoder_outputs is not None:
            enc = getattr(encoder_outputs, "last_hidden_state", encoder_outputs)
            if isinstance(enc, (tuple, list)):
                enc = enc[0]
            return eng.cross_kv(enc.to(eng.device, eng.dtype).contiguous())
        if frames is None:
            raise ValueError("either frames (raw audio, (B, samples)) or encoder_outputs is required")
        return eng.cross_kv(eng.encode_mel(eng.mel(frames.to(eng.device, torch.float32)), row_bias=row_bias))

    @torch.no_grad()
    def forward(self, frames=None, decoder_input_ids=None, decoder_attention_mask=None, encoder_outputs=None, past_key_values=None,
                use_cache=None, cache_position=None, **unused):
        """Teacher-forced logits (B, T, vocab) fp32 -- `Mapperatorinator.forward` (modeling_mapperatorinator.py:174-228)
        as `model_forward` uses it (server.py:160-181).  `frames` is raw audio, as in the reference (the spectrogram is
        part of the model, :175).  Returns an object with `.logits` (+ `.encoder_last_hidden_state=None`).
        With `use_cache=True` / `past_key_values=<HIPDecodeCache>` the call is INCREMENTAL, as HF's generation loop makes it
        (:186-228 hand `past_key_values` / `cache_position` to the transformer): the ids given are appended behind the positions the
        cache has seen (one mh_t5_step per column), the logits of exactly those columns come back and `.past_key_values` is the cache
        to pass to the next call.  `cache_position`, when given, must be the positions the cache expects next."""
        for k in ("labels", "decoder_inputs_embeds", "inputs_embeds"):
            if unused.get(k) is not None:
                raise NotImplementedError(f"forward({k}=...) is not part of the inference seam")
        if decoder_input_ids is None:
            raise ValueError("decoder_input_ids is required")
        eng = self.engine
        if past_key_values is not None or use_cache:
            cache = past_key_values
            if cache is None:
                row_bias = self._row_bias(decoder_input_ids.shape[0], unused) if encoder_outputs is None else None
                eng._enter()
                with torch.cuda.stream(eng.stream):
                    kv = self._cross_kv(frames, encoder_outputs, row_bias)
                eng._leave()
                cache = HIPDecodeCache(eng, kv, decoder_input_ids.shape[0], decoder_attention_mask)
            elif not isinstance(cache, HIPDecodeCache):
                raise TypeError("past_key_values must be the HIPDecodeCache a previous forward(use_cache=True) returned")
            if cache_position is not None:
                want = torch.arange(cache.seen, cache.seen + decoder_input_ids.shape[1])
                if not torch.equal(torch.as_tensor(cache_position).cpu().to(torch.int64).view(-1), want):
                    raise ValueError(f"cache_position {cache_position} does not continue the cache ({cache.seen} positions seen)")
            logits = cache.append(decoder_input_ids, decoder_attention_mask)
            return types.SimpleNamespace(logits=logits, encoder_last_hidden_state=None, past_key_values=cache, loss=None)
        ids = decoder_input_ids.to(eng.device, torch.int32).contiguous()
        mask = (decoder_attention_mask.to(eng.device).to(torch.uint8).contiguous()
                if decoder_attention_mask is not None else None)
        row_bias = self._row_bias(ids.shape[0], unused) if encoder_outputs is None else None
        eng._enter()
        with torch.cuda.stream(eng.stream):
            logits = eng.decoder_forward(self._cross_kv(frames, encoder_outputs, row_bias), ids, mask)
        eng._leave()
        return types.SimpleNamespace(logits=logits, encoder_last_hidden_state=None, past_key_values=None, loss=None)

    __call__ = forward

    @torch.no_grad()
    def generate(self, inputs=None, frames=None, decoder_input_ids=None, decoder_attention_mask=None,
                 negative_prompt=None, negative_prompt_attention_mask=None, encoder_outputs=None, logits_processor=None,
                 eos_token_id=None, do_sample=False, num_beams=1, top_k=0, top_p=1.0, max_length=None,
                 pad_token_id=None, seed=None, **unused):
        """`model.generate(**model_kwargs, **generate_kwargs, logits_processor=..., eos_token_id=...)` exactly as the
        reference's `model_generate` calls it (server.py:143-151): the processor OBJECTS are translated
        (server.sampling_from_processors), `past_key_values` / `use_cache` are accepted and unused (the engine owns
        its caches).  Returns int64 (B, prompt + new) on the model's device, pad_token_id after each row's EOS."""
        from .server import sampling_from_processors
        audio = inputs if inputs is not None else frames
        if decoder_input_ids is None:
            raise ValueError("decoder_input_ids is required (the reference always passes the prompt)")
        enc_states = None
        if encoder_outputs is not None:      # B2: `generate(encoder_outputs=BaseModelOutput(...))` -- the encoder stage is skipped
            enc_states = getattr(encoder_outputs, "last_hidden_state", encoder_outputs)
            if isinstance(enc_states, (tuple, list)):
                enc_states = enc_states[0]
            if num_beams != 1:
                raise NotImplementedError("generate(encoder_outputs=...) with beams: pass the audio")
        elif audio is None:
            raise ValueError("either inputs / frames (raw audio) or encoder_outputs is required")
        cfgm = self.config
        max_length = int(max_length or cfgm.max_target_positions)
        pad = cfgm.pad_token_id if pad_token_id is None else pad_token_id
        sp = sampling_from_processors(logits_processor, cfgm.vocab_size, do_sample=do_sample, top_k=top_k, top_p=top_p,
                                      max_length=max_length, pad_token_id=pad, seed=seed)
        eos = eos_token_id if eos_token_id is not None else [cfgm.eos_token_id]
        eos = [eos] if isinstance(eos, int) else list(eos)
        if sp.cfg_scale > 1.0 and negative_prompt is None:
            raise ValueError("guidance needs negative_prompt (modeling_mapperatorinator.py:243-254)")
        row_bias = None if enc_states is not None else self._row_bias(decoder_input_ids.shape[0], unused)   # (the conditioning acts in the encoder)
        if num_beams != 1:
            if unused.get("cross_kv_fp8"):
                raise NotImplementedError("cross_kv_fp8 with beam search: the step-wise beam entry streams the bf16 cross K / V")
            out = self.engine.generate_beam(audio, decoder_input_ids, decoder_attention_mask, eos, sp, int(num_beams),
                                            negative_prompt=negative_prompt if sp.cfg_scale > 1.0 else None,
                                            sample_fn=unused.get("beam_sample_fn"),
                                            **({} if row_bias is None else dict(row_bias=row_bias)))
            return out["tokens"].to(self.device)
        out = self.engine.generate(audio, decoder_input_ids, decoder_attention_mask, eos, sp,
                                   negative_prompt=negative_prompt if sp.cfg_scale > 1.0 else None,
                                   negative_mask=negative_prompt_attention_mask, cross_kv_fp8=bool(unused.get("cross_kv_fp8", False)),
                                   encoder_states=enc_states,
                                   **({} if row_bias is None else dict(row_bias=row_bias)))
        return out["tokens"].to(self.device)
