"""TEST INFRASTRUCTURE ONLY -- drives the *imported, unmodified* reference (this container only).

Builds the reference objects for the hot path exactly the way SURVEY.md Appendix C describes
and exposes them to `oracle/make_golden.py` and the pinning tests:

  build_reference_t5(size, ...)  -> (model: Mapperatorinator, tokenizer: Tokenizer, args)
  reference_generate(...)        -> reference `model_generate` (osuT5/osuT5/inference/server.py:83-156)
  build_reference_dit(...)       -> osu_diffusion DiT (osu_diffusion/utils/models.py)
  reference_ddpm(...)            -> diffusion.p_sample_loop (gaussian_diffusion.py:469-512)
"""
from __future__ import annotations

import copy

import numpy as np
import torch

from . import ref_shims


def _train_config(size: str, src_seq_len: int, tgt_seq_len: int, n_mels: int):
    ref_shims.install()
    from osuT5.osuT5.config import TrainConfig
    from osuT5.osuT5.event import ContextType
    args = TrainConfig()
    args = copy.deepcopy(args)
    args.model.name = f"google/t5-v1_1-{size}"
    args.model.input_features = False
    args.model.project_encoder_input = True
    args.model.embed_decoder_input = True
    args.model.do_style_embed = False
    args.model.do_difficulty_embed = False
    args.model.do_mapper_embed = False
    args.model.do_song_position_embed = False
    args.model.cond_size = 0
    args.model.overwrite = {"dropout_rate": 0.0}
    args.model.add_config = {}
    args.model.spectrogram.implementation = "nnAudio"
    args.model.spectrogram.log_scale = False
    args.model.spectrogram.sample_rate = 16000
    args.model.spectrogram.n_fft = 1024
    args.model.spectrogram.n_mels = n_mels
    args.model.spectrogram.hop_length = 128
    args.model.spectrogram.f_min = 0
    args.model.spectrogram.f_max = 8000
    args.model.spectrogram.pad_mode = "constant"
    d = args.data
    d.dataset_type = "ors"
    d.src_seq_len = src_seq_len
    d.tgt_seq_len = tgt_seq_len
    d.context_types = [{"in": [], "out": [ContextType.MAP]}]
    d.add_out_context_types = False
    for name in dir(d):
        if name.startswith("add_") and name.endswith("_token") and isinstance(getattr(d, name), bool):
            setattr(d, name, False)
    d.add_descriptors = False
    d.add_kiai = False
    d.add_kiai_special_token = False
    d.types_first = False
    d.gamemodes = [0]
    d.add_sv = False
    d.add_positions = False
    d.add_distances = True
    d.add_timing_points = False
    d.add_pre_tokens = False
    d.add_pre_tokens_at_step = -1
    d.sustain_interval = 0
    return args


TINY_OVERWRITE = {"d_model": 128, "d_ff": 256, "num_heads": 2, "num_layers": 2, "num_decoder_layers": 2}


def build_reference_t5(size="small", src_seq_len=1251, tgt_seq_len=512, n_mels=388,
                       dtype=torch.float32, seed=0, lm_head_gain=1.0, overwrite=None, types_first=False, cond=None):
    """reference `_get_model` (osuT5/osuT5/utils/model_utils.py:102-114) with the reference
    initialisers (HF `_init_weights`), seeded."""
    if size == "tiny":  # test-only size: t5-small backbone config with the dims overwritten
        size, overwrite = "small", dict(TINY_OVERWRITE, **(overwrite or {}))
    args = _train_config(size, src_seq_len, tgt_seq_len, n_mels)
    if types_first:
        args = types_first_train_config(src_seq_len, tgt_seq_len, n_mels)
        args.model.name = f"google/t5-v1_1-{size}"
    if overwrite:
        args.model.overwrite = dict(args.model.overwrite, **overwrite)
    from osuT5.osuT5.tokenizer import Tokenizer
    from osuT5.osuT5.utils.model_utils import _get_model
    tok = Tokenizer(args)   # (built before the conditioning switches: with do_mapper_embed the tokenizer wants a mapper table file)
    if cond:   # difficulty + mapper + song-position embedders (modeling_mapperatorinator.py:104-128), cond_dim each
        args.model.do_difficulty_embed = args.model.do_mapper_embed = args.model.do_song_position_embed = True
        args.model.cond_dim = cond["cond_dim"]
        args.model.cond_size = 3 * cond["cond_dim"]
        tok.num_mapper_classes = cond["num_mappers"]     # what the mapper table would set (tokenizer.py:586); model_utils.py:56 reads it
    torch.manual_seed(seed)
    model = _get_model(args, tok, torch.float32, "eager").eval()
    if lm_head_gain != 1.0:
        with torch.no_grad():
            model.transformer.lm_head.weight.mul_(lm_head_gain)
    if dtype != torch.float32:
        model = model.to(dtype)
    return model, tok, args


def ts_range(tok):
    """(first, one-past-last) TIME_SHIFT id of a reference or mirror tokenizer"""
    s = [v for k, v in tok.event_start.items() if k.name == "TIME_SHIFT"][0]
    e = [v for k, v in tok.event_end.items() if k.name == "TIME_SHIFT"][0]
    return s, e


def reference_cond_vectors(model, difficulty=None, mapper_idx=None, song_position=None):
    """The per-row conditioning vectors exactly as `Mapperatorinator.forward` builds them from its own embedder modules
    (modeling_mapperatorinator.py:395-409): (B, cond_size), order difficulty | mapper | song position."""
    with torch.no_grad():
        conds = []
        if model.do_difficulty_embed:
            conds.append(model.difficulty_embedder(difficulty))
        if model.do_mapper_embed:
            conds.append(model.mapper_embedder(mapper_idx))
        if model.do_song_position_embed:
            conds.append(model.song_pos_embedder(song_position))
        return torch.cat(conds, -1) if conds else None


def reference_encode(model, audio: torch.Tensor, cond: torch.Tensor = None) -> torch.Tensor:
    """mel (| conditioning vectors repeated over the frames) -> encoder_embedder -> T5 encoder, called by hand
    (work-around for the positional `inputs_embeds` bug at modeling_mapperatorinator.py:438-443; SURVEY.md headline
    finding 3); the concatenation is :411-412."""
    with torch.no_grad():
        mel = model.spectrogram(audio).to(model.transformer.dtype)
        if cond is not None:
            mel = torch.cat([mel, cond.to(mel.dtype).unsqueeze(1).expand(-1, mel.shape[1], -1)], -1)
        emb = model.encoder_embedder(mel)
        return model.transformer.encoder(inputs_embeds=emb).last_hidden_state


def default_generate_kwargs(max_length: int, **over):
    kw = dict(precision="fp32", do_sample=False, num_beams=1, top_p=1.0, top_k=0,
              max_length=max_length, cfg_scale=1.0, timeshift_bias=0, types_first=False,
              temperature=1.0, timing_temperature=1.0, mania_column_temperature=1.0,
              taiko_hit_temperature=1.0, lookback_time=0, lookahead_time=0,
              context_type="map", pad_token_id=0)
    kw.update(over)
    return kw


def reference_generate(model, tok, audio, prompt, generate_kwargs, attention_mask=None, negative_prompt=None,
                       negative_mask=None, record_scores=None, cond=None):
    """The reference's own `model_generate` (server.py:83-156) via the `encoder_outputs` route.
    `record_scores`: a list that receives the processed scores of every step (what the merged
    LogitsProcessorList returns inside HF `_sample`), observed without touching reference code."""
    ref_shims.install()
    from osuT5.osuT5.inference.server import model_generate
    from transformers import LogitsProcessorList
    from transformers.modeling_outputs import BaseModelOutput
    enc = reference_encode(model, audio, cond)
    if attention_mask is None:
        attention_mask = prompt.ne(0)
    mk = dict(inputs=audio, encoder_outputs=BaseModelOutput(last_hidden_state=enc),
              decoder_input_ids=prompt, decoder_attention_mask=attention_mask)
    if negative_prompt is not None:
        mk.update(negative_prompt=negative_prompt,
                  negative_prompt_attention_mask=negative_prompt.ne(0) if negative_mask is None else negative_mask)
    orig = LogitsProcessorList.__call__
    if record_scores is not None:
        def spy(self, input_ids, scores, **kw):
            out = orig(self, input_ids, scores, **kw)
            record_scores.append(out.detach().float().cpu().clone())
            return out
        LogitsProcessorList.__call__ = spy
    try:
        return model_generate(model, tok, mk, dict(generate_kwargs))
    finally:
        LogitsProcessorList.__call__ = orig


def types_first_train_config(src_seq_len: int, tgt_seq_len: int, n_mels: int = 388):
    """A tokenizer configuration that carries every token family the types_first processors look at: beat types,
    mania types, scroll speeds, context sos/eos (all gamemodes, SVs, timing points, kiai)."""
    args = _train_config("small", src_seq_len, tgt_seq_len, n_mels)
    d = args.data
    d.types_first = True
    d.gamemodes = [0, 1, 2, 3]
    d.add_sv = True
    d.add_timing_points = True
    d.add_kiai = True
    d.add_out_context_types = True
    return args


def build_reference_dit(name="DiT-S", context_size=272, class_size=300, seed=0, rerandomise=True):
    """osu_diffusion DiT (models.py:213-279, sizes :384-405).  adaLN / final layers are zero-init
    in the reference (:270-279) which makes the output identically 0; for a meaningful parity
    check they are re-randomised N(0, 0.02) (SURVEY.md 8d 'Value distributions')."""
    ref_shims.install()
    from osu_diffusion import DiT_models
    torch.manual_seed(seed)
    model = DiT_models[name](context_size=context_size, class_size=class_size).eval()
    if rerandomise:
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for blk in model.blocks:
                blk.adaLN_modulation[-1].weight.normal_(0, 0.02, generator=g)
                blk.adaLN_modulation[-1].bias.normal_(0, 0.02, generator=g)
            model.final_layer.adaLN_modulation[-1].weight.normal_(0, 0.02, generator=g)
            model.final_layer.adaLN_modulation[-1].bias.normal_(0, 0.02, generator=g)
            model.final_layer.linear.weight.normal_(0, 0.02, generator=g)
            model.final_layer.linear.bias.normal_(0, 0.02, generator=g)
            for m in model.modules():
                if isinstance(m, torch.nn.Linear) and m.bias is not None and float(m.bias.abs().sum()) == 0.0:
                    m.bias.normal_(0, 0.02, generator=g)
            for blk in model.blocks:
                blk.attn.in_proj_bias.normal_(0, 0.02, generator=g)
                blk.attn.out_proj.bias.normal_(0, 0.02, generator=g)
    return model


def reference_diffusion(timesteps=(100, 0, 0, 0, 0, 0, 0, 0, 0, 0), diffusion_steps=1000,
                        noise_schedule="squaredcos_cap_v2"):
    ref_shims.install()
    from osu_diffusion import create_diffusion
    return create_diffusion(timestep_respacing=list(timesteps), diffusion_steps=diffusion_steps,
                            noise_schedule=noise_schedule)


def reference_ddpm(model, diffusion, z, c, y, cfg_scale, attn_mask, noise_list):
    """`diffusion.p_sample_loop(model.forward_with_cfg, ...)` (diffusion_pipeline.py:243-252) with the
    per-step gaussian noise injected (pops from `noise_list`, one tensor per step, in call order)."""
    from osu_diffusion.utils.diffusion import gaussian_diffusion as gd
    it = iter(noise_list)
    orig = gd.th.randn_like
    gd.th.randn_like = lambda x, *a, **k: next(it).to(x)
    try:
        with torch.no_grad():
            return diffusion.p_sample_loop(
                model.forward_with_cfg, z.shape, z, clip_denoised=True,
                model_kwargs=dict(c=c, y=y, cfg_scale=cfg_scale, attn_mask=attn_mask,
                                  key_padding_mask=None), device=z.device)
    finally:
        gd.th.randn_like = orig


def reference_pipeline_positions(model, seq_x, seq_o, seq_c, class_vector, unk_class_vector, noise_list, *, timesteps,
                                 seq_len, max_seq_len, overlap_buffer, cfg_scale, refine_iters=0, start_time=None,
                                 end_time=None, diffusion_steps=1000, noise_schedule="squaredcos_cap_v2", sliders=(),
                                 pad_sequence=False):
    """The reference's own `DiffisionPipeline.generate` (diffusion_pipeline.py:111-287) driven from the tensors that
    `events_to_sequence` returns: the object is built without its constructor, `events_to_sequence`,
    `get_class_vector` and `events_with_pos` are replaced by stand-ins that hand the given tensors through (Event
    grouping needs the `slider` package, absent here), everything in between is unmodified reference code.  The
    gaussian draws are popped from `noise_list` in call order (one per p_sample call, refine steps included).
    `sliders`: DiffusionSlider-like objects handed through as the 6th value of `events_to_sequence` (their end points are
    then re-projected by the reference's own `denoised_fn` / SliderPath code)."""
    ref_shims.install()
    import diffusion_pipeline as dp
    from osu_diffusion.utils.diffusion import gaussian_diffusion as gd
    pipe = object.__new__(dp.DiffisionPipeline)
    pipe.device = "cpu"
    pipe.model = model
    pipe.tokenizer = None
    pipe.refine_model = model if refine_iters > 0 else None
    pipe.diffusion_steps, pipe.noise_schedule = diffusion_steps, noise_schedule
    pipe.seq_len, pipe.max_seq_len, pipe.overlap_buffer = seq_len, max_seq_len, overlap_buffer
    pipe.timesteps, pipe.cfg_scale, pipe.refine_iters = list(timesteps), cfg_scale, refine_iters
    pipe.random_init, pipe.types_first, pipe.pad_sequence = False, False, bool(pad_sequence)
    pipe.start_time, pipe.end_time = start_time, end_time
    calls = []

    def get_class_vector(config):
        calls.append(1)
        return (class_vector if len(calls) == 1 else unk_class_vector).clone()

    ref_sliders = [dp.DiffusionSlider(np.asarray(s.seq_indices), int(s.end_index), s.curve_type, float(s.length))
                   for s in sliders]
    pipe.events_to_sequence = lambda events, timing, sm: (seq_x, seq_o, seq_c, seq_x.shape[1], {}, ref_sliders)
    pipe.get_class_vector = get_class_vector
    pipe.events_with_pos = lambda events, positions, seq_indices: positions

    class _Cfg:  # the fields `generate` reads (diffusion_pipeline.py:151-155)
        slider_multiplier, difficulty, negative_descriptors, circle_size = 1.4, None, None, None

    it = iter(noise_list)
    orig = gd.th.randn_like
    gd.th.randn_like = lambda x, *a, **k: next(it).to(x)
    try:
        with torch.no_grad():
            return pipe.generate([], _Cfg(), None)
    finally:
        gd.th.randn_like = orig


def reference_pipeline_generate(model, events, generation_config, timing, tokenizer_state, noise_list, *, timesteps,
                                seq_len, max_seq_len, overlap_buffer, cfg_scale, refine_iters=0, start_time=None,
                                end_time=None, diffusion_steps=1000, noise_schedule="squaredcos_cap_v2",
                                types_first=False, has_sv=True, pad_sequence=False):
    """The reference's `DiffisionPipeline.generate` with NOTHING replaced: its own `events_to_sequence` (Event grouping,
    slider list), `get_class_vector` over its own Tokenizer loaded from `tokenizer_state`, the window loop with
    `denoised_fn`, and `events_with_pos`.  `events`: objects with `.type.name` / `.value` (translated to the reference's
    Event class); returns the reference's events as (type name, value) pairs.  Gaussian draws from `noise_list`."""
    ref_shims.install()
    import diffusion_pipeline as dp
    from osu_diffusion.utils.diffusion import gaussian_diffusion as gd
    from osu_diffusion.utils.tokenizer import Tokenizer
    from osuT5.osuT5.tokenizer import Event, EventType
    tok = Tokenizer()
    tok.load_state_dict(tokenizer_state)
    pipe = object.__new__(dp.DiffisionPipeline)
    pipe.device = "cpu"
    pipe.model, pipe.tokenizer = model, tok
    pipe.refine_model = model if refine_iters > 0 else None
    pipe.diffusion_steps, pipe.noise_schedule = diffusion_steps, noise_schedule
    pipe.seq_len, pipe.max_seq_len, pipe.overlap_buffer = seq_len, max_seq_len, overlap_buffer
    pipe.timesteps, pipe.cfg_scale, pipe.refine_iters = list(timesteps), cfg_scale, refine_iters
    pipe.random_init, pipe.types_first, pipe.pad_sequence = False, bool(types_first), bool(pad_sequence)
    pipe.start_time, pipe.end_time, pipe.has_sv = start_time, end_time, bool(has_sv)
    it = iter(noise_list)
    orig = gd.th.randn_like
    gd.th.randn_like = lambda x, *a, **k: next(it).to(x)
    try:
        with torch.no_grad():
            out = pipe.generate([Event(EventType[e.type.name], e.value) for e in events], generation_config, timing)
    finally:
        gd.th.randn_like = orig
    return [(e.type.name, int(e.value)) for e in out]


def make_reference_processor(tok, model, *, src_seq_len: int, tgt_seq_len: int, lookback: float = 0.5, lookahead: float = 0.4,
                             train_lookahead: float = 0.0, cfg_scale: float = 1.0, types_first: bool = False,
                             add_pre_tokens: bool = False, hop_length: int = 128, sample_rate: int = 16000):
    """The reference's `Processor` (osuT5/osuT5/inference/processor.py:71-150) without its constructor -- that one wants a
    whole InferenceConfig and an OsuParser (the absent `slider` package) -- with every attribute the sequential /
    parallel generation loops read, computed by the constructor's own formulas."""
    ref_shims.install()
    from osuT5.osuT5.event import Event, EventType
    from osuT5.osuT5.inference import processor as ref_proc
    proc = object.__new__(ref_proc.Processor)
    ms_step = ref_proc.MILISECONDS_PER_STEP
    proc.model, proc.tokenizer, proc.precision, proc.device = model, tok, "fp32", "cpu"
    proc.tgt_seq_len = tgt_seq_len
    proc.frame_seq_len = src_seq_len - 1
    proc.frame_size, proc.sample_rate = hop_length, sample_rate
    proc.samples_per_sequence = proc.frame_seq_len * proc.frame_size
    proc.miliseconds_per_sequence = proc.samples_per_sequence * ref_proc.MILISECONDS_PER_SECOND / proc.sample_rate
    proc.lookback_time = lookback * proc.miliseconds_per_sequence
    proc.lookback_time_range = range(tok.event_start[EventType.TIME_SHIFT],
                                     tok.encode(Event(EventType.TIME_SHIFT, int(proc.lookback_time / ms_step))))
    proc.lookahead_max_time = (1 - lookahead) * proc.miliseconds_per_sequence
    proc.lookahead_time = lookahead * proc.miliseconds_per_sequence
    proc.lookahead_time_range = range(tok.encode(Event(EventType.TIME_SHIFT, int(proc.lookahead_max_time / ms_step))),
                                      tok.event_end[EventType.TIME_SHIFT])
    proc.eos_time = (1 - train_lookahead) * proc.miliseconds_per_sequence
    proc.center_pad_decoder = False
    for name in ("add_out_context_types", "add_gamemode_token", "add_style_token", "add_diff_token", "add_mapper_token",
                 "add_year_token", "add_hitsounded_token", "add_song_length_token", "add_global_sv_token", "add_cs_token",
                 "add_keycount_token", "add_hold_note_ratio_token", "add_scroll_speed_ratio_token", "add_descriptors",
                 "add_sv_special_token", "add_kiai_special_token", "add_song_position_token", "add_kiai", "add_gd_context",
                 "add_timing", "do_style_embed", "do_difficulty_embed", "do_mapper_embed", "do_song_position_embed",
                 "add_positions", "add_sv", "add_mania_sv", "add_to_beatmap"):
        setattr(proc, name, False)
    proc.max_pre_token_len, proc.add_pre_tokens = -1, add_pre_tokens
    proc.start_time = proc.end_time = None
    proc.cfg_scale, proc.top_p, proc.top_k, proc.temperature = cfg_scale, 0.9, 0, 0.9
    proc.timing_temperature, proc.mania_column_temperature, proc.taiko_hit_temperature = 0.9, 0.9, 0.9
    proc.do_sample, proc.num_beams, proc.parallel, proc.max_batch_size = False, 1, False, 4
    proc.timeshift_bias, proc.types_first, proc.last_generation_stats = 0.0, types_first, None
    return proc


# ---- the Whisper-family backbone the released V30-V32 checkpoints use ---------------------------------
VARWHISPER_TINY_OVERWRITE = {"d_model": 128, "encoder_layers": 2, "decoder_layers": 2, "encoder_attention_heads": 2,
                             "decoder_attention_heads": 2, "encoder_ffn_dim": 256, "decoder_ffn_dim": 256}


def build_reference_varwhisper(size="small", src_seq_len=1024, tgt_seq_len=256, n_mels=128, seed=0, head_gain=1.0,
                               overwrite=None, attn_implementation="sdpa", attention_bias=True, global_attn_every_n_layers=1,
                               local_attention=128):
    """reference `_get_model` on configs/model/varwhisper_{small,base}_v3.yaml: backbone 'OliBomby/varwhisper-<size>'
    (custom_transformers/modeling_varwhisper.py), input_features = true, project_encoder_input = false, the torchaudio
    log-mel front-end (128 mels, f_min 20, reflect padding, log1p), untied head.  size "tiny": test-only dims."""
    ref_shims.varwhisper_module()
    if size == "tiny":
        size, overwrite = "tiny", dict(VARWHISPER_TINY_OVERWRITE, **(overwrite or {}))
    args = _train_config("small", src_seq_len, tgt_seq_len, n_mels)
    args.model.name = f"OliBomby/varwhisper-{size}"
    args.model.input_features = True
    args.model.project_encoder_input = False
    args.model.overwrite = dict({"tie_word_embeddings": False}, **(overwrite or {}))
    args.model.attention_bias = attention_bias                        # configs/model/default.yaml:23 (true in the released configs)
    args.model.global_attn_every_n_layers = global_attn_every_n_layers
    args.model.local_attention = local_attention
    sp = args.model.spectrogram
    sp.implementation, sp.log_scale, sp.n_mels, sp.f_min, sp.pad_mode = "torchaudio", True, n_mels, 20, "reflect"
    from osuT5.osuT5.tokenizer import Tokenizer
    from osuT5.osuT5.utils.model_utils import _get_model
    tok = Tokenizer(args)
    torch.manual_seed(seed)
    model = _get_model(args, tok, torch.float32, attn_implementation).eval()
    if head_gain != 1.0:
        with torch.no_grad():
            model.transformer.proj_out.weight.mul_(head_gain)
    return model, tok, args


def reference_encode_whisper(model, audio: torch.Tensor) -> torch.Tensor:
    """log-mel -> (B, n_mels, L) input_features -> the backbone's own encoder (conv front-end + layers + final norm): what
    `OsuTEncoder.forward` does for input_features = true (modeling_mapperatorinator.py:395-443), called by hand for the
    same reason as `reference_encode`."""
    with torch.no_grad():
        mel = model.spectrogram(audio).to(model.transformer.dtype)
        return model.transformer.get_encoder()(torch.swapaxes(mel, 1, 2)).last_hidden_state


def reference_generate_whisper(model, tok, audio, prompt, generate_kwargs, attention_mask=None, negative_prompt=None,
                               record_scores=None):
    """`reference_generate` for a Whisper-family backbone: the reference's own `model_generate` with the encoder states
    of `reference_encode_whisper` handed over as `encoder_outputs`."""
    ref_shims.varwhisper_module()
    from osuT5.osuT5.inference.server import model_generate
    from transformers import LogitsProcessorList
    from transformers.modeling_outputs import BaseModelOutput
    enc = reference_encode_whisper(model, audio)
    mk = dict(inputs=audio, encoder_outputs=BaseModelOutput(last_hidden_state=enc), decoder_input_ids=prompt,
              decoder_attention_mask=prompt.ne(0) if attention_mask is None else attention_mask)
    if negative_prompt is not None:
        mk.update(negative_prompt=negative_prompt, negative_prompt_attention_mask=negative_prompt.ne(0))
    orig = LogitsProcessorList.__call__
    if record_scores is not None:
        def spy(self, input_ids, scores, **kw):
            out = orig(self, input_ids, scores, **kw)
            record_scores.append(out.detach().float().cpu().clone())
            return out
        LogitsProcessorList.__call__ = spy
    try:
        return model_generate(model, tok, mk, dict(generate_kwargs))
    finally:
        LogitsProcessorList.__call__ = orig


# ---- the other two Whisper-family backbones: 'Tiger14n/ropewhisper-*' (V30 / V31) and 'openai/whisper-*' (V28 / V29) ---------
def build_reference_whisper_family(kind, src_seq_len=1024, tgt_seq_len=256, n_mels=None, seed=0, overwrite=None, cond=None,
                                   attn_implementation="eager"):
    """reference `_get_model` on configs/model/whisper_small_v2.yaml (kind "rope": 'Tiger14n/ropewhisper-*', input_features =
    true, project_encoder_input = false, torchaudio log-mel with 80 mels, the conditioning embedders as conv1 channels) or
    configs/model/whisper_{base,small}.yaml (kind "hf": 'openai/whisper-*', input_features = true, project_encoder_input = true
    -> the wrapper's encoder_embedder in front of conv1, the nnAudio mel of default.yaml).  `overwrite`: backbone dims (the
    presets of mapperatorinator_amd.whisper_engine); `cond`: dict(cond_dim, num_mappers) switches the difficulty / mapper /
    song-position embedders on (whisper_small_v2.yaml:9-13)."""
    assert kind in ("rope", "hf")
    (ref_shims.ropewhisper_module if kind == "rope" else ref_shims.hf_whisper_module)()
    if n_mels is None:
        n_mels = 80 if kind == "rope" else 388
    args = _train_config("small", src_seq_len, tgt_seq_len, n_mels)
    args.model.name = "Tiger14n/ropewhisper-small" if kind == "rope" else "openai/whisper-small"
    args.model.input_features = True
    args.model.project_encoder_input = kind == "hf"
    args.model.overwrite = dict({"tie_word_embeddings": False}, **(overwrite or {}))
    if kind == "rope":
        sp = args.model.spectrogram
        sp.implementation, sp.log_scale, sp.n_mels, sp.f_min, sp.pad_mode = "torchaudio", True, n_mels, 20, "reflect"
    from osuT5.osuT5.tokenizer import Tokenizer
    from osuT5.osuT5.utils.model_utils import _get_model
    tok = Tokenizer(args)
    if cond:
        args.model.do_difficulty_embed = args.model.do_mapper_embed = args.model.do_song_position_embed = True
        args.model.cond_dim = cond["cond_dim"]
        args.model.cond_size = 3 * cond["cond_dim"]
        tok.num_mapper_classes = cond["num_mappers"]
    torch.manual_seed(seed)
    model = _get_model(args, tok, torch.float32, attn_implementation).eval()
    return model, tok, args


def reference_encode_whisper_family(model, audio: torch.Tensor, cond: torch.Tensor = None) -> torch.Tensor:
    """`Mapperatorinator.forward`'s encoder half for input_features = true (modeling_mapperatorinator.py:191-213), called by
    hand like `reference_encode`: spectrogram -> (| conditioning vectors over the frames) -> (encoder_embedder when
    project_encoder_input) -> swapaxes -> the backbone's own encoder."""
    with torch.no_grad():
        x = model.spectrogram(audio).to(model.transformer.dtype)
        if cond is not None:
            x = torch.cat([x, cond.to(x.dtype).unsqueeze(1).expand(-1, x.shape[1], -1)], -1)
        if model.project_encoder_input:
            x = model.encoder_embedder(x)
        return model.transformer.get_encoder()(torch.swapaxes(x, 1, 2)).last_hidden_state


def reference_generate_whisper_family(model, tok, audio, prompt, generate_kwargs, attention_mask=None, negative_prompt=None,
                                      record_scores=None, cond=None, positions_from_mask=False):
    """the reference's own `model_generate` with the encoder states of `reference_encode_whisper_family` as `encoder_outputs`.
    `positions_from_mask` (stock HF Whisper only): hand the decoder the position ids transformers 4.57's Whisper
    `prepare_inputs_for_generation` derives itself -- decoder_position_ids = (decoder_attention_mask.cumsum(-1) - 1).clamp(min=0), the
    code the RoPEWhisper fork copied (modeling_ropewhisper.py:2015-2018) -- as an explicit model kwarg: the installed 5.x forwards it to
    the decoder and extends it per step by itself (generation/utils.py `_update_model_kwargs_for_generation`), so the run shows what
    the reference's PINNED transformers does with left-padded prompts.  (HF's check of kwarg names against the wrapper's forward
    signature is switched off for the call: the wrapper takes it through **kwargs.)"""
    from osuT5.osuT5.inference.server import model_generate
    from transformers import LogitsProcessorList
    from transformers.modeling_outputs import BaseModelOutput
    enc = reference_encode_whisper_family(model, audio, cond)
    mk = dict(inputs=audio, encoder_outputs=BaseModelOutput(last_hidden_state=enc), decoder_input_ids=prompt,
              decoder_attention_mask=prompt.ne(0) if attention_mask is None else attention_mask)
    if negative_prompt is not None:
        mk.update(negative_prompt=negative_prompt, negative_prompt_attention_mask=negative_prompt.ne(0))
    from transformers import GenerationMixin
    orig_validate = GenerationMixin._validate_model_kwargs
    if positions_from_mask:
        mk["decoder_position_ids"] = (mk["decoder_attention_mask"].long().cumsum(-1) - 1).clamp(min=0)
        GenerationMixin._validate_model_kwargs = lambda self, kw: None
    orig = LogitsProcessorList.__call__
    if record_scores is not None:
        def spy(self, input_ids, scores, **kw):
            out = orig(self, input_ids, scores, **kw)
            record_scores.append(out.detach().float().cpu().clone())
            return out
        LogitsProcessorList.__call__ = spy
    try:
        return model_generate(model, tok, mk, dict(generate_kwargs))
    finally:
        LogitsProcessorList.__call__ = orig
        GenerationMixin._validate_model_kwargs = orig_validate
