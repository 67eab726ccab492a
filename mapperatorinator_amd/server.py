"""Drop-in for the reference's decode boundary B1:
`model_generate(model, tokenizer, model_kwargs, generate_kwargs) -> (LongTensor[B, P+new] on CPU, stats)`
(osuT5/osuT5/inference/server.py:83-156), with `model` a `MapperatorinatorHIP`.

Same kwargs, same EOS-set construction (server.py:72-80), same processor order
(MonotonicTimeShift -> TimeshiftBias -> Temperature -> LookbackBias, server.py:106-134), same
stats dict (server.py:50-69).  Options the HIP path does not implement raise NotImplementedError
(CFG batch doubling, beam search, types_first conditional temperature / lookback renormalisation:
SURVEY.md 8f rank 3) -- never a silent approximation.
"""
from __future__ import annotations

import time

import torch

from . import _lib
from .event import ContextType, EventType

MILISECONDS_PER_SECOND = 1000
MILISECONDS_PER_STEP = 10


def get_eos_token_id(tokenizer, lookback_time: float = 0, lookahead_time: float = 0, context_type=None):
    """EOS set: eos, the context's eos, leading TIME_SHIFT ids inside the lookback window and trailing
    ones inside the lookahead window (reference server.py:72-80)."""
    ts0, ts1 = _ev(tokenizer.event_start, "TIME_SHIFT"), _ev(tokenizer.event_end, "TIME_SHIFT")
    ids = [tokenizer.eos_id]
    ceos = getattr(tokenizer, "context_eos", {}) or {}
    if context_type is not None:
        for k, v in ceos.items():
            if getattr(k, "value", k) == getattr(context_type, "value", context_type):
                ids.append(v)
    if lookback_time > 0:
        ids.extend(range(ts0, ts0 + int(lookback_time / MILISECONDS_PER_STEP)))
    if lookahead_time > 0:
        ids.extend(range(ts1 - int(lookahead_time / MILISECONDS_PER_STEP), ts1))
    return ids


def _ev(table: dict, name: str):
    """Look an EventType up by NAME so that the reference's own enum keys work too."""
    for k, v in table.items():
        if getattr(k, "name", None) == name:
            return v
    raise KeyError(name)


def _prompt_token_counts(model_kwargs, pad_token_id):
    mask = model_kwargs.get("decoder_attention_mask")
    if isinstance(mask, torch.Tensor):
        return mask.to(torch.long).sum(dim=-1).cpu()
    ids = model_kwargs.get("decoder_input_ids")
    if not isinstance(ids, torch.Tensor):
        return None
    if pad_token_id is None:
        return torch.full((ids.shape[0],), ids.shape[1], dtype=torch.long)
    return ids.ne(pad_token_id).to(torch.long).sum(dim=-1).cpu()


def _build_generation_stats(result, model_kwargs, pad_token_id, elapsed_seconds):
    """generated (non-pad, post-prompt) tokens / wall time -- the reference's own metric (server.py:50-69)."""
    prompt_counts = _prompt_token_counts(model_kwargs, pad_token_id)
    out_counts = (result.ne(pad_token_id).to(torch.long).sum(dim=-1) if pad_token_id is not None
                  else torch.full((result.shape[0],), result.shape[1], dtype=torch.long))
    if prompt_counts is not None:
        out_counts = torch.clamp(out_counts - prompt_counts, min=0)
    total = int(out_counts.sum().item())
    return {"generated_tokens": total, "generated_tokens_per_sample": out_counts.tolist(),
            "elapsed_seconds": float(elapsed_seconds),
            "tokens_per_second": total / elapsed_seconds if elapsed_seconds > 0 else 0.0}


def build_sampling(tokenizer, generate_kwargs: dict, max_target_positions: int):
    """Translate the reference's generate kwargs (processor.py:156-170,358-360) into MhSampling + EOS ids."""
    gk = dict(generate_kwargs)
    gk.pop("precision", None)
    cfg_scale = gk.pop("cfg_scale", 1.0)
    timeshift_bias = gk.pop("timeshift_bias", 0)
    types_first = gk.pop("types_first", False)
    temperature = gk.pop("temperature", 1.0)
    timing_t = gk.pop("timing_temperature", temperature)
    mania_t = gk.pop("mania_column_temperature", temperature)
    taiko_t = gk.pop("taiko_hit_temperature", temperature)
    lookback_time = gk.pop("lookback_time", 0.0)
    lookahead_time = gk.pop("lookahead_time", 0.0)
    context_type = gk.pop("context_type", None)
    if context_type is not None:
        context_type = ContextType(getattr(context_type, "value", context_type))
    if cfg_scale > 1.0:
        raise NotImplementedError("classifier-free guidance (cfg_scale > 1) is not on the HIP path yet")
    if gk.get("num_beams", 1) != 1:
        raise NotImplementedError("beam search is not on the HIP path (num_beams must be 1)")
    if types_first and (timing_t != temperature or mania_t != temperature or taiko_t != temperature):
        raise NotImplementedError("ConditionalTemperatureLogitsWarper (types_first) is not on the HIP path")
    if types_first and lookback_time > 0:
        raise NotImplementedError("LookbackBiasLogitsWarper with types_first=True is not on the HIP path")

    ts0, ts1 = _ev(tokenizer.event_start, "TIME_SHIFT"), _ev(tokenizer.event_end, "TIME_SHIFT")
    sp = _lib.MhSampling()
    sp.do_sample = int(bool(gk.get("do_sample", False)))
    sp.top_k = int(gk.get("top_k", 0) or 0)
    sp.top_p = float(gk.get("top_p", 1.0))
    sp.temperature = float(temperature)
    sp.timeshift_bias = float(timeshift_bias)
    sp.ts_start, sp.ts_end = int(ts0), int(ts1)
    sos = [tokenizer.sos_id] + list((getattr(tokenizer, "context_sos", {}) or {}).values())
    if len(sos) > 16:
        raise NotImplementedError("more than 16 SOS-type ids")
    sp.n_sos = len(sos)
    for i, v in enumerate(sos):
        sp.sos_ids[i] = int(v)
    # LookbackBiasLogitsWarper, types_first=False: ids [ts_start, encode(TIME_SHIFT, lookback/10)) -> -inf
    # (logit_processors.py:93-96,111-114)
    sp.lookback_mask_end = 0
    if lookback_time > 0:
        er = None
        for k, v in tokenizer.event_range.items():
            if getattr(k, "name", None) == "TIME_SHIFT":
                er = v
        sp.lookback_mask_end = int(ts0 + int(lookback_time / MILISECONDS_PER_STEP) - er.min_value)
    sp.pad_id = int(gk.get("pad_token_id", getattr(tokenizer, "pad_id", 0)) or 0)
    sp.max_length = int(gk.get("max_length", max_target_positions))
    sp.seed = int(gk.get("seed", torch.initial_seed())) & 0xFFFFFFFFFFFFFFFF
    eos = get_eos_token_id(tokenizer, lookback_time=lookback_time, lookahead_time=lookahead_time,
                           context_type=context_type)
    return sp, eos


@torch.no_grad()
def model_generate(model, tokenizer, model_kwargs, generate_kwargs):
    """See module docstring.  `model_kwargs['inputs']`: raw audio float32 (B, Ns)."""
    generate_kwargs = dict(generate_kwargs)
    for k in ("negative_prompt", "negative_prompt_attention_mask"):
        if model_kwargs.get(k) is not None:
            raise NotImplementedError("negative prompts need CFG, which is not on the HIP path yet")
    for k in ("beatmap_idx", "difficulty", "mapper_idx", "song_position"):
        if model_kwargs.get(k) is not None:
            raise NotImplementedError(f"conditioning input {k!r} is not part of the T5 north-star configs")
    audio = model_kwargs["inputs"]
    prompt = model_kwargs["decoder_input_ids"]
    mask = model_kwargs.get("decoder_attention_mask")
    sp, eos = build_sampling(tokenizer, generate_kwargs, model.config.max_target_positions)
    if sp.max_length > model.config.max_target_positions:
        raise ValueError(f"max_length {sp.max_length} exceeds max_target_positions "
                         f"{model.config.max_target_positions}")
    pad_token_id = generate_kwargs.get("pad_token_id", getattr(tokenizer, "pad_id", None))

    start = time.perf_counter()
    out = model.engine.generate(audio, prompt, mask, eos, sp)
    elapsed = time.perf_counter() - start
    result = out["tokens"]
    stats = _build_generation_stats(result, model_kwargs, pad_token_id, elapsed)
    return result, stats
