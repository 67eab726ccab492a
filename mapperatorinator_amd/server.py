"""Drop-in for the reference's decode boundary B1:
`model_generate(model, tokenizer, model_kwargs, generate_kwargs) -> (LongTensor[B, P+new] on CPU, stats)`
(osuT5/osuT5/inference/server.py:83-156), with `model` a `MapperatorinatorHIP`.

Same kwargs, same EOS-set construction (server.py:72-80), same processor order
(CFG -> MonotonicTimeShift -> TimeshiftBias -> Temperature | ConditionalTemperature -> LookbackBias,
server.py:106-134), same stats dict (server.py:50-69).  Classifier-free guidance follows the reference's
batch layout: the negative prompt rows first, the prompt rows second, one shared encoder output per pair
(modeling_mapperatorinator.py:243-254).  `num_beams > 1` runs HF's beam search over the step-wise decode entry
(mapperatorinator_amd/beam.py), with `do_sample` as HF's beam-sample (`generate_kwargs["beam_sample_fn"]` replaces the device's
`torch.multinomial` draw; what is not built raises NotImplementedError -- never a silent approximation).  `RequestBatcher` at the end of the file is the batching policy of the reference's
InferenceServer (server.py:343-424) in front of this function.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import _lib
from .event import ContextType

MILISECONDS_PER_SECOND = 1000
MILISECONDS_PER_STEP = 10

# events that carry a time (reference osuT5/osuT5/dataset/data_utils.py TIMED_EVENTS), by name
TIMED_EVENT_NAMES = ("CIRCLE", "SPINNER", "SPINNER_END", "SLIDER_HEAD", "LAST_ANCHOR", "SLIDER_END", "BEAT", "MEASURE",
                     "TIMING_POINT", "KIAI", "HOLD_NOTE", "HOLD_NOTE_END", "DRUMROLL", "DRUMROLL_END", "DENDEN",
                     "DENDEN_END", "SCROLL_SPEED_CHANGE")

FLAG_TIMED, FLAG_COND0, FLAG_LOOKBACK_EOS = 1, 2, 16   # bits of MhSampling.tok_flags (include/mapperhip.h)


class Sampling(_lib.MhSampling):
    """MhSampling + the host copy of its `tok_flags` table (the engine uploads it and fills the pointer)."""
    host_tok_flags = None


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


def _has(table: dict, name: str) -> bool:
    return any(getattr(k, "name", None) == name for k in table)


def get_beat_type_tokens(tokenizer):
    """(reference logit_processors.py:14-21)"""
    ids = [_ev(tokenizer.event_start, "BEAT"), _ev(tokenizer.event_start, "MEASURE")]
    if _has(tokenizer.event_start, "TIMING_POINT"):
        ids.append(_ev(tokenizer.event_start, "TIMING_POINT"))
    return tuple(ids)


def get_mania_type_tokens(tokenizer):
    """(reference logit_processors.py:24-29)"""
    if not _has(tokenizer.event_start, "HOLD_NOTE_END"):
        return ()
    return tuple(_ev(tokenizer.event_start, n) for n in ("CIRCLE", "HOLD_NOTE", "HOLD_NOTE_END"))


def get_scroll_speed_tokens(tokenizer):
    """(reference logit_processors.py:32-34)"""
    if not _has(tokenizer.event_start, "SCROLL_SPEED"):
        return ()
    return tuple(range(_ev(tokenizer.event_start, "SCROLL_SPEED"), _ev(tokenizer.event_end, "SCROLL_SPEED")))


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


_SEED_CALLS: dict = {}
_SEED_CALLS_MAX = 4096                    # distinct explicit seeds remembered (least recently used ones are forgotten)
_SEED_LOCK = __import__("threading").Lock()


def fresh_seed(seed=None, call_index=None) -> int:
    """Seed of one generate call.  The reference samples with `torch.multinomial`, which ADVANCES the global generator
    on every draw, so two calls never replay the same uniforms; the in-kernel RNG is keyed by (seed, row, column), so
    the call's seed must move instead.  Without an explicit seed: one 63-bit draw from torch's default generator per call
    (still a deterministic function of `torch.manual_seed`).  With an explicit `seed` (a caller that wants reproducible
    runs): the n-th call made with that seed gets splitmix64(seed, n) -- the windows and waves of a song, or consecutive
    `model_generate` calls, draw from different streams, and the same sequence of calls reproduces the same tokens
    (`reset_seed_calls()` restarts the count).  That count is process-wide (guarded by a lock, bounded); a caller whose
    draws must not depend on what else ran in the process -- other threads, a batcher splitting requests differently, the
    ranks of a sharded job -- passes its OWN count as `call_index` (generate kwarg `seed_call_index`)."""
    if seed is None:
        return int(torch.randint(0, 2 ** 63 - 1, (1,), dtype=torch.int64).item())
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    if call_index is not None:
        # the caller owns the count (a scheduler / batcher / sharded rank that wants its draws independent of what else
        # ran in the process): stream (seed, call_index), nothing global is read or advanced
        n = int(call_index)
    else:
        with _SEED_LOCK:
            n = _SEED_CALLS.pop(seed, 0)
            _SEED_CALLS[seed] = n + 1                 # (re-inserted last: the dict stays in least-recently-used order)
            while len(_SEED_CALLS) > _SEED_CALLS_MAX:
                _SEED_CALLS.pop(next(iter(_SEED_CALLS)))
    z = (seed + 0x9E3779B97F4A7C15 * (n + 1)) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def reset_seed_calls() -> None:
    """Forget how many calls were made with each explicit seed (see `fresh_seed`)."""
    with _SEED_LOCK:
        _SEED_CALLS.clear()


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
    num_beams = int(gk.get("num_beams", 1) or 1)

    ts0, ts1 = _ev(tokenizer.event_start, "TIME_SHIFT"), _ev(tokenizer.event_end, "TIME_SHIFT")
    sp = Sampling()
    sp.cfg_scale = float(cfg_scale) if cfg_scale > 1.0 else 1.0        # server.py:107 `if cfg_scale > 1.0`
    flags = np.zeros(int(tokenizer.vocab_size_out), dtype=np.uint8)
    need_flags = False
    if types_first:
        # ConditionalTemperatureLogitsWarper (logit_processors.py:59-73): (temperature, token set, offset) rules
        rules = []
        for t, ids, off in ((timing_t, get_beat_type_tokens(tokenizer), 1),
                            (mania_t, get_mania_type_tokens(tokenizer), 3),
                            (taiko_t, get_scroll_speed_tokens(tokenizer), 1)):
            if t != temperature and len(ids) > 0:
                rules.append((t, ids, off))
        sp.n_cond = len(rules)
        for j, (t, ids, off) in enumerate(rules):
            sp.cond_temp[j], sp.cond_offset[j] = float(t), int(off)
            flags[np.asarray(ids, dtype=np.int64)] |= FLAG_COND0 << j
            need_flags = True
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
        if types_first:
            # LookbackBiasLogitsWarper types_first=True (logit_processors.py:99-108): eos ids + timed-event ids
            sp.lookback_types_first = 1
            ceos_all = list((getattr(tokenizer, "context_eos", {}) or {}).values())
            flags[np.asarray([tokenizer.eos_id] + ceos_all, dtype=np.int64)] |= FLAG_LOOKBACK_EOS
            for name in TIMED_EVENT_NAMES:
                if _has(tokenizer.event_start, name):
                    flags[_ev(tokenizer.event_start, name):_ev(tokenizer.event_end, name)] |= FLAG_TIMED
            need_flags = True
    sp.host_tok_flags = flags if need_flags else None
    sp.num_beams = num_beams               # (host attribute: beam search runs through mapperatorinator_amd.beam)
    sp.pad_id = int(gk.get("pad_token_id", getattr(tokenizer, "pad_id", 0)) or 0)
    sp.max_length = int(gk.get("max_length", max_target_positions))
    sp.seed = fresh_seed(gk.get("seed"), gk.get("seed_call_index")) if sp.do_sample else 0   # greedy decoding leaves the global generator alone
    # ConditionalTemperatureLogitsWarper looks at row 0's history for the whole batch in the reference
    # (logit_processors.py:75-80); callers that batch rows of DIFFERENT songs / shards, where the reference would have
    # run batch-1 calls, ask for the per-row form (scheduler.py, sharding.py)
    sp.cond_per_row = int(bool(gk.get("conditional_temperature_per_row", False)))
    sp.rng_row0 = int(gk.get("rng_row_offset", 0)) & 0xFFFFFFFF
    eos = get_eos_token_id(tokenizer, lookback_time=lookback_time, lookahead_time=lookahead_time,
                           context_type=context_type)
    return sp, eos


def sampling_from_processors(processors, vocab_size_out: int, *, do_sample=False, top_k=0, top_p=1.0, max_length: int,
                             pad_token_id: int = 0, seed=None):
    """The reference's live processor OBJECTS (the `logits_processor=` list that server.py:106-134 builds and hands
    to `model.generate`) -> MhSampling.  Recognised by class name so that both the reference's classes and HF's
    are accepted without importing either; anything else is refused -- a python callable cannot run inside the
    captured decode step."""
    sp = Sampling()
    sp.do_sample, sp.top_k, sp.top_p = int(bool(do_sample)), int(top_k or 0), float(1.0 if top_p is None else top_p)
    sp.temperature, sp.cfg_scale, sp.pad_id, sp.max_length = 1.0, 1.0, int(pad_token_id or 0), int(max_length)
    sp.seed = fresh_seed(seed) if sp.do_sample else 0
    flags = np.zeros(int(vocab_size_out), dtype=np.uint8)
    need_flags = False
    order = {"ClassifierFreeGuidanceLogitsProcessor": 0, "MonotonicTimeShiftLogitsProcessor": 1, "TimeshiftBias": 2,
             "TemperatureLogitsWarper": 3, "ConditionalTemperatureLogitsWarper": 3, "LookbackBiasLogitsWarper": 4}
    last = -1
    bias_range = None
    for proc in list(processors or []):
        name = type(proc).__name__
        if name not in order:
            raise NotImplementedError(f"logits processor {name} is not one of the reference's (server.py:106-134)")
        if order[name] <= last:
            raise NotImplementedError("processors must come in the reference's order (server.py:106-134)")
        last = order[name]
        if name == "ClassifierFreeGuidanceLogitsProcessor":
            sp.cfg_scale = float(proc.guidance_scale)
        elif name == "MonotonicTimeShiftLogitsProcessor":
            sp.ts_start, sp.ts_end = int(proc.time_shift_start), int(proc.time_shift_end)
            sos = [int(v) for v in torch.as_tensor(proc.sos_ids).tolist()]
            if len(sos) > 16:
                raise NotImplementedError("more than 16 SOS-type ids")
            sp.n_sos = len(sos)
            for i, v in enumerate(sos):
                sp.sos_ids[i] = v
        elif name == "TimeshiftBias":
            sp.timeshift_bias = float(proc.timeshift_bias)
            bias_range = (int(proc.time_range.start), int(proc.time_range.stop))
        elif name == "TemperatureLogitsWarper":
            sp.temperature = float(proc.temperature)
        elif name == "ConditionalTemperatureLogitsWarper":
            sp.temperature = float(proc.temperature)
            if len(proc.conditionals) > 3:
                raise NotImplementedError("more than 3 conditional temperature rules")
            sp.n_cond = len(proc.conditionals)
            for j, (t, ids, off) in enumerate(proc.conditionals):
                sp.cond_temp[j], sp.cond_offset[j] = float(t), int(off)
                flags[np.asarray(list(ids), dtype=np.int64)] |= FLAG_COND0 << j
                need_flags = True
        elif name == "LookbackBiasLogitsWarper":
            start, end = int(proc.lookback_start), int(proc.lookback_end)
            sp.lookback_mask_end = end
            if sp.ts_end > sp.ts_start and start != sp.ts_start:
                raise NotImplementedError("lookback range must start at the first TIME_SHIFT id")
            if sp.ts_end <= sp.ts_start:
                raise NotImplementedError("LookbackBiasLogitsWarper without MonotonicTimeShiftLogitsProcessor")
            if proc.types_first:
                sp.lookback_types_first = 1
                flags[torch.as_tensor(proc.eos_ids).cpu().numpy().astype(np.int64)] |= FLAG_LOOKBACK_EOS
                flags[torch.as_tensor(proc.timed_tokens).cpu().numpy().astype(np.int64)] |= FLAG_TIMED
                need_flags = True
    if bias_range is not None and bias_range != (sp.ts_start, sp.ts_end):
        raise NotImplementedError("TimeshiftBias over a range other than the TIME_SHIFT ids")
    if sp.cfg_scale <= 1.0:
        sp.cfg_scale = 1.0
    sp.host_tok_flags = flags if need_flags else None
    return sp


@torch.no_grad()
def model_generate(model, tokenizer, model_kwargs, generate_kwargs):
    """See module docstring.  `model_kwargs['inputs']`: raw audio float32 (B, Ns)."""
    generate_kwargs = dict(generate_kwargs)
    # conditioning inputs (modeling_mapperatorinator.py:174-228): used when the model carries the embedders, ignored
    # otherwise -- exactly the reference's `if self.do_*_embed` switches
    cond = getattr(model, "cond", None)
    row_bias = None
    if cond is not None and cond.active:
        vec = cond.vectors(model_kwargs["inputs"].shape[0], beatmap_idx=model_kwargs.get("beatmap_idx"),
                           difficulty=model_kwargs.get("difficulty"), mapper_idx=model_kwargs.get("mapper_idx"),
                           song_position=model_kwargs.get("song_position"))
        row_bias = cond.channels(vec, model.dtype) if cond.as_channels else cond.row_bias(vec, model.dtype)
    audio = model_kwargs["inputs"]
    prompt = model_kwargs["decoder_input_ids"]
    mask = model_kwargs.get("decoder_attention_mask")
    sp, eos = build_sampling(tokenizer, generate_kwargs, model.config.max_target_positions)
    if sp.max_length > model.config.max_target_positions:
        raise ValueError(f"max_length {sp.max_length} exceeds max_target_positions "
                         f"{model.config.max_target_positions}")
    pad_token_id = generate_kwargs.get("pad_token_id", getattr(tokenizer, "pad_id", None))

    neg, neg_mask = model_kwargs.get("negative_prompt"), model_kwargs.get("negative_prompt_attention_mask")
    if sp.cfg_scale > 1.0 and neg is None:
        # HF's processor would raise on the batch-size check (logits not doubled without a negative prompt)
        raise ValueError("cfg_scale > 1 needs model_kwargs['negative_prompt'] (modeling_mapperatorinator.py:243-254)")
    if sp.cfg_scale <= 1.0:
        # the reference doubles the batch whenever a negative prompt is passed and, without the CFG processor, HF
        # then fails on the shape mismatch; its own caller only passes one when cfg_scale > 1 (processor.py:1171)
        neg = neg_mask = None

    start = time.perf_counter()
    extra = {} if row_bias is None else dict(row_bias=row_bias)
    if getattr(sp, "num_beams", 1) > 1:
        # HF beam search (processor.py:159 `num_beams`; the timing generator uses two beams): mapperatorinator_amd/beam.py
        if generate_kwargs.get("cross_kv_fp8"):
            raise NotImplementedError("cross_kv_fp8 with beam search: the step-wise beam entry streams the bf16 cross K / V")
        out = model.engine.generate_beam(audio, prompt, mask, eos, sp, sp.num_beams, negative_prompt=neg,
                                         sample_fn=generate_kwargs.get("beam_sample_fn"),
                                         use_kernel=generate_kwargs.get("beam_use_kernel"), **extra)
    else:
        out = model.engine.generate(audio, prompt, mask, eos, sp, negative_prompt=neg, negative_mask=neg_mask,
                                    cross_kv_fp8=bool(generate_kwargs.get("cross_kv_fp8", False)), **extra)
    elapsed = time.perf_counter() - start
    result = out["tokens"]
    stats = _build_generation_stats(result, model_kwargs, pad_token_id, elapsed)
    return result, stats


# ---- request batching (second half) ---------------------------------------------------------------
class RequestBatcher:
    """The batching policy of the reference's `InferenceServer._batch_thread` (osuT5/osuT5/inference/server.py:343-424)
    in front of `model_generate`, without its control plane (unix socket, listener / client threads, idle monitor stay
    the reference's: its `InferenceServer` works unchanged once its module-level `model_generate` is this module's --
    INTEGRATION.md, tests/test_oracle_pinned.py::test_reference_inference_server_batches_into_our_model_generate).

    Same policy: requests are grouped by their generate kwargs; one batch takes whole or PARTIAL requests of the first
    group until `max_batch_size // batch_multiplier` rows are filled (2 x beams under guidance, server.py:353-355); every
    tensor with more than one dimension is left-padded to the widest request, `decoder_input_ids` decides how many pad
    columns are cut from each request's result again (:377-399); a request is answered when all its rows are done, with
    the summed token and time statistics (:401-417).

    `max_batch_size` defaults to 32 rows: the decode engine runs two 16-row chains and its step time does not drop
    below 32 rows (decode), so smaller batches only lose throughput.  64 rows (the engine's maximum: two 32-row
    chains) decode 14 % more tokens per second at 1.75 x the per-token latency (profiles/r06_small_batch_decode.txt)."""

    def __init__(self, model, tokenizer, max_batch_size: int = 32, generate_fn=None):
        self.model, self.tokenizer, self.max_batch_size = model, tokenizer, int(max_batch_size)
        self.generate_fn = generate_fn or model_generate
        self.grouped_requests: dict = {}
        self.prefetch = True          # start the next batch's H2D before this batch decodes (HIP engine only)
        self._staged, self._stager = None, None
        self._seed_calls: dict = {}   # explicit seed -> batches decoded with it BY THIS batcher (fresh_seed's call_index)

    def submit(self, model_kwargs: dict, generate_kwargs: dict) -> dict:
        """Queue one request (what `_client_handler` does on `conn.recv()`, :297-322); returns its record, whose
        'result' is filled by `step` once 'work_done' == 'total_work'."""
        record = dict(model_kwargs=model_kwargs, total_work=int(model_kwargs["inputs"].shape[0]), work_done=0, work_taken=0, result=None,
                      generated_tokens=0, elapsed_seconds=0.0, done=False, error=None)
        self.grouped_requests.setdefault(frozenset(generate_kwargs.items()), []).append(record)
        return record

    @property
    def pending(self) -> bool:
        return bool(self.grouped_requests) or self._staged is not None

    @staticmethod
    def _cut(model_kwargs: dict, start: int, length: int) -> dict:
        return {k: v[start:start + length] if isinstance(v, torch.Tensor) else v for k, v in model_kwargs.items()}

    def _take_batch(self):
        """-> (generate kwargs, [(cut model_kwargs, request, rows)]) of the next batch, or None"""
        if not self.grouped_requests:
            return None
        key = next(iter(self.grouped_requests))
        requests = self.grouped_requests[key]
        generate_kwargs = dict(key)
        cfg_scale, num_beams = generate_kwargs.get("cfg_scale", 1.0), generate_kwargs.get("num_beams", 1)
        multiplier = 2 * num_beams if cfg_scale > 1 else num_beams
        room = self.max_batch_size // multiplier
        if room <= 0:
            raise ValueError(f"max_batch_size {self.max_batch_size} holds no row at batch multiplier {multiplier}")
        batch = []
        while room > 0 and requests:
            req = requests.pop(0)
            left = req["total_work"] - req["work_taken"]       # rows not yet handed to a batch (the batch before this one
            work = min(left, room)                              # may still be decoding: `work_done` lags behind)
            batch.append((self._cut(req["model_kwargs"], req["work_taken"], work), req, work))
            req["work_taken"] += work
            room -= work
            if left > work:
                requests.insert(0, req)            # the rest of it leads the next batch
        if not requests:
            del self.grouped_requests[key]
        return generate_kwargs, batch

    def _collate(self, batch):
        keys = [k for k, v in batch[0][0].items() if v is not None]
        paddings = [0] * len(batch)
        collated = {}
        for k in keys:
            parts = [b[0][k] for b in batch]
            if isinstance(parts[0], torch.Tensor) and parts[0].dim() > 1:
                width = max(t.size(-1) for t in parts)
                if k == "decoder_input_ids":
                    paddings = [width - t.size(-1) for t in parts]
                parts = [torch.nn.functional.pad(t, (width - t.size(-1), 0)) for t in parts]
            collated[k] = torch.cat(parts, dim=0)
        return collated, paddings

    def _stage(self, taken):
        """collate a taken batch and start the H2D copy of its audio (pinned buffer, copy stream): by the time the batch
        before it has decoded, the audio sits in HBM"""
        if taken is None:
            return None
        generate_kwargs, batch = taken
        collated, paddings = self._collate(batch)
        ev = None
        engine = getattr(self.model, "engine", None)
        if self.prefetch and engine is not None and isinstance(getattr(engine, "device", None), torch.device) \
                and engine.device.type == "cuda":
            if self._stager is None:
                from .t5_engine import HostStager
                self._stager = HostStager(engine.device)
            collated["inputs"], ev = self._stager.stage(collated["inputs"], torch.float32)
        return generate_kwargs, batch, collated, paddings, ev

    def step(self) -> int:
        """One batch of the first group through `model_generate`; returns the number of rows it held (0: nothing queued).
        The batch AFTER it is taken and its audio sent towards the device before this one starts decoding."""
        cur = self._staged if self._staged is not None else self._stage(self._take_batch())
        self._staged = None
        if cur is None:
            return 0
        generate_kwargs, batch, collated, paddings, ev = cur
        self._staged = self._stage(self._take_batch())
        if ev is not None:
            torch.cuda.current_stream(collated["inputs"].device).wait_event(ev)
        if generate_kwargs.get("seed") is not None and generate_kwargs.get("seed_call_index") is None:
            n = self._seed_calls.get(generate_kwargs["seed"], 0)
            self._seed_calls[generate_kwargs["seed"]] = n + 1
            generate_kwargs = dict(generate_kwargs, seed_call_index=n)
        try:
            outputs, stats = self.generate_fn(self.model, self.tokenizer, collated, generate_kwargs)
        except BaseException as e:
            # the reference answers every request of a failed batch with RETRY_SIGNAL (server.py:418-424) instead of leaving
            # its client waiting: here each of them is closed with the error (rows of it still queued or staged are dropped
            # with it -- the caller resubmits the whole request), then the error goes on to whoever drives the batcher
            self._fail([req for _, req, _ in batch], e)
            raise
        per_row = stats.get("generated_tokens_per_sample", [])
        row = 0
        for (_, req, work), pad in zip(batch, paddings):
            if req["error"] is not None:           # closed by a failure of an earlier part: its rows are not delivered
                row += work
                continue
            out = outputs[row:row + work, pad:]
            req["generated_tokens"] += sum(per_row[row:row + work])
            row += work
            if req["result"] is not None and req["result"].shape[1] != out.shape[1]:
                # parts of one request decoded in different batches can stop at different lengths; the reference's
                # torch.cat raises here (and its client retries): widen with the pad id finished rows already carry
                width = max(req["result"].shape[1], out.shape[1])
                fill = generate_kwargs.get("pad_token_id", getattr(self.tokenizer, "pad_id", 0))
                req["result"] = torch.nn.functional.pad(req["result"], (0, width - req["result"].shape[1]), value=fill)
                out = torch.nn.functional.pad(out, (0, width - out.shape[1]), value=fill)
            req["result"] = out if req["result"] is None else torch.cat((req["result"], out), dim=0)
            req["work_done"] += work
            req["elapsed_seconds"] += stats.get("elapsed_seconds", 0.0)
            if req["work_done"] >= req["total_work"]:
                secs, toks = req["elapsed_seconds"], req["generated_tokens"]
                req["result"] = {"output": req["result"],
                                 "stats": {"generated_tokens": toks, "elapsed_seconds": secs,
                                           "tokens_per_second": toks / secs if secs > 0 else 0.0}}
                req["done"] = True
        return row

    def _fail(self, reqs, exc) -> None:
        dead = {id(r) for r in reqs}
        for r in reqs:
            r["error"], r["done"], r["result"] = exc, True, None
        for key in list(self.grouped_requests):
            left = [r for r in self.grouped_requests[key] if id(r) not in dead]
            if left:
                self.grouped_requests[key] = left
            else:
                del self.grouped_requests[key]
        # rows of a failed request that the prefetched batch already holds stay in it (the batch is collated and on its way to
        # the device); `step` skips them when it hands the results out

    def drain(self) -> int:
        """Run batches until nothing is queued; returns how many batches ran."""
        n = 0
        while self.step():
            n += 1
        return n
