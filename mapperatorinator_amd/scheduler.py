"""Window scheduler for sequential songs on top of the HIP engine.

The reference generates a song window by window (`Processor.generate_sequential`, osuT5/osuT5/inference/
processor.py:308-368): window i's prompt contains the events that window i-1 produced, so the windows of ONE song are
a chain; each iteration runs mel -> encoder -> decode for a single window (batch 1) and hops through the host between
the stages.  Two things in that loop do not depend on the chain and are taken out of it here:

  * the mel + encoder + cross-K/V of EVERY window depends on the audio only: all windows of all songs are encoded up
    front in MFMA-sized batches and their cross-attention K/V stays resident in HBM (46 MB per 10 s window at
    osuT5-base; a 3-minute song is 0.8 GB -- hundreds of songs fit in 288 GB);
  * different songs are independent: wave w decodes window w of every song that still has one as ONE batch
    (rows grouped by identical generate kwargs -- the lookback / lookahead EOS windows differ on a song's first and
    last window).

Prompt building stays with the caller -- it is the reference's own integer host logic (`get_prompts`,
`prepare_context_sequences`, `add_predicted_tokens_to_context`, processor.py:1022-1052,1226-1268): a `SongJob` supplies
`prompt_fn(window) -> model_kwargs-like dict` and receives every finished window through `on_result`.  Rows are
batch-invariant in the engine, so the tokens are exactly those of the reference-shaped loop (one `model_generate` per
window), which is what tests/test_gpu_t5.py::test_window_scheduler_matches_sequential_loop checks.
"""
from __future__ import annotations

import dataclasses
import time
from typing import Callable, Optional

import torch

from .server import _build_generation_stats, build_sampling
from .t5_engine import HostStager


@dataclasses.dataclass
class SongJob:
    """frames: (n_windows, samples) raw audio windows of one song (what `prepare_frames` yields per window).
    prompt_fn(window_index) -> dict(decoder_input_ids=(1, P) int64, [decoder_attention_mask], [negative_prompt],
        [generate_kwargs=dict]) -- called when the previous window of THIS song has been delivered.
    on_result(window_index, tokens (P + new,) int64 cpu incl. the prompt, stats dict)."""
    frames: torch.Tensor
    prompt_fn: Callable[[int], dict]
    on_result: Callable[[int, torch.Tensor, dict], None]
    generate_kwargs: dict = dataclasses.field(default_factory=dict)
    # models with conditioning embedders: window index -> dict(difficulty=, mapper_idx=, song_position=, beatmap_idx=) of
    # scalars / (2,) for THIS window (the reference sets song_position per window, processor.py:341-345)
    conditioning_fn: Optional[Callable[[int], dict]] = None


def _left_pad(rows, pad_id: int, dtype):
    width = max(r.shape[-1] for r in rows)
    out = torch.full((len(rows), width), pad_id, dtype=dtype)
    for i, r in enumerate(rows):
        out[i, width - r.shape[-1]:] = r.reshape(-1).to(dtype)
    return out


class SequentialWindowScheduler:
    def __init__(self, model, tokenizer, *, encode_batch: int = 32, decode_batch: int = 32):
        if decode_batch > 64 or decode_batch < 1:
            raise ValueError("decode_batch must be in [1, 64] (32 at most when windows run under classifier-free guidance)")
        self.model, self.tokenizer = model, tokenizer
        self.engine = model.engine
        self.encode_batch, self.decode_batch = int(encode_batch), int(decode_batch)
        self.stats = dict(windows=0, decode_calls=0, encode_calls=0, generated_tokens=0, elapsed_seconds=0.0)
        self._seed_calls: dict = {}       # explicit seed -> decode calls made with it BY THIS scheduler

    # ---- stage 1: everything that depends on the audio only ------------------------------------------------
    @torch.no_grad()
    def encode_all(self, jobs: list[SongJob]):
        """-> per job a tensor [n_dec_layers, 2, n_windows, H, L, 64] of resident cross-attention K/V."""
        eng = self.engine
        flat = torch.cat([j.frames for j in jobs], 0)
        row_bias = self._row_bias(jobs)
        parts = []
        on_gpu = torch.device(eng.device).type == "cuda"     # (CPU stand-in engines of the host tests take the plain path)
        stager = HostStager(eng.device) if on_gpu else None
        eb = self.encode_batch

        def stage(a):
            part = flat[a:a + eb]
            return stager.stage(part, torch.float32) if on_gpu else (part.to(eng.device, torch.float32), None)
        nxt = stage(0)                                        # batch a + 1 crosses PCIe while batch a is encoded
        eng._enter()
        with eng.on_stream():
            for a in range(0, flat.shape[0], eb):
                cur, ev = nxt
                if a + eb < flat.shape[0]:
                    nxt = stage(a + eb)
                if ev is not None:
                    eng.stream.wait_event(ev)
                rb = None if row_bias is None else row_bias[a:a + eb]
                parts.append(eng.cross_kv(eng.encode_mel(eng.mel(cur), row_bias=rb)))
                if on_gpu:
                    cur.record_stream(eng.stream)
                self.stats["encode_calls"] += 1
            kv = torch.cat(parts, 2) if len(parts) > 1 else parts[0]
        eng._leave()
        out, a = [], 0
        for j in jobs:
            out.append(kv[:, :, a:a + j.frames.shape[0]])
            a += j.frames.shape[0]
        return out

    def _row_bias(self, jobs):
        """(total windows, d_model) fp32 for models with conditioning embedders (one row per window, in `frames` order),
        None otherwise."""
        cond = getattr(self.model, "cond", None)
        if cond is None or not cond.active:
            return None
        keys = ("beatmap_idx", "difficulty", "mapper_idx", "song_position")
        rows = {k: [] for k in keys}
        for j in jobs:
            if j.conditioning_fn is None:
                raise ValueError("this model has conditioning embedders: every SongJob needs a conditioning_fn")
            for w in range(j.frames.shape[0]):
                kw = j.conditioning_fn(w)
                for k in keys:
                    rows[k].append(kw.get(k))
        n = len(rows["difficulty"])
        kw = {}
        for k in keys:
            if all(v is None for v in rows[k]):
                continue
            if any(v is None for v in rows[k]):
                raise ValueError(f"conditioning input {k!r} is given for some windows only")
            kw[k] = torch.stack([torch.as_tensor(v, dtype=torch.float32 if k in ("difficulty", "song_position") else torch.long)
                                 for v in rows[k]])
        vec = cond.vectors(n, **kw)
        return cond.channels(vec, self.model.dtype) if cond.as_channels else cond.row_bias(vec, self.model.dtype)

    # ---- stage 2: waves of dependent windows -----------------------------------------------------------------
    @torch.no_grad()
    def run(self, jobs: list[SongJob]):
        eng, tok = self.engine, self.tokenizer
        start = time.perf_counter()
        kvs = self.encode_all(jobs)
        n_waves = max(j.frames.shape[0] for j in jobs)
        pad_id = int(getattr(tok, "pad_id", 0))
        for w in range(n_waves):
            active = [i for i, j in enumerate(jobs) if w < j.frames.shape[0]]
            asks = {}
            for i in active:
                ask = dict(jobs[i].prompt_fn(w))
                gk = dict(jobs[i].generate_kwargs, **ask.pop("generate_kwargs", {}))
                key = repr(sorted(gk.items(), key=lambda kv_: kv_[0]))
                asks.setdefault(key, []).append((i, ask, gk))
            for group in asks.values():
                # guidance doubles the decode batch (negative rows + prompt rows): half as many windows per call
                step = max(1, self.decode_batch // 2) if float(group[0][2].get("cfg_scale", 1.0)) > 1.0 else self.decode_batch
                # beam search decodes (window, beam) rows: num_beams rows per window (processor.py:159; the timing pass uses 2)
                step = max(1, step // max(1, int(group[0][2].get("num_beams", 1) or 1)))
                for a in range(0, len(group), step):
                    self._decode_group(jobs, kvs, w, group[a:a + step], pad_id)
        self.stats["elapsed_seconds"] += time.perf_counter() - start
        return self.stats

    def _decode_group(self, jobs, kvs, w, group, pad_id):
        eng, tok = self.engine, self.tokenizer
        # rows of different songs share this batch where the reference runs one batch-1 call per window: the conditional
        # temperature must look at each row's OWN history (the reference's processor reads row 0 of its batch)
        gk = dict(group[0][2], conditional_temperature_per_row=True)
        if gk.get("seed") is not None and gk.get("seed_call_index") is None:
            # explicit seed: the n-th decode call of THIS scheduler with that seed draws from stream (seed, n) -- the count is
            # the scheduler's own, so a run reproduces whatever else samples in the process (server.fresh_seed)
            n = self._seed_calls.get(gk["seed"], 0)
            self._seed_calls[gk["seed"]] = n + 1
            gk["seed_call_index"] = n
        sp, eos = build_sampling(tok, gk, self.model.config.max_target_positions)
        cfg = sp.cfg_scale > 1.0
        nb = int(getattr(sp, "num_beams", 1) or 1)
        if len(group) * (2 if cfg else 1) * nb > 64:
            raise ValueError(f"{len(group)} windows{' x 2 (guidance)' if cfg else ''}{f' x {nb} beams' if nb > 1 else ''} exceed "
                             f"the engine's 64-row decode batch")
        prompts = _left_pad([g[1]["decoder_input_ids"] for g in group], pad_id, torch.int64)
        masks = None
        if any(g[1].get("decoder_attention_mask") is not None for g in group):
            masks = _left_pad([g[1]["decoder_attention_mask"] if g[1].get("decoder_attention_mask") is not None
                               else torch.ones_like(g[1]["decoder_input_ids"]) for g in group], 0, torch.uint8)
        elif any(g[1]["decoder_input_ids"].shape[-1] != prompts.shape[1] for g in group):
            # ragged prompts: only the left padding added HERE must not be attended -- a pad_id inside a caller's own
            # prompt stays visible, exactly as in the batch-1 call without a mask
            masks = _left_pad([torch.ones_like(g[1]["decoder_input_ids"]) for g in group], 0, torch.uint8)
        neg = None
        if cfg:
            if any(g[1].get("negative_prompt") is None for g in group):
                raise ValueError("cfg_scale > 1 needs a negative_prompt for every window")
            neg_rows = _left_pad([g[1]["negative_prompt"] for g in group], pad_id, torch.int64)
            if neg_rows.shape[1] > prompts.shape[1]:
                raise ValueError("negative prompt longer than the prompt")
            # as prepare_inputs_for_generation: the negative prompt overwrites the first columns of a copy of the prompt
            neg = prompts.clone()
            neg[:, :neg_rows.shape[1]] = neg_rows
        dev = eng.device
        eos_table = torch.zeros(eng.packed.vocab_out, dtype=torch.uint8)
        eos_table[torch.as_tensor(sorted(set(int(e) for e in eos if 0 <= int(e) < eng.packed.vocab_out)),
                                  dtype=torch.long)] = 1
        eos_ids = torch.as_tensor(sorted(set(int(e) for e in eos)), dtype=torch.int64)
        t0 = time.perf_counter()
        eng._enter()
        with eng.on_stream():
            kv = torch.stack([kvs[i][:, :, w] for i, _, _ in group], 2).contiguous()   # rows of this wave, gathered
            p_all = torch.cat([neg, prompts], 0) if cfg else prompts
            m_all = None if masks is None else (torch.cat([masks, masks], 0) if cfg else masks)
            # generate_kwargs["cross_kv_fp8"]: the token steps stream an e4m3 copy of this wave's cross K / V
            if gk.get("cross_kv_fp8") and nb > 1:
                raise NotImplementedError("cross_kv_fp8 with beam search: the step-wise beam entry streams the bf16 cross K / V")
            kv8 = eng.cross_kv_fp8(kv) if gk.get("cross_kv_fp8") else None
            if nb == 1:
                tokens, n_out, _ = eng.decode(kv, p_all.to(dev, torch.int32).contiguous(),
                                              None if m_all is None else m_all.to(dev).contiguous(),
                                              eos_table.to(dev), sp, kv_fp8=kv8)
        eng._leave()
        if nb > 1:
            # HF beam search over the step-wise decode entry (beam.py), the windows of this wave as its batch: every window's
            # hypotheses are ranked among themselves only, so a window decodes as in the reference's batch-1 call; rows that
            # end early carry HF's fill (the first EOS id) and are cut at their first EOS-set id below like any other row
            from .beam import beam_search
            result = beam_search(eng, kv, p_all, m_all, eos, sp, nb, sample_fn=gk.get("beam_sample_fn")).to(torch.int64).cpu()
        else:
            eng.synchronize()
            n_cols = int(n_out.item())
            if cfg:
                tokens = tokens[len(group):]
            result = tokens[:, :n_cols].to(torch.int64).cpu()
        elapsed = time.perf_counter() - t0
        self.stats["decode_calls"] += 1
        P = prompts.shape[1]
        redo = {}
        for r, (i, ask, _) in enumerate(group):
            own = ask["decoder_input_ids"].shape[-1]
            row = result[r, P - own:]                      # strip the padding this batch added on the left
            # a row that finished early carries pad_id up to the batch's longest row: cut after its first EOS-set id,
            # which is where a batch-1 call for this window would have ended
            body = row[own:]
            hit = torch.isin(body, eos_ids).nonzero()
            if hit.numel():
                row = row[:own + int(hit[0]) + 1]
            elif own < P and row.shape[0] < sp.max_length:
                # no EOS and the row stopped at max_length COLUMNS, `P - own` of which were this batch's left padding: the
                # reference's batch-1 call (no padding) would have gone on to max_length tokens.  Rare (windows end by
                # EOS); such rows are decoded again among rows of their own prompt length, i.e. without padding.
                redo.setdefault(own, []).append(group[r])
                continue
            st = _build_generation_stats(row[None], dict(decoder_input_ids=ask["decoder_input_ids"].reshape(1, -1)),
                                         pad_id, elapsed)
            self.stats["windows"] += 1
            self.stats["generated_tokens"] += st["generated_tokens"]
            jobs[i].on_result(w, row, st)
        for sub in redo.values():
            self._decode_group(jobs, kvs, w, sub, pad_id)


# ---- the reference's own sequential loop on the scheduler -------------------------------------------------------------
def processor_generate_kwargs(proc, **window_kwargs) -> dict:
    """The kwargs `Processor.model_generate` (osuT5/osuT5/inference/processor.py:155-170) hands to the module-level
    `model_generate`: the window's own (lookback_time, lookahead_time, context_type) + the processor's sampling knobs."""
    return dict(window_kwargs, precision=proc.precision, do_sample=proc.do_sample, num_beams=proc.num_beams, top_p=proc.top_p,
                top_k=proc.top_k, max_length=proc.tgt_seq_len, cfg_scale=proc.cfg_scale, timeshift_bias=proc.timeshift_bias,
                types_first=proc.types_first, temperature=proc.temperature, timing_temperature=proc.timing_temperature,
                mania_column_temperature=proc.mania_column_temperature, taiko_hit_temperature=proc.taiko_hit_temperature)


def processor_song_job(proc, *, sequences, in_context, out_context, context_index: int, req_special_tokens,
                       model_kwargs: Optional[dict] = None) -> SongJob:
    """ONE pass of the reference's `Processor.generate_sequential` (processor.py:308-368) over output context
    `context_index` of one song, cut at its `self.model_generate(...)` call: everything in front of the call (the window's
    prompts from the contexts as they stand -- `prepare_context_sequences`, `get_prompts`, `pad_prompts` -- the lookback /
    lookahead EOS windows, the per-window song position) is `prompt_fn` / `conditioning_fn`, everything behind it
    (`_record_generation_stats`, `result[0, max_len:]`, `add_predicted_tokens_to_context`) is `on_result`.  `proc` is the
    reference's own `Processor` (duck-typed: its methods and attributes are used as they are -- the host logic stays the
    reference's), so a song decoded through SequentialWindowScheduler leaves `out_context` exactly as the reference loop
    does; many songs (one job each) share the scheduler's waves.  Several output contexts of a song = one scheduler run
    per `context_index`, in order, as the reference's outer loop does."""
    frames_all, frame_times, song_length = sequences
    n = len(frames_all)
    context = out_context[context_index]
    pending = {}
    pad_id = proc.tokenizer.pad_id
    model_kwargs = dict(model_kwargs or {})

    def prompt_fn(w: int) -> dict:
        frame_time = frame_times[w].item()
        trim_lookback = w != 0 and proc.lookback_time > 0
        trim_lookahead = w != n - 1
        cond_prompt, uncond_prompt = proc.get_prompts(
            proc.prepare_context_sequences(in_context, frame_time, False, req_special_tokens),
            proc.prepare_context_sequences(out_context[:context_index + 1], frame_time, True, req_special_tokens))
        [prompt, uncond_prompt], max_len = proc.pad_prompts([cond_prompt, uncond_prompt])
        pending[w] = (frame_time, max_len, trim_lookback, trim_lookahead)
        ask = dict(decoder_input_ids=prompt, decoder_attention_mask=prompt.ne(pad_id),
                   generate_kwargs=processor_generate_kwargs(
                       proc, lookback_time=proc.lookback_time if trim_lookback else 0,
                       lookahead_time=proc.lookahead_time if trim_lookahead else 0, context_type=context["context_type"].value))
        if uncond_prompt is not None:
            ask["negative_prompt"] = uncond_prompt
        return ask

    def on_result(w: int, row: torch.Tensor, stats: dict) -> None:
        frame_time, max_len, trim_lookback, trim_lookahead = pending.pop(w)
        proc._record_generation_stats(stats)
        proc.add_predicted_tokens_to_context(context, row[max_len:].cpu(), frame_time, trim_lookback, trim_lookahead)

    conditioning_fn = None
    if getattr(proc, "do_song_position_embed", False) or any(k in model_kwargs for k in ("difficulty", "mapper_idx", "beatmap_idx")):
        def conditioning_fn(w: int) -> dict:
            kw = {k: (v[0] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == 1 else v)
                  for k, v in model_kwargs.items() if k in ("difficulty", "mapper_idx", "beatmap_idx")}
            if getattr(proc, "do_song_position_embed", False):
                ft = frame_times[w].item()
                kw["song_position"] = torch.tensor([ft / song_length, (ft + proc.miliseconds_per_sequence) / song_length],
                                                   dtype=torch.float32)
            return kw

    frames = torch.stack([proc.prepare_frames(f)[0] for f in frames_all])
    return SongJob(frames=frames, prompt_fn=prompt_fn, on_result=on_result, conditioning_fn=conditioning_fn)
