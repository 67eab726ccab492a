"""Beam search on the HIP engine (`num_beams > 1`).

The reference hands `num_beams` to HF `generate` (osuT5/osuT5/inference/processor.py:147,159; its timing generator decodes
with two beams, config.py:71) and reorders its static caches per step (`MapperatorinatorCache.reorder_cache`,
inference/cache_utils.py:16-20).  The selection logic is HF's (third-party; the algorithm below restates the published,
vectorised `GenerationMixin._beam_search` -- transformers >= 4.50; line references are to the installed 5.x
`generation/utils.py`): per step

    log_probs = processors(sequences, log_softmax(logits))            (b: processors act on LOG-PROBABILITIES here)
    accumulated = log_probs + running_beam_scores                      -> top K = max(2, 1 + #eos) * num_beams continuations
    candidates that hit an EOS id / max_length leave the running set   (e), the next `num_beams` others keep running
    finished hypotheses among the TOP num_beams candidates are merged into the finished set by score / length ** penalty   (f)
    the cache rows follow the surviving beams                           (g)
    stop when no running beam can still beat the worst finished one (early_stopping = False heuristic)

The device side is the step-wise entry of the library (`mh_t5_step`: one decoder position for all (chunk, beam) rows, raw
logits out; `mh_t5_reorder_cache`).  Round 6: for greedy beams the WHOLE bookkeeping above is one kernel per token, `mh_beam_step`
(csrc/beam.hip: log_softmax, guidance, processors, top-K over beams x V by an in-LDS sort, EOS split, next-beam selection, finished-set
merge, the early-stopping heuristic) -- `_beam_search_kernel` below; the host reads three flags per chunk and step.  The torch-op
form (`beam_search` with use_kernel=False; ~40 ATen launches per token) stays for beam-SAMPLE, whose continuations are drawn by
torch.multinomial / an injected sampler, and as the cross-check of the kernel (tests run both against the reference goldens).  The logits processors of
server.py:106-134 are applied here with torch ops (the in-kernel sampler of the greedy / sampling path selects per row and
cannot rank across beams): classifier-free guidance, MonotonicTimeShift, TimeshiftBias, (Conditional)Temperature and the lookback
mask; the types_first lookback renormalisation is refused in beam mode.

Beam-sample (`do_sample` with beams, round 5; processor.py:147-160 hands both to `generate`): HF appends its top-k / top-p warpers
BEHIND the reference's processor list with `min_tokens_to_keep = #eos + 1` (utils.py `_get_logits_processor`), and
`_get_top_k_continuations` draws the K continuations with `torch.multinomial(softmax(accumulated), K)` -- without replacement, in
draw order (not sorted: "inside the top num_beams" then means "among the first num_beams draws").  The draw uses torch's generator
on the engine's device, like HF on a GPU; `sample_fn(probs, k)` replaces it (parity tests inject the sampler the reference golden
was drawn with).

Guidance under beams (round 5; the timing pass sets beams, `super_timing_generator.py:28`, and `processor.py:709` halves its batch
for exactly this case) follows what the reference + HF do, quirk included:
  * `prepare_inputs_for_generation` (modeling_mapperatorinator.py:243-254) doubles the (chunk, beam) rows every step: the first
    half carries the negative prompt over the first columns, the second half is the prompt's own rows;
  * HF's `ClassifierFreeGuidanceLogitsProcessor` runs FIRST in the list, on log_softmax(logits) of all 2R rows, and treats the
    first half as the conditional one: guided = second + (first - second) * scale, R rows; the other processors see the prompt rows'
    sequences;
  * `MapperatorinatorCache.reorder_cache` (inference/cache_utils.py:16-20) gathers the doubled cache with `beam_idx.repeat(2)` --
    indices into the FIRST half for both halves: from the first reorder on, the prompt half's self-attention cache holds copies of the
    negative half's rows.  Reproduced verbatim (golden `t5_tiny_beam.npz`, runs `b2g` / `b3g`).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def _gather(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.take_along_dim(t, idx, dim=1)


class BeamProcessors:
    """The reference's processor list (server.py:106-134) on (rows, V) LOG-PROBABILITIES, from an MhSampling struct."""

    def __init__(self, sp, device, min_tokens_to_keep: int = 1):
        self.min_tokens_to_keep = int(min_tokens_to_keep)
        if sp.lookback_types_first and sp.lookback_mask_end > sp.ts_start:
            raise NotImplementedError("beam search with the types_first lookback renormalisation is not on the HIP path")
        self.sp = sp
        self.sos = torch.tensor([sp.sos_ids[i] for i in range(sp.n_sos)], dtype=torch.long, device=device)
        flags = getattr(sp, "host_tok_flags", None)
        self.flags = None if flags is None else torch.as_tensor(flags, dtype=torch.uint8, device=device)

    def __call__(self, ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        sp = self.sp
        scores = scores.clone()
        R, T = ids.shape
        # MonotonicTimeShiftLogitsProcessor (logit_processors.py:136-183)
        if sp.ts_end > sp.ts_start:
            idx = torch.arange(T, device=ids.device).expand(R, -1)
            is_ts = (ids >= sp.ts_start) & (ids < sp.ts_end)
            is_sos = torch.isin(ids, self.sos)
            last_ts = torch.where(is_ts, idx, -1).max(1).values
            last_sos = torch.where(is_sos, idx, -1).max(1).values
            val = torch.where(last_ts != -1, ids[torch.arange(R, device=ids.device), last_ts.clamp(min=0)] - sp.ts_start, 0)
            apply = (last_ts != -1) & (last_ts > last_sos)
            vocab = torch.arange(sp.ts_start, sp.ts_end, device=ids.device)
            bad = apply[:, None] & (vocab[None, :] < (sp.ts_start + val)[:, None])
            scores[:, sp.ts_start:sp.ts_end] = scores[:, sp.ts_start:sp.ts_end].masked_fill(bad, float("-inf"))
            if sp.timeshift_bias != 0.0:                                     # TimeshiftBias (:36-44)
                scores[:, sp.ts_start:sp.ts_end] += sp.timeshift_bias
        # (Conditional)TemperatureLogitsWarper (:47-82): row 0's history picks the temperature of the whole call
        temp = torch.full((R,), float(sp.temperature), device=ids.device)
        if sp.n_cond > 0:
            if self.flags is None:
                raise ValueError("conditional temperature needs tok_flags")
            rows = ids if sp.cond_per_row else ids[:1].expand(R, -1)
            chosen = torch.zeros(R, dtype=torch.bool, device=ids.device)
            for j in range(sp.n_cond):
                off = int(sp.cond_offset[j])
                if T >= off:
                    hit = ((self.flags[rows[:, T - off]] & (2 << j)) != 0) & ~chosen
                    temp = torch.where(hit, torch.full_like(temp, float(sp.cond_temp[j])), temp)
                    chosen |= hit
        scores = scores / temp[:, None]
        if sp.lookback_mask_end > sp.ts_start:                                # LookbackBiasLogitsWarper, types_first False (:111-114)
            scores[:, sp.ts_start:sp.lookback_mask_end] = float("-inf")
        if sp.do_sample:      # HF's own warpers, appended behind the list: TopK then TopP, at least #eos + 1 tokens kept under beams
            keep = self.min_tokens_to_keep
            if sp.top_k > 0:
                kth = torch.topk(scores, min(max(int(sp.top_k), keep), scores.shape[-1]))[0][..., -1, None]
                scores = scores.masked_fill(scores < kth, float("-inf"))
            if sp.top_p < 1.0:
                srt, order = torch.sort(scores, descending=False)
                drop = srt.softmax(dim=-1).cumsum(dim=-1) <= (1 - float(sp.top_p))
                drop[..., -keep:] = False
                scores = scores.masked_fill(drop.scatter(1, order, drop), float("-inf"))
        return scores


def kernel_path_available(sp, num_beams: int, vocab_out: int, n_eos: int) -> bool:
    """mh_beam_step covers greedy beams (no beam-sample), 2 .. 8 beams, K <= 4096 candidates, beams x V scores + the K candidates
    within 120 KB of LDS."""
    K = min(max(2, 1 + n_eos) * num_beams, num_beams * vocab_out)
    k_pad = 1 << max(0, (K - 1).bit_length())
    return (not sp.do_sample) and 2 <= num_beams <= 8 and K <= 4096 and num_beams * vocab_out * 4 + 16 + k_pad * 8 <= 120 * 1024


@torch.no_grad()
def _beam_search_kernel(engine, cross_kv, prompt, prompt_mask, eos_ids, sp, num_beams, length_penalty, early_stopping) -> torch.Tensor:
    """`beam_search` with the per-token bookkeeping in mh_beam_step: per token mh_t5_step -> mh_beam_step -> mh_t5_reorder_cache and ONE
    D2H copy of 3 G flags (HF's loop condition); the hypotheses live in int32 device arrays that the kernel ping-pongs."""
    dev, lib, p = engine.device, engine.lib, engine.packed
    cfg = sp.cfg_scale > 1.0
    neg_prompt = neg_mask = None
    if cfg:
        if prompt.shape[0] % 2:
            raise ValueError("guidance: the prompt batch must be [negative rows | prompt rows]")
        half = prompt.shape[0] // 2
        neg_prompt, prompt = prompt[:half], prompt[half:]
        neg_mask = None if prompt_mask is None else prompt_mask[:half]
        prompt_mask = None if prompt_mask is None else prompt_mask[half:]
        cross_kv = torch.cat([cross_kv, cross_kv], dim=2)
    G, P = prompt.shape
    nb = int(num_beams)
    R = G * nb
    RE = 2 * R if cfg else R
    if RE > 64:
        raise ValueError(f"{G} chunks x {nb} beams{' x 2 (guidance)' if cfg else ''} exceed the engine's 64-row decode batch")
    V = p.vocab_out
    max_length = int(sp.max_length)
    if not (1 <= P < max_length <= p.tgt_len):
        raise ValueError("prompt / max_length do not fit the cache")
    if sp.lookback_types_first and sp.lookback_mask_end > sp.ts_start:
        raise NotImplementedError("beam search with the types_first lookback renormalisation is not on the HIP path")
    eos_list = [int(e) for e in eos_ids]
    K = min(max(2, 1 + len(eos_list)) * nb, nb * V)
    fill = int(sp.pad_id) or (eos_list[0] if eos_list else -1)
    n_new = max_length - P
    eos_table = torch.zeros(V, dtype=torch.uint8, device=dev)
    good = [e for e in eos_list if 0 <= e < V]
    if good:
        eos_table[torch.tensor(good, dtype=torch.long, device=dev)] = 1
    flags_h = getattr(sp, "host_tok_flags", None)
    if flags_h is not None:
        flags_d = torch.as_tensor(flags_h, dtype=torch.uint8).to(dev)
        sp.tok_flags = flags_d.data_ptr()
    elif sp.n_cond:
        raise ValueError("conditional temperature needs tok_flags (build the sampling struct with server.build_sampling)")
    ids0 = prompt.to(dev, torch.int32).repeat_interleave(nb, 0)
    mask = None if prompt_mask is None else prompt_mask.to(dev).to(torch.uint8).repeat_interleave(nb, 0).contiguous()
    ids0e = ids0
    if cfg:
        ids0e = torch.cat([neg_prompt.to(dev, torch.int32).repeat_interleave(nb, 0), ids0], 0)
        if mask is not None:
            mask = torch.cat([neg_mask.to(dev).to(torch.uint8).repeat_interleave(nb, 0), mask], 0).contiguous()

    def state():
        run = torch.full((G, nb, max_length), fill, dtype=torch.int32, device=dev)
        run[:, :, :P] = ids0.view(G, nb, P)
        rs = torch.zeros((G, nb), dtype=torch.float32, device=dev)
        rs[:, 1:] = -1e9
        return dict(run=run, rs=rs, rb=torch.full((G, nb, n_new), -1, dtype=torch.int32, device=dev), seq=run.clone(),
                    bs=torch.full((G, nb), -1e9, dtype=torch.float32, device=dev),
                    bb=torch.full((G, nb, n_new), -1, dtype=torch.int32, device=dev),
                    fin=torch.zeros((G, nb), dtype=torch.uint8, device=dev))
    st = [state(), state()]
    heuristic_open = torch.ones(G, dtype=torch.uint8, device=dev)
    src = torch.zeros(RE, dtype=torch.int32, device=dev)       # the kernel writes both halves under guidance
    feed = torch.zeros(RE, dtype=torch.int32, device=dev)      # ... and the token every row is fed next
    flags = torch.zeros((G, 3), dtype=torch.int32, device=dev)
    flags_host = torch.zeros((G, 3), dtype=torch.int32).pin_memory()
    need = lib.mh_t5_decode_workspace_bytes(C.byref(p.cfg), RE)
    ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
    scratch = torch.empty(int(lib.mh_t5_reorder_cache_scratch_bytes(C.byref(p.cfg), RE, max_length)), dtype=torch.uint8, device=dev)
    logits = torch.empty((RE, V), dtype=torch.float32, device=dev)
    stream = engine._s()
    bs = _lib.MhBeamStep()
    bs.logits, bs.eos_table = logits.data_ptr(), eos_table.data_ptr()
    bs.G, bs.num_beams, bs.V, bs.P, bs.max_length, bs.K = G, nb, V, P, max_length, K
    bs.cfg, bs.cfg_scale, bs.length_penalty = int(cfg), float(sp.cfg_scale), float(length_penalty)
    bs.early_stopping = 2 if early_stopping == "never" else (1 if early_stopping is True else 0)
    bs.sp = sp
    bs.heuristic_open, bs.src, bs.last, bs.flags = heuristic_open.data_ptr(), src.data_ptr(), feed.data_ptr(), flags.data_ptr()

    def step(tokens: torch.Tensor, pos: int):
        rc = lib.mh_t5_step(C.byref(p.cfg), C.byref(p.w), cross_kv.data_ptr(), RE, nb, tokens.data_ptr(), pos, _lib.ptr(mask), P,
                            logits.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        _lib.check(rc, "mh_t5_step")

    engine._enter()
    with torch.cuda.stream(engine.stream):
        for pos in range(P - 1):
            step(ids0e[:, pos].contiguous(), pos)
        feed.copy_(ids0e[:, P - 1])                    # the last prompt column (under guidance the first half has the negative prompt's)
        cur_len, par = P, 0
        while True:
            step(feed, cur_len - 1)
            a, b = st[par], st[par ^ 1]
            bs.cur_len = cur_len
            bs.run_in, bs.rs_in, bs.rb_in = a["run"].data_ptr(), a["rs"].data_ptr(), a["rb"].data_ptr()
            bs.seq_in, bs.bs_in, bs.bb_in, bs.fin_in = a["seq"].data_ptr(), a["bs"].data_ptr(), a["bb"].data_ptr(), a["fin"].data_ptr()
            bs.run_out, bs.rs_out, bs.rb_out = b["run"].data_ptr(), b["rs"].data_ptr(), b["rb"].data_ptr()
            bs.seq_out, bs.bs_out, bs.bb_out, bs.fin_out = b["seq"].data_ptr(), b["bs"].data_ptr(), b["bb"].data_ptr(), b["fin"].data_ptr()
            _lib.check(lib.mh_beam_step(C.byref(bs), stream), "mh_beam_step")
            par ^= 1
            # (`src` / `feed` were written by the kernel for all RE rows: under guidance `beam_idx.repeat(2)`, cache_utils.py:18, and the
            # beam's token for both halves)
            rc = lib.mh_t5_reorder_cache(C.byref(p.cfg), RE, src.data_ptr(), cur_len, ws.data_ptr(), ws.numel(),
                                         scratch.data_ptr(), scratch.numel(), stream)
            _lib.check(rc, "mh_t5_reorder_cache")
            cur_len += 1
            flags_host.copy_(flags, non_blocking=True)
            engine.stream.synchronize()
            f = flags_host.numpy()
            go_on = bool(f[:, 0].any()) and not (bool(f[:, 2].all()) and early_stopping is True) and not bool(f[:, 1].all())
            if not go_on:
                break
        fin = st[par]
        best = fin["seq"][:, 0, :].to(torch.int64)
        n_gen = int(((fin["bb"][:, 0, :] + 1).bool()).sum(dim=1).max())
        out = best[:, :P + n_gen].clone()
    engine._leave()
    engine.synchronize()
    return out


@torch.no_grad()
def beam_search(engine, cross_kv: torch.Tensor, prompt: torch.Tensor, prompt_mask: Optional[torch.Tensor], eos_ids, sp,
                num_beams: int, length_penalty: float = 1.0, early_stopping=False, sample_fn=None,
                use_kernel: Optional[bool] = None) -> torch.Tensor:
    """cross_kv: the G chunks' cross K/V (engine.cross_kv); prompt int (G, P) left-padded, prompt_mask (G, P) or None.
    Under guidance (sp.cfg_scale > 1) `prompt` / `prompt_mask` carry 2G rows, [negative-prompt rows | prompt rows] (what
    T5Engine.generate and the scheduler build), and cross_kv still has G rows.
    Returns int64 (G, P + new) on the engine's device: the best hypothesis per chunk, shorter ones filled the way HF does
    (`pad_token_id or eos_token_id[0]`: with pad id 0 that is the FIRST EOS id).
    `use_kernel`: None = mh_beam_step whenever it covers the call (greedy beams, see kernel_path_available), False = the torch-op
    bookkeeping below, True = the kernel or an error."""
    can = kernel_path_available(sp, int(num_beams), engine.packed.vocab_out, len(list(eos_ids))) and sample_fn is None
    if use_kernel is True and not can:
        raise NotImplementedError("mh_beam_step does not cover this call (beam-sample, > 8 beams, beams x V > 16384 or K > 4096)")
    if can and use_kernel is not False:
        return _beam_search_kernel(engine, cross_kv, prompt, prompt_mask, eos_ids, sp, num_beams, length_penalty, early_stopping)
    dev, lib, p = engine.device, engine.lib, engine.packed
    cfg = sp.cfg_scale > 1.0
    neg_prompt = None
    if cfg:
        if prompt.shape[0] % 2:
            raise ValueError("guidance: the prompt batch must be [negative rows | prompt rows]")
        half = prompt.shape[0] // 2
        neg_prompt, prompt = prompt[:half], prompt[half:]
        neg_mask = None if prompt_mask is None else prompt_mask[:half]
        prompt_mask = None if prompt_mask is None else prompt_mask[half:]
        cross_kv = torch.cat([cross_kv, cross_kv], dim=2)          # [layer][k|v][chunk][H][L][64]: the negative rows read their chunk's K / V
    G, P = prompt.shape
    nb = int(num_beams)
    R = G * nb                                                     # rows the beam bookkeeping ranks
    RE = 2 * R if cfg else R                                       # rows the engine decodes
    if RE > 64:
        raise ValueError(f"{G} chunks x {nb} beams{' x 2 (guidance)' if cfg else ''} exceed the engine's 64-row decode batch")
    V = p.vocab_out
    max_length = int(sp.max_length)
    if not (1 <= P < max_length <= p.tgt_len):
        raise ValueError("prompt / max_length do not fit the cache")
    eos_list = [int(e) for e in eos_ids]
    procs = BeamProcessors(sp, dev, min_tokens_to_keep=len(eos_list) + 1 if eos_list else 2)
    do_sample = bool(sp.do_sample)
    if do_sample and sample_fn is None:
        sample_fn = torch.multinomial
    eos_t = torch.tensor(sorted(set(eos_list)), dtype=torch.long, device=dev)
    n_eos = len(eos_list)
    K = max(2, 1 + n_eos) * nb
    K = min(K, nb * V)
    top_mask = torch.zeros(K, dtype=torch.bool, device=dev)
    top_mask[:nb] = True
    fill = int(sp.pad_id) or (eos_list[0] if eos_list else -1)
    ids0 = prompt.to(dev, torch.int64).repeat_interleave(nb, 0)                       # (R, P): rows (chunk, beam)
    mask = None if prompt_mask is None else prompt_mask.to(dev).to(torch.uint8).repeat_interleave(nb, 0).contiguous()
    ids0e = ids0                                                                      # what the engine is fed over the prompt columns
    if cfg:
        ids0e = torch.cat([neg_prompt.to(dev, torch.int64).repeat_interleave(nb, 0), ids0], 0)
        if mask is not None:
            mask = torch.cat([neg_mask.to(dev).to(torch.uint8).repeat_interleave(nb, 0), mask], 0).contiguous()
    running = torch.full((G, nb, max_length), fill, dtype=torch.int64, device=dev)
    running[:, :, :P] = ids0.view(G, nb, P)
    sequences = running.clone()
    running_scores = torch.zeros((G, nb), dtype=torch.float32, device=dev)
    running_scores[:, 1:] = -1e9
    beam_scores = torch.full((G, nb), -1e9, dtype=torch.float32, device=dev)
    finished = torch.zeros((G, nb), dtype=torch.bool, device=dev)
    heuristic_open = torch.ones((G, 1), dtype=torch.bool, device=dev)
    n_new = max_length - P
    running_bidx = torch.full((G, nb, n_new), -1, dtype=torch.int32, device=dev)
    beam_bidx = running_bidx.clone()

    need = lib.mh_t5_decode_workspace_bytes(C.byref(p.cfg), RE)
    ws = torch.empty(int(need), dtype=torch.uint8, device=dev)                        # owned by this call: holds the caches
    scratch = torch.empty(int(lib.mh_t5_reorder_cache_scratch_bytes(C.byref(p.cfg), RE, max_length)), dtype=torch.uint8, device=dev)
    logits = torch.empty((RE, V), dtype=torch.float32, device=dev)
    stream = engine._s()
    scale = float(sp.cfg_scale)

    def step(tokens: torch.Tensor, pos: int):
        t32 = tokens.to(torch.int32).contiguous()
        rc = lib.mh_t5_step(C.byref(p.cfg), C.byref(p.w), cross_kv.data_ptr(), RE, nb, t32.data_ptr(), pos, _lib.ptr(mask), P,
                            logits.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        _lib.check(rc, "mh_t5_step")

    engine._enter()
    with torch.cuda.stream(engine.stream):
        for pos in range(P - 1):                                                      # the prompt, token by token
            step(ids0e[:, pos], pos)
        cur_len = P
        first = True
        while True:
            flat = running[:, :, :cur_len].reshape(R, cur_len)
            last = flat[:, cur_len - 1]
            if cfg:      # both halves are fed the beam's last token; over the prompt columns the first half has the negative prompt's
                last = torch.cat([last if cur_len > P else ids0e[:R, P - 1], last], 0)
            step(last, cur_len - 1)
            lsm = torch.log_softmax(logits, dim=-1)
            if cfg:      # HF's processor, first in the list: `cond, uncond = scores.split(R); uncond + (cond - uncond) * scale`
                lsm = lsm[R:] + (lsm[:R] - lsm[R:]) * scale
            log_probs = procs(flat, lsm)
            acc = (log_probs.view(G, nb, V) + running_scores[:, :, None]).reshape(G, nb * V)
            if do_sample:     # K continuations drawn without replacement, kept in draw order
                topk_idx = sample_fn(torch.softmax(acc, dim=-1), K).to(dev)
                topk_lp = torch.gather(acc, 1, topk_idx)
            else:
                topk_lp, topk_idx = torch.topk(acc, k=K)
            src_beam = topk_idx // V
            topk_bidx = _gather(running_bidx, src_beam)
            topk_seq = _gather(running, src_beam)
            topk_seq[:, :, cur_len] = topk_idx % V
            topk_bidx[:, :, cur_len - P] = (src_beam + torch.arange(G, device=dev).view(-1, 1) * nb).to(torch.int32)
            # d. stopping criteria on the K candidates: EOS id as the new last token, or max_length reached
            hits = torch.isin(topk_seq[:, :, cur_len], eos_t) | (cur_len + 1 >= max_length)
            # e. the running beams of the next step: best num_beams candidates that did not just finish
            run_lp = topk_lp + hits.to(torch.float32) * -1.0e9
            nxt = torch.topk(run_lp, k=nb)[1]
            running, running_scores, running_bidx = _gather(topk_seq, nxt), _gather(run_lp, nxt), _gather(topk_bidx, nxt)
            # f. finished hypotheses: only candidates inside the top num_beams count
            just = hits & top_mask[None, :]
            fin_lp = topk_lp / ((cur_len + 1 - P) ** length_penalty)
            full = torch.all(finished, dim=-1, keepdim=True) & (early_stopping is True)
            fin_lp = fin_lp + full.to(torch.float32) * -1.0e9
            fin_lp = fin_lp + (~heuristic_open).to(torch.float32) * -1.0e9
            fin_lp = fin_lp + (~just) * -1.0e9
            m_seq = torch.cat((sequences, topk_seq), dim=1)
            m_sc = torch.cat((beam_scores, fin_lp), dim=1)
            m_bi = torch.cat((beam_bidx, topk_bidx), dim=1)
            m_fin = torch.cat((finished, just), dim=1)
            sel = torch.topk(m_sc, k=nb)[1]
            sequences, beam_scores, beam_bidx, finished = _gather(m_seq, sel), _gather(m_sc, sel), _gather(m_bi, sel), _gather(m_fin, sel)
            # g. the caches follow the beams that keep running
            src = running_bidx[..., cur_len - P].reshape(R).to(torch.int32).contiguous()
            if cfg:      # `beam_idx.repeat(2)` (cache_utils.py:18): BOTH halves take their rows from the first half
                src = torch.cat([src, src], 0).contiguous()
            rc = lib.mh_t5_reorder_cache(C.byref(p.cfg), RE, src.data_ptr(), cur_len, ws.data_ptr(), ws.numel(), scratch.data_ptr(),
                                         scratch.numel(), stream)
            _lib.check(rc, "mh_t5_reorder_cache")
            cur_len += 1
            if early_stopping == "never" and length_penalty > 0.0:
                hyp_len = max_length - P
            else:
                hyp_len = cur_len - P
            best_running = running_scores[:, :1] / (hyp_len ** length_penalty)
            worst_fin = torch.where(finished, torch.min(beam_scores, dim=1, keepdim=True)[0], -1.0e9)
            heuristic_open = heuristic_open & torch.any(best_running > worst_fin, dim=-1, keepdim=True)
            go_on = torch.any(heuristic_open) & ~(torch.all(finished) & (early_stopping is True)) & ~torch.all(hits)
            first = False
            if not bool(go_on):
                break
        best = sequences[:, 0, :]
        n_gen = int(((beam_bidx[:, 0, :] + 1).bool()).sum(dim=1).max())
        out = best[:, :P + n_gen].clone()
    engine._leave()
    engine.synchronize()
    return out
