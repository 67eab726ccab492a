"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch-CPU fp32 tensor ops) of the osuT5 hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module;
it is the checker, never the product.  It follows, function by function:

  rms_norm            HF T5LayerNorm (transformers 4.57.3 models/t5/modeling_t5.py, "T5LayerNorm");
                      restated in the reference at osuT5/osuT5/model/custom_transformers/t5.py:50-62
  bucket / bias       custom_transformers/t5.py:88-141 (`_relative_position_bucket`), :143-168 (`compute_bias`)
  attention           custom_transformers/t5.py:170-250 (no 1/sqrt(d) scaling, additive bias, fp32 softmax)
  gated FFN           HF T5DenseGatedActDense with gelu_new; custom_transformers/t5.py:50-62 region
  encoder / decoder   custom_transformers/t5.py:305-355 (block), :358-469 (stack); bias shared from layer 0
  wrapper             osuT5/osuT5/model/modeling_mapperatorinator.py:174-207 (mel -> encoder_embedder,
                      decoder_embedder, lm_head without d^-0.5 rescale since tie_word_embeddings=False)
  greedy loop         HF GenerationMixin._sample as driven by osuT5/osuT5/inference/server.py:83-156
                      (SURVEY.md Appendix A), StaticCache semantics of inference/cache_utils.py:23-35
  processors          osuT5/osuT5/inference/logit_processors.py:36-44 (TimeshiftBias), :136-183 (MonotonicTimeShift),
                      :111-114 (LookbackBias, types_first=False), :116-133 (types_first=True),
                      :47-82 (ConditionalTemperature), HF TemperatureLogitsWarper,
                      HF ClassifierFreeGuidanceLogitsProcessor on the batch layout of
                      modeling_mapperatorinator.py:243-254

PINNING (tests/test_oracle_pinned.py, runs where /root/reference exists): hidden states, logits and
greedy ids agree with the imported reference (`Mapperatorinator` + HF T5 via the reference's own
`model_generate`) on seeded weights; the resulting golden vectors are committed under tests/golden/.

`rounding`: None = pure fp32; "bf16" = the storage contract of mapperatorinator_amd/t5_engine.py
(weights and GEMM operands rounded to bf16, everything else fp32).
"""
from __future__ import annotations

import math

import torch

from . import mel as omel


def _bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class Rounding:
    def __init__(self, mode=None):
        assert mode in (None, "bf16")
        self.mode = mode

    def __call__(self, x):  # activation that becomes a GEMM operand
        return _bf16(x) if self.mode == "bf16" else x

    w = __call__  # parameters


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def rms_norm(x, w, eps):
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def bucket(rel, bidirectional, num_buckets=32, max_distance=128):
    ret = torch.zeros_like(rel)
    n = num_buckets
    if bidirectional:
        n //= 2
        ret = ret + (rel > 0).to(torch.long) * n
        rp = torch.abs(rel)
    else:
        rp = -torch.min(rel, torch.zeros_like(rel))
    max_exact = n // 2
    is_small = rp < max_exact
    big = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact)
                       * (n - max_exact)).to(torch.long)
    big = torch.min(big, torch.full_like(big, n - 1))
    return ret + torch.where(is_small, rp, big)


class T5Oracle:
    """Stateless math over a reference-named state_dict (see PackedT5 for the same key names)."""

    def __init__(self, sd: dict, d_model, d_ff, n_heads, n_enc, n_dec, eps=1e-6, n_buckets=32, max_distance=128,
                 rounding=None, enc_mx8=False):
        """`enc_mx8` (with rounding="bf16"): the contract of MhT5Config.enc_operand_dtype = MH_MX8 -- the encoder blocks' four
        projections and the cross-K/V projection multiply the MX-fp8 images (oracle/mx8.py) of their bf16 operands; everything
        else as the bf16 contract.  (float32 matmul of the dequantised operands: the matrix core's own 13-bit alignment window,
        profiles/r04_micro_mx8_precision.txt, is far below the quantisation step and is not modelled.)"""
        assert not enc_mx8 or rounding == "bf16"
        self.enc_mx8 = enc_mx8
        self.r = Rounding(rounding)
        self.sd = {k: self.r.w(v.detach().to(torch.float32)) for k, v in sd.items()
                   if v.dtype.is_floating_point}
        self.d, self.dff, self.H, self.ne, self.nd, self.eps = d_model, d_ff, n_heads, n_enc, n_dec, eps
        self.nb, self.md = n_buckets, max_distance

    # ---- pieces ----------------------------------------------------------------------------
    def _heads(self, x):  # (B, T, H*64) -> (B, H, T, 64)
        B, T, _ = x.shape
        return x.view(B, T, self.H, 64).transpose(1, 2)

    def _attn(self, q, k, v, bias, mask=None):
        r = self.r
        scores = torch.matmul(r(q), r(k).transpose(-1, -2))
        if bias is not None:
            scores = scores + bias
        if mask is not None:
            # HF adds finfo.min (not -inf): a fully masked (left-pad) query row stays finite
            scores = scores.masked_fill(~mask, torch.finfo(torch.float32).min)
        p = torch.softmax(scores, dim=-1)
        out = torch.matmul(r(p), r(v))
        B, H, T, _ = out.shape
        return out.transpose(1, 2).reshape(B, T, H * 64)

    def _ffn(self, x, pre, lin=None):
        r, sd = self.r, self.sd
        lin = lin or (lambda a, name: a @ sd[name].t())
        g = lin(x, pre + "wi_0.weight")
        u = lin(x, pre + "wi_1.weight")
        return lin(r(gelu_new(g) * u), pre + "wo.weight")

    def _enc_lin(self, a, name):
        """a projection of the encoder side: plain, or on the MX-fp8 images of both operands"""
        if not self.enc_mx8:
            return a @ self.sd[name].t()
        from .mx8 import fake_quant_torch
        cache = self.__dict__.setdefault("_mxw", {})
        if name not in cache:
            cache[name] = fake_quant_torch(self.sd[name])
        return fake_quant_torch(a) @ cache[name].t()

    def enc_bias(self, L):
        pos = torch.arange(L)
        rel = pos[None, :] - pos[:, None]
        tab = self.sd["transformer.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
        return tab[bucket(rel, True, self.nb, self.md)].permute(2, 0, 1)[None]  # (1,H,L,L)

    def dec_bias(self, q_pos, klen):
        rel = torch.arange(klen)[None, :] - torch.as_tensor(q_pos).reshape(-1, 1)
        tab = self.sd["transformer.decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
        return tab[bucket(rel, False, self.nb, self.md)].permute(2, 0, 1)[None]  # (1,H,q,klen)

    # ---- encoder ---------------------------------------------------------------------------
    def embed_mel(self, mel, cond=None):
        """`cond` (B, cond_size): the per-row conditioning vectors, repeated over the frames and concatenated to the mel
        frames in front of encoder_embedder (modeling_mapperatorinator.py:411-414)."""
        r, sd = self.r, self.sd
        if cond is not None:
            mel = torch.cat([mel, cond.to(mel.dtype).unsqueeze(1).expand(-1, mel.shape[1], -1)], -1)
        return r(mel) @ sd["encoder_embedder.weight"].t() + sd["encoder_embedder.bias"]

    def encoder(self, h):
        r, sd = self.r, self.sd
        B, L, _ = h.shape
        bias = self.enc_bias(L)
        for l in range(self.ne):
            b = f"transformer.encoder.block.{l}."
            a = b + "layer.0.SelfAttention."
            lin = self._enc_lin
            n = r(rms_norm(h, sd[b + "layer.0.layer_norm.weight"], self.eps))
            q = self._heads(r(lin(n, a + "q.weight")))
            k = self._heads(r(lin(n, a + "k.weight")))
            v = self._heads(r(lin(n, a + "v.weight")))
            h = h + lin(r(self._attn(q, k, v, bias)), a + "o.weight")
            n = r(rms_norm(h, sd[b + "layer.1.layer_norm.weight"], self.eps))
            h = h + self._ffn(n, b + "layer.1.DenseReluDense.", lin)
        return rms_norm(h, sd["transformer.encoder.final_layer_norm.weight"], self.eps)

    def encode_audio(self, audio, n_mels=388, cond=None):
        mel = omel.mel_spectrogram(audio, n_mels=n_mels)
        return self.encoder(self.embed_mel(mel, cond))

    # ---- decoder ---------------------------------------------------------------------------
    def cross_kv(self, enc):
        r, sd = self.r, self.sd
        e = r(enc)
        out = []
        for l in range(self.nd):
            x = f"transformer.decoder.block.{l}.layer.1.EncDecAttention."
            out.append((self._heads(r(self._enc_lin(e, x + "k.weight"))), self._heads(r(self._enc_lin(e, x + "v.weight")))))
        return out

    def decoder_step(self, tok, pos, cache, ckv, key_mask):
        """tok (B,) ids fed at position `pos`; cache: list of (K,V) (B,H,Tmax,64) updated in place;
        key_mask (B, Tmax) bool (True = attend) for positions <= pos.  Returns fp32 logits (B, V)."""
        r, sd = self.r, self.sd
        h = sd["decoder_embedder.weight"][tok][:, None, :]
        bias = self.dec_bias([pos], pos + 1)
        m = key_mask[:, None, None, :pos + 1]
        for l in range(self.nd):
            b = f"transformer.decoder.block.{l}."
            a = b + "layer.0.SelfAttention."
            x = b + "layer.1.EncDecAttention."
            n = r(rms_norm(h, sd[b + "layer.0.layer_norm.weight"], self.eps))
            q = self._heads(r(n @ sd[a + "q.weight"].t()))
            K, V = cache[l]
            K[:, :, pos] = self._heads(r(n @ sd[a + "k.weight"].t()))[:, :, 0]
            V[:, :, pos] = self._heads(r(n @ sd[a + "v.weight"].t()))[:, :, 0]
            h = h + r(self._attn(q, K[:, :, :pos + 1], V[:, :, :pos + 1], bias, m)) @ sd[a + "o.weight"].t()
            n = r(rms_norm(h, sd[b + "layer.1.layer_norm.weight"], self.eps))
            q = self._heads(r(n @ sd[x + "q.weight"].t()))
            h = h + r(self._attn(q, ckv[l][0], ckv[l][1], None)) @ sd[x + "o.weight"].t()
            n = r(rms_norm(h, sd[b + "layer.2.layer_norm.weight"], self.eps))
            h = h + self._ffn(n, b + "layer.2.DenseReluDense.")
        n = r(rms_norm(h, sd["transformer.decoder.final_layer_norm.weight"], self.eps))
        return (n @ sd["transformer.lm_head.weight"].t())[:, 0, :]

    def decoder_forward(self, ids, ckv, key_mask=None):
        """Teacher-forced logits of EVERY position at once: ids (B, T) -> fp32 (B, T, V).  Same arithmetic and the same
        rounding points as T calls of `decoder_step` (causal mask instead of a growing cache); exists because the
        step-by-step form needs ~1 s per step at B = 32 on a CPU.  key_mask (B, T) bool, True = attend."""
        r, sd = self.r, self.sd
        B, T = ids.shape
        h = sd["decoder_embedder.weight"][ids]
        bias = self.dec_bias(list(range(T)), T)                       # (1, H, T, T)
        m = torch.ones(T, T, dtype=torch.bool).tril()[None, None]
        if key_mask is not None:
            m = m & key_mask[:, None, None, :T]
        for l in range(self.nd):
            b = f"transformer.decoder.block.{l}."
            a = b + "layer.0.SelfAttention."
            x = b + "layer.1.EncDecAttention."
            n = r(rms_norm(h, sd[b + "layer.0.layer_norm.weight"], self.eps))
            q = self._heads(r(n @ sd[a + "q.weight"].t()))
            k = self._heads(r(n @ sd[a + "k.weight"].t()))
            v = self._heads(r(n @ sd[a + "v.weight"].t()))
            h = h + r(self._attn(q, k, v, bias, m)) @ sd[a + "o.weight"].t()
            n = r(rms_norm(h, sd[b + "layer.1.layer_norm.weight"], self.eps))
            q = self._heads(r(n @ sd[x + "q.weight"].t()))
            h = h + r(self._attn(q, ckv[l][0], ckv[l][1], None)) @ sd[x + "o.weight"].t()
            n = r(rms_norm(h, sd[b + "layer.2.layer_norm.weight"], self.eps))
            h = h + self._ffn(n, b + "layer.2.DenseReluDense.")
        n = r(rms_norm(h, sd["transformer.decoder.final_layer_norm.weight"], self.eps))
        return n @ sd["transformer.lm_head.weight"].t()

    def monotonic_scores(self, logits, ids, ts_start, ts_end, sos_ids):
        """MonotonicTimeShiftLogitsProcessor (logit_processors.py:136-183) applied to teacher-forced logits: row t of
        `logits` (B, T, V) is masked with the history ids[:, :t+1].  Returns a new tensor."""
        B, T, _ = logits.shape
        out = logits.clone().float()
        last_val = torch.full((B,), -1, dtype=torch.long)
        sos = torch.as_tensor(list(sos_ids), dtype=torch.long)
        col = torch.arange(ts_start, ts_end)
        for t in range(T):
            tok = ids[:, t]
            is_ts = (tok >= ts_start) & (tok < ts_end)
            last_val = torch.where(is_ts, tok - ts_start, torch.where(torch.isin(tok, sos), torch.full_like(tok, -1), last_val))
            bad = (last_val >= 0)[:, None] & (col[None, :] < (ts_start + last_val)[:, None])
            out[:, t, ts_start:ts_end][bad] = float("-inf")
        return out

    # ---- generation ------------------------------------------------------------------------
    def generate(self, enc, prompt, prompt_mask, eos_ids, max_length, ts_start, ts_end, sos_ids, pad_id=0,
                 temperature=1.0, timeshift_bias=0.0, lookback_mask_end=0, forced=None, return_logits=False,
                 negative_prompt=None, negative_mask=None, cfg_scale=1.0, cond_rules=(), lookback_types_first=None):
        """Greedy decode with HF `_sample` bookkeeping.  prompt (B,P) int64 left-padded, prompt_mask bool.
        Returns ids (B, n_cols) [, processed scores per produced column (list of (B,V))].

        negative_prompt/negative_mask/cfg_scale: classifier-free guidance with the reference's batch layout
            (modeling_mapperatorinator.py:243-254 + HF ClassifierFreeGuidanceLogitsProcessor): rows [0,B) of the
            doubled batch carry the negative prompt, rows [B,2B) the prompt.
        cond_rules: [(temperature, token ids, offset)] of ConditionalTemperatureLogitsWarper (logit_processors.py:47-82)
        lookback_types_first: dict(eos_ids=[...], timed_ids=[...]) switches LookbackBiasLogitsWarper to its
            types_first=True branch (:116-133) over [ts_start, lookback_mask_end)."""
        B, P = prompt.shape
        cfg = negative_prompt is not None and cfg_scale > 1.0
        NB = 2 * B if cfg else B
        ckv = self.cross_kv(enc.repeat(2, 1, 1) if cfg else enc)
        cache = [(torch.zeros(NB, self.H, max_length, 64), torch.zeros(NB, self.H, max_length, 64))
                 for _ in range(self.nd)]
        key_mask = torch.ones(NB, max_length, dtype=torch.bool)
        key_mask[NB - B:, :P] = prompt_mask.bool() if prompt_mask is not None else True
        if cfg:
            # `negative_prompt_attention_mask` never reaches modeling_mapperatorinator.py:249-250: it is a named
            # parameter of HF `GenerationMixin.generate` (unbatched-CFG support) and is consumed there, so both
            # halves of the doubled batch run under the PROMPT's mask (observed on the imported reference; pinned
            # by tests/golden/t5_tiny_tf.npz whose negative prompts are padded differently from the prompts)
            key_mask[:B, :P] = key_mask[B:, :P]
        ids = prompt.clone()
        unfinished = torch.ones(B, dtype=torch.bool)
        eos = torch.as_tensor(sorted(set(eos_ids)), dtype=torch.long)
        sos = torch.as_tensor(list(sos_ids), dtype=torch.long)
        all_scores = []
        last_scores = None
        max_offset = max([o for _, _, o in cond_rules], default=0)
        for pos in range(0, max_length - 1):
            feed = ids[:, pos] if forced is None or pos < P else forced[:, pos]
            if cfg:
                neg_feed = negative_prompt[:, pos] if pos < negative_prompt.shape[1] else feed
                feed = torch.cat([neg_feed, feed])
            logits = self.decoder_step(feed, pos, cache, ckv, key_mask)
            if pos + 1 < P:
                continue
            hist = ids if forced is None else torch.cat([prompt, forced[:, P:pos + 1]], 1)
            scores = logits.clone().float()
            if cfg:   # HF: first half = "cond", second half = "uncond"
                cond_l, uncond_l = scores[:B], scores[B:]
                scores = uncond_l + (cond_l - uncond_l) * cfg_scale
            # MonotonicTimeShiftLogitsProcessor
            idx = torch.arange(hist.shape[1]).expand(B, -1)
            is_ts = (hist >= ts_start) & (hist < ts_end)
            is_sos = torch.isin(hist, sos)
            last_ts = torch.where(is_ts, idx, -1).max(1).values
            last_sos = torch.where(is_sos, idx, -1).max(1).values
            val = torch.where(last_ts != -1, hist[torch.arange(B), last_ts.clamp(min=0)] - ts_start, 0)
            apply = (last_ts != -1) & (last_ts > last_sos)
            vocab = torch.arange(ts_start, ts_end)
            bad = vocab[None, :] < (ts_start + val)[:, None]
            sl = scores[:, ts_start:ts_end]
            sl[apply[:, None] & bad] = float("-inf")
            if timeshift_bias != 0:
                scores[:, ts_start:ts_end] += timeshift_bias
            # (Conditional)TemperatureLogitsWarper: row 0's last ids pick the temperature of the whole batch
            temp = temperature
            if cond_rules:
                lookback = hist[0, -max_offset:]
                for t, toks, off in cond_rules:
                    if len(lookback) >= off and int(lookback[-off]) in toks:
                        temp = t
                        break
            scores = scores / temp
            if lookback_mask_end > ts_start:
                if lookback_types_first is None:
                    scores[:, ts_start:lookback_mask_end] = float("-inf")
                else:
                    entering = scores
                    if last_scores is not None:
                        timed = torch.as_tensor(sorted(lookback_types_first["timed_ids"]), dtype=torch.long)
                        eos_l = torch.as_tensor(list(lookback_types_first["eos_ids"]), dtype=torch.long)
                        last_timed = torch.isin(hist[:, -1], timed)
                        if last_timed.any():
                            in_lb = torch.zeros(scores.shape[1], dtype=torch.bool)
                            in_lb[ts_start:lookback_mask_end] = True
                            last_probs = torch.softmax(last_scores, -1)
                            probs = torch.softmax(scores, -1)
                            prob_eos = last_probs[:, eos_l].sum(-1)
                            prob_event = 1 - prob_eos
                            sc = 1 / (probs[:, ~in_lb].sum(-1) * prob_event + prob_eos)
                            probs[:, in_lb] = 0
                            probs[:, ~in_lb] *= sc[:, None]
                            probs[:, ts_start] = torch.clip((sc - 1) * prob_eos / prob_event, 0, 1)
                            scores = torch.where(last_timed[:, None], torch.log(probs), scores)
                    last_scores = entering
            all_scores.append(scores)
            nxt = scores.argmax(-1)
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad_id))
            ids = torch.cat([ids, nxt[:, None]], 1)
            if forced is None:
                done = torch.isin(nxt, eos) | (ids.shape[1] >= max_length)
                unfinished = unfinished & ~done
                if not unfinished.any():
                    break
        return (ids, all_scores) if return_logits else ids


    # ---- beam search ---------------------------------------------------------------------------------
    def _processed_log_probs(self, logits, hist, ts_start, ts_end, sos_ids, temperature, timeshift_bias, lookback_mask_end,
                             top_k=0, top_p=1.0, min_tokens_to_keep=1, cfg_scale=1.0):
        """HF beam search hands the processors LOG-PROBABILITIES (generation/utils.py `_beam_search` step b).  `cfg_scale > 1`:
        `logits` holds 2R rows [negative-prompt rows | prompt rows]; HF's ClassifierFreeGuidanceLogitsProcessor runs first and takes
        the FIRST half as the conditional one (`uncond + (cond - uncond) * scale`, no renormalisation), R rows go on."""
        scores = torch.log_softmax(logits.float(), dim=-1)
        if cfg_scale > 1.0:
            first, second = scores.split(scores.shape[0] // 2, dim=0)
            scores = second + (first - second) * cfg_scale
        sos = set(int(v) for v in sos_ids)
        for r in range(hist.shape[0]):
            row = hist[r].tolist()
            last_ts = max((i for i, t in enumerate(row) if ts_start <= t < ts_end), default=-1)
            last_sos = max((i for i, t in enumerate(row) if t in sos), default=-1)
            if last_ts != -1 and last_ts > last_sos:
                scores[r, ts_start:ts_start + (row[last_ts] - ts_start)] = float("-inf")
        if timeshift_bias != 0:
            scores[:, ts_start:ts_end] += timeshift_bias
        scores = scores / temperature
        if lookback_mask_end > ts_start:
            scores[:, ts_start:lookback_mask_end] = float("-inf")
        # beam-sample only: HF's TopK / TopP warpers, appended behind the reference's list (generation/utils.py
        # `_get_logits_processor`; logits_process.py TopKLogitsWarper / TopPLogitsWarper), row by row
        for r in range(scores.shape[0]):
            if top_k > 0:
                kth = torch.sort(scores[r], descending=True)[0][min(max(top_k, min_tokens_to_keep), scores.shape[1]) - 1]
                scores[r][scores[r] < kth] = float("-inf")
            if top_p < 1.0:
                asc, order = torch.sort(scores[r], descending=False)
                drop = asc.softmax(dim=-1).cumsum(dim=-1) <= (1 - top_p)
                drop[-min_tokens_to_keep:] = False
                scores[r][order[drop]] = float("-inf")
        return scores

    def generate_beam(self, enc, prompt, prompt_mask, eos_ids, max_length, ts_start, ts_end, sos_ids, num_beams, pad_id=0,
                      temperature=1.0, timeshift_bias=0.0, lookback_mask_end=0, length_penalty=1.0, sample_fn=None, top_k=0, top_p=1.0,
                      negative_prompt=None, cfg_scale=1.0):
        """HF `GenerationMixin._beam_search` (third-party; the vectorised form of transformers >= 4.50, early_stopping =
        False, num_return_sequences = 1) as `model_generate` reaches it with `num_beams > 1` (processor.py:159), restated
        chunk by chunk with explicit candidate lists; the self-attention cache rows are re-gathered per step as
        `MapperatorinatorCache.reorder_cache` does (inference/cache_utils.py:16-20).  Returns ids (B, n_cols): the best
        hypothesis per chunk, shorter rows filled with `pad_token_id or eos_token_id[0]` (HF's `output_fill_value`).
        `sample_fn(probs (B, nb V), K)`: beam-SAMPLE -- the K continuations are drawn without replacement from
        softmax(accumulated) in ONE call for all chunks (`_get_top_k_continuations`), kept in draw order, and the processed
        log-probabilities pass HF's top-k / top-p warpers with `min_tokens_to_keep = #eos + 1` first.
        `negative_prompt` with `cfg_scale > 1`: guidance under beams as the reference + HF run it -- every (chunk, beam) row decoded
        twice, [rows fed the negative prompt over the first prompt columns | the prompt's own rows] (prepare_inputs_for_generation,
        modeling_mapperatorinator.py:243-254), guided log-probabilities for the ranking, and the doubled cache gathered with
        `beam_idx.repeat(2)` (inference/cache_utils.py:16-20): BOTH halves take their rows from the FIRST half."""
        B, P = prompt.shape
        cfg = negative_prompt is not None and cfg_scale > 1.0
        nb, V = int(num_beams), self.sd[[k for k in self.sd if k.endswith("lm_head.weight") or k.endswith("proj_out.weight")][0]].shape[0]
        R = B * nb
        RE = 2 * R if cfg else R                                                   # rows the decoder runs
        ckv = self.cross_kv(enc.repeat_interleave(nb, 0))
        if cfg:
            ckv = [(torch.cat([k, k], 0), torch.cat([v, v], 0)) for k, v in ckv]
        cache = [(torch.zeros(RE, self.H, max_length, 64), torch.zeros(RE, self.H, max_length, 64)) for _ in range(self.nd)]
        key_mask = torch.ones(RE, max_length, dtype=torch.bool)
        if prompt_mask is not None:      # (the negative rows attend under the prompt's mask: HF swallows negative_prompt_attention_mask)
            key_mask[:, :P] = prompt_mask.bool().repeat_interleave(nb, 0).repeat(2 if cfg else 1, 1)
        fed_prompt = prompt.repeat_interleave(nb, 0)                               # what the decoder is fed over the prompt columns
        if cfg:
            neg = prompt.clone()
            neg[:, :negative_prompt.shape[1]] = negative_prompt
            fed_prompt = torch.cat([neg.repeat_interleave(nb, 0), fed_prompt], 0)
        eos_list = [int(e) for e in eos_ids]
        eos = set(eos_list)
        K = max(2, 1 + len(eos_list)) * nb
        fill = pad_id or (eos_list[0] if eos_list else -1)
        f32 = lambda v: torch.tensor(v, dtype=torch.float32)
        run_seq = [[prompt[b].tolist() for _ in range(nb)] for b in range(B)]
        run_score = [[f32(0.0)] + [f32(-1e9)] * (nb - 1) for _ in range(B)]
        run_hist = [[[] for _ in range(nb)] for _ in range(B)]                     # beam indices per generated position
        fin = [[(f32(-1e9), None, [], False) for _ in range(nb)] for _ in range(B)]  # (score, seq, beam-index history, finished)
        open_h = [True] * B
        for pos in range(P - 1):
            self.decoder_step(fed_prompt[:, pos], pos, cache, ckv, key_mask)
        cur = P
        while True:
            flat = torch.tensor([run_seq[b][k] for b in range(B) for k in range(nb)])
            last = flat[:, cur - 1]
            if cfg:      # both halves are fed the beam's last token; at the last prompt column the first half still has the negative prompt's
                last = torch.cat([last if cur > P else fed_prompt[:R, P - 1], last], 0)
            logits = self.decoder_step(last, cur - 1, cache, ckv, key_mask)
            lp = self._processed_log_probs(logits, flat, ts_start, ts_end, sos_ids, temperature, timeshift_bias, lookback_mask_end,
                                           cfg_scale=cfg_scale if cfg else 1.0,
                                           **(dict(top_k=top_k, top_p=top_p, min_tokens_to_keep=(len(eos_list) + 1) if eos_list else 2)
                                              if sample_fn is not None else {}))
            src_rows = []
            all_hit = True
            accs = [torch.stack([lp[b * nb + k] + run_score[b][k] for k in range(nb)]).reshape(-1) for b in range(B)]
            drawn = None if sample_fn is None else sample_fn(torch.softmax(torch.stack(accs), dim=-1), min(K, accs[0].numel()))
            for b in range(B):
                acc = accs[b]
                if drawn is None:
                    top_lp, top_idx = torch.topk(acc, k=min(K, acc.numel()))
                else:
                    top_idx = drawn[b]
                    top_lp = acc[top_idx]
                cand = []
                for lpv, ix in zip(top_lp, top_idx.tolist()):
                    k0, tok = ix // V, ix % V
                    seq = run_seq[b][k0] + [tok]
                    hit = (tok in eos) or (len(seq) >= max_length)
                    cand.append((lpv, seq, run_hist[b][k0] + [b * nb + k0], hit))
                all_hit = all_hit and all(c[3] for c in cand)
                # e. running beams: the best nb candidates after pushing the just-finished ones down by 1e9
                run_lp = torch.stack([c[0] + (f32(-1e9) if c[3] else f32(0.0)) for c in cand])
                nxt = torch.topk(run_lp, k=nb)[1].tolist()
                new_seq, new_score, new_hist = [cand[i][1] for i in nxt], [run_lp[i] for i in nxt], [cand[i][2] for i in nxt]
                # f. finished hypotheses, from the top nb candidates only
                all_fin = all(f[3] for f in fin[b])
                merged = list(fin[b])
                for j, c in enumerate(cand):
                    sc = c[0] / f32(float((cur + 1 - P) ** length_penalty))
                    if not open_h[b]:
                        sc = sc + f32(-1e9)
                    just = c[3] and j < nb
                    if not just:
                        sc = sc + f32(-1e9)
                    merged.append((sc, c[1], c[2], just))
                order = torch.topk(torch.stack([m[0] for m in merged]), k=nb)[1].tolist()
                fin[b] = [merged[i] for i in order]
                run_seq[b], run_score[b], run_hist[b] = new_seq, new_score, new_hist
                src_rows += [h[-1] for h in new_hist]
            src = torch.tensor(src_rows)
            if cfg:
                src = src.repeat(2)                                                # `beam_idx.repeat(2)`: indices into the FIRST half, twice
            for l in range(self.nd):                                               # reorder_cache(beam_idx)
                Kc, Vc = cache[l]
                cache[l] = (Kc[src].clone(), Vc[src].clone())
            cur += 1
            any_open = False
            for b in range(B):
                best_running = run_score[b][0] / f32(float((cur - P) ** length_penalty))
                worst = min(f[0] for f in fin[b])
                improv = any(bool(best_running > (worst if f[3] else f32(-1e9))) for f in fin[b])
                open_h[b] = open_h[b] and improv
                any_open = any_open or open_h[b]
            if not any_open or all_hit:
                break
        n_gen = max(len(fin[b][0][2]) for b in range(B))
        out = torch.full((B, P + n_gen), fill, dtype=torch.long)
        for b in range(B):
            seq = fin[b][0][1] if fin[b][0][1] is not None else prompt[b].tolist()
            seq = seq[:P + n_gen]
            out[b, :len(seq)] = torch.tensor(seq)
        return out
