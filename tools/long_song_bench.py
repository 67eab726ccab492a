"""BASELINE configs[4]-shaped measurement: KV-cached decode of whole songs (default: 32 songs x 18 windows of 10 s = 3 minutes
each) through the window scheduler (mapperatorinator_amd/scheduler.py): every window of every song is encoded up front,
its cross-attention K/V stays resident in HBM, wave w decodes window w of all songs as one batch; window w's prompt carries
the last tokens window w-1 produced (the reference's sequential dependency, processor.py:308-368).
Prints one JSON line.  Synthetic audio, random-init weights; greedy decoding; the tokenizer EOS set is active (a window ends when all its rows did).

    python tools/long_song_bench.py [--size large] [--songs 32] [--windows 18] [--new-tokens 384] [--fp8-kv]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mapperatorinator_amd import Tokenizer  # noqa: E402
from mapperatorinator_amd.modeling import MapperatorinatorHIP  # noqa: E402
from mapperatorinator_amd.scheduler import SequentialWindowScheduler, SongJob  # noqa: E402
from mapperatorinator_amd.t5_engine import T5_PRESETS  # noqa: E402
from mh_testing import random_t5_state_dict, synthetic_audio_varied  # noqa: E402


def run(size="large", songs=32, windows=18, new_tokens=384, context_tokens=32, fp8_kv=(False,), device="cuda:0", model=None,
        enc_operand_dtype=None, collect_tokens=False):
    """One dict per entry of `fp8_kv` (the model is built once).  Each carries `roofline_step`: SURVEY 8d bytes of one token
    step (decoder weights + lm_head once, + per song the cross K/V of every layer and the self K/V at the mean position)
    over the measured seconds per token step of the WHOLE run (encode + prompt building + decode), vs the HBM peak.
    enc_operand_dtype="mx8": the encoder blocks and the cross-K/V projection on MX-fp8 operands (BASELINE configs[4]).
    collect_tokens: each dict also carries "_tokens"[song][window] = the window's generated ids (a python list; drop before printing)."""
    dev = torch.device(device)
    src, n_samples = 1251, 160000
    tok = Tokenizer.benchmark_vocab(src_seq_len=src)
    dims = T5_PRESETS[size]
    tgt = 1 + context_tokens + new_tokens
    if model is None:
        model = MapperatorinatorHIP(random_t5_state_dict(dims, tok.vocab_size_in, tok.vocab_size_out, seed=0, lm_head_gain=6.0), dims,
                                    vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out, src_seq_len=src,
                                    tgt_seq_len=tgt, dtype=torch.bfloat16, device=dev, enc_operand_dtype=enc_operand_dtype)
    audio = synthetic_audio_varied(songs * windows, n_samples, seed=3).view(songs, windows, n_samples)
    out = []
    for f8 in fp8_kv:
        gk = dict(max_length=tgt, do_sample=False, cross_kv_fp8=bool(f8))
        last = [None] * songs
        toks = [[None] * windows for _ in range(songs)]
        n_tok = [0]
        cols = [0]

        def make_job(i):
            def prompt_fn(w):
                ctx = last[i][-context_tokens:] if last[i] is not None else torch.zeros(0, dtype=torch.long)
                pad = torch.zeros(context_tokens - ctx.numel(), dtype=torch.long)          # fixed prompt width: one shape per wave
                return dict(decoder_input_ids=torch.cat([pad, torch.tensor([tok.sos_id]), ctx])[None],
                            decoder_attention_mask=torch.cat([pad, torch.ones(1 + ctx.numel(), dtype=torch.long)])[None])

            def on_result(w, row, st):
                last[i] = row[1 + context_tokens:]
                if collect_tokens:
                    toks[i][w] = [int(t) for t in last[i]]
                n_tok[0] += int(row.numel() - 1 - context_tokens)
                if i == 0:
                    cols[0] += int(row.numel() - 1 - context_tokens)       # token steps the wave of window w ran
            return SongJob(frames=audio[i], prompt_fn=prompt_fn, on_result=on_result, generate_kwargs=gk)

        sched = SequentialWindowScheduler(model, tok, encode_batch=32, decode_batch=min(64, songs))
        warm = SequentialWindowScheduler(model, tok, encode_batch=32, decode_batch=min(64, songs))
        warm.run([SongJob(frames=audio[0, :1], prompt_fn=make_job(0).prompt_fn, on_result=lambda *a: None, generate_kwargs=gk)])
        last[:] = [None] * songs
        n_tok[0] = cols[0] = 0
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        stats = sched.run([make_job(i) for i in range(songs)])
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        # roofline of the token step (HBM-bound): what one step of one wave must move / the time the run spent per step
        es, L, d, dff, inner = 2, dims.n_dec_layers, dims.d_model, dims.d_ff, dims.inner
        kv_es = 1 if f8 else 2
        weights = L * ((3 * inner * d + inner * d) + (inner * d + inner * d) + (2 * dff * d + d * dff)) * es + d * tok.vocab_size_out * es
        cross = L * 2 * dims.n_heads * src * 64 * kv_es
        self_kv = L * 2 * dims.n_heads * (1 + context_tokens + new_tokens / 2.0) * 64 * es
        step_bytes = weights + min(64, songs) * (cross + self_kv)
        steps = max(1, cols[0]) * max(1, -(-songs // 64))
        us_per_step = dt / steps * 1e6
        out.append({"workload": f"osuT5-{size} bf16{' (encoder + cross-K/V projection on MX-fp8 operands)' if enc_operand_dtype else ''}, "
                                f"{songs} songs x {windows} windows of 10 s, {new_tokens} new "
                                f"tokens per window, {context_tokens} context tokens, cross K/V {'e4m3' if f8 else 'bf16'}",
                    "event_tokens_per_s": round(n_tok[0] / dt, 1), "seconds": round(dt, 3), "tokens": n_tok[0],
                    "song_seconds_per_s": round(songs * windows * 10.0 / dt, 1),
                    "decode_calls": stats["decode_calls"], "encode_calls": stats["encode_calls"],
                    "resident_cross_kv_gb": round(songs * windows * L * 2 * dims.n_heads * src * 64 * 2 / 1e9, 2),
                    "roofline_step": {"bound": "hbm", "alg_bytes_per_token_step": int(step_bytes), "us_per_token_step": round(us_per_step, 1),
                                      "achieved": round(step_bytes / (us_per_step * 1e-6) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                      "frac": round(step_bytes / (us_per_step * 1e-6) / 1e9 / 8000.0, 4),
                                      "how": "SURVEY 8d bytes of one token step of one wave / (whole-run seconds / token steps of "
                                             "song 0): encode, prompt building and D2H are inside the time"}})
        if collect_tokens:
            out[-1]["_tokens"] = toks
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="large", choices=list(T5_PRESETS))
    ap.add_argument("--songs", type=int, default=32)
    ap.add_argument("--windows", type=int, default=18)
    ap.add_argument("--new-tokens", type=int, default=384)
    ap.add_argument("--context-tokens", type=int, default=32, help="tokens of the previous window carried into the prompt")
    ap.add_argument("--fp8-kv", action="store_true")
    args = ap.parse_args()
    for line in run(args.size, args.songs, args.windows, args.new_tokens, args.context_tokens, (bool(args.fp8_kv),)):
        print(json.dumps(line))


if __name__ == "__main__":
    main()
