"""The reference's DEFAULT call shape (configs/inference/default.yaml:54 `parallel: false`): `generate_sequential`
(osuT5/osuT5/inference/processor.py:308-368) calls `model_generate` with ONE window -- batch 1, or 2 rows under
classifier-free guidance (`prepare_inputs_for_generation` doubles the batch, modeling_mapperatorinator.py:242-253).

Measures the KV-cached decode loop alone (cross K/V resident, events on the engine's stream) for batch 1 and batch 2 (CFG):
microseconds per token step and tokens/s, with the SURVEY 8d roofline of one token step (decoder weights + lm_head once + per row
the cross K/V of every layer and the self K/V at the mean position, over 8 TB/s).  Synthetic audio, random-init weights, the EOS
table zeroed so every row runs to length.  `run()` is what bench.py's aux.decode_b1 / decode_b2_cfg lines call.

    python tools/small_batch_decode.py [--model t5-base|varwhisper-small|whisper-small|ropewhisper-small] [--new-tokens 256]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBS = 8000.0


def build(model_name: str, tgt: int, device, dtype=torch.bfloat16, options=None):
    from mapperatorinator_amd import Tokenizer
    from mapperatorinator_amd.modeling import MapperatorinatorHIP
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mh_testing import random_t5_state_dict, random_varwhisper_state_dict, random_whisper_family_state_dict
    from mapperatorinator_amd.whisper_engine import VARWHISPER_PRESETS
    fam, size = model_name.split("-")
    if fam == "t5":
        frames, n_mels = 1251, 388
        tok = Tokenizer.benchmark_vocab(src_seq_len=frames)
        dims = T5_PRESETS[size]
        sd = random_t5_state_dict(dims, tok.vocab_size_in, tok.vocab_size_out, seed=0, lm_head_gain=6.0)
        kw = {}
    else:
        # the released chunk sizes: V32 varwhisper 2048 frames, V29 whisper 1024, V30 ropewhisper 4096 (configs/train/v*.yaml)
        frames = {"varwhisper": 2048, "whisper": 1024, "ropewhisper": 4096}[fam]
        n_mels = {"varwhisper": 128, "whisper": 388, "ropewhisper": 80}[fam]
        tok = Tokenizer.benchmark_vocab(src_seq_len=frames)
        dims = VARWHISPER_PRESETS[size]
        if fam == "varwhisper":
            sd = random_varwhisper_state_dict(dims.d_model, dims.n_heads, dims.n_enc_layers, dims.n_dec_layers, dims.d_ff, tok.vocab_size_in,
                                              tok.vocab_size_out, seed=0, head_gain=6.0, gains={"decoder_embedder": 0.5})
        else:
            # (ropewhisper: the V30 wiring -- difficulty / mapper / song-position vectors, 3 x 128 conditioning channels into conv1)
            sd = random_whisper_family_state_dict("hf" if fam == "whisper" else "rope", dims.d_model, dims.n_heads, dims.n_enc_layers,
                                                  dims.n_dec_layers, dims.d_ff, tok.vocab_size_in, tok.vocab_size_out, n_mels,
                                                  src_positions=frames // 2, tgt_positions=tgt, cond_size=384 if fam == "ropewhisper" else 0,
                                                  seed=0, head_gain=6.0, gains={"decoder_embedder": 0.5})
            if fam == "ropewhisper":
                from mh_testing import add_random_cond_embedders
                add_random_cond_embedders(sd, cond_dim=128, num_mappers=11, seed=0)
        kw = dict(f_min=0 if fam == "whisper" else 20)
    model = MapperatorinatorHIP(sd, dims, vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out, n_mels=n_mels,
                                src_seq_len=frames, tgt_seq_len=tgt, dtype=dtype, device=device, options=options, **kw)
    return model, tok, dims, frames


def step_bytes(dims, n_rows: int, kv_rows: int, src_len: int, vocab_out: int, es: int, t_avg: float, gated: bool) -> float:
    """SURVEY.md 8d algorithmic bytes of one token step: decoder weights + head once, per cross-K/V row the K/V of every layer,
    per decode row the self K/V written so far."""
    d, dff, inner, L = dims.d_model, dims.d_ff, dims.n_heads * 64, dims.n_dec_layers
    per_layer = (3 * inner * d + inner * d) + (inner * d + inner * d) + ((2 if gated else 1) * dff * d + d * dff)
    weights = L * per_layer * es + d * vocab_out * es
    cross = L * 2 * dims.n_heads * src_len * 64 * es
    self_kv = L * 2 * dims.n_heads * t_avg * 64 * es
    return weights + kv_rows * cross + n_rows * self_kv


def run(model_name="t5-base", new_tokens=256, device="cuda:0", reps=3, model_tuple=None, options=None):
    from mapperatorinator_amd.server import build_sampling
    dev = torch.device(device)
    tgt = 1 + new_tokens
    model, tok, dims, frames = model_tuple or build(model_name, tgt, dev, options=options)
    eng = model.engine
    gated = not model.is_whisper
    src_len = eng.packed.src_len
    from mh_testing import synthetic_audio_varied
    audio = synthetic_audio_varied(1, (frames - 1) * 128, seed=5).to(dev)
    eos_table = torch.zeros(tok.vocab_size_out, dtype=torch.uint8, device=dev)
    out = {}
    with torch.no_grad():
        eng._enter()
        with torch.cuda.stream(eng.stream):
            cc = getattr(eng.packed, "cond_channels", 0)
            rb = torch.randn(1, cc, generator=torch.Generator().manual_seed(1)).to(dev) if cc else None
            kv = eng.cross_kv(eng.encode_mel(eng.mel(audio), row_bias=rb))
        eng._leave()
        torch.cuda.synchronize(dev)
        for name, rows, cfg_scale in (("b1", 1, 1.0), ("b2_cfg", 2, 2.0)):
            gk = dict(do_sample=False, num_beams=1, max_length=tgt, temperature=1.0, context_type="map", pad_token_id=0, cfg_scale=cfg_scale)
            sp, _ = build_sampling(tok, gk, tgt)
            prompt = torch.full((rows, 1), tok.sos_id, dtype=torch.int32, device=dev)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            times = []
            for r in range(reps + 1):
                eng._enter()
                with torch.cuda.stream(eng.stream):
                    ev0.record(eng.stream)
                    tokens, n_out, _ = eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
                    ev1.record(eng.stream)
                eng._leave()
                torch.cuda.synchronize(dev)
                if r:
                    times.append(ev0.elapsed_time(ev1))
            ms = sorted(times)[len(times) // 2]
            us_step = ms * 1e3 / new_tokens
            by = step_bytes(dims, rows, 1, src_len, tok.vocab_size_out, 2, new_tokens / 2, gated)
            out[name] = {"rows": rows, "us_per_token_step": round(us_step, 1), "tokens_per_s": round(new_tokens / (ms / 1e3), 1),
                         "roofline_step": {"bound": "hbm", "alg_bytes_per_step": int(by), "floor_us": round(by / (HBM_PEAK_GBS * 1e3), 1),
                                           "achieved": round(by / (us_step * 1e-6) / 1e9, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                           "frac": round(by / (us_step * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}
    out["workload"] = (f"{model_name} bf16, ONE window ({frames} mel frames -> {src_len} encoder positions), {new_tokens} greedy tokens, decode loop "
                       "only (cross K/V resident): the reference's default `parallel: false` call shape; b2_cfg = the doubled batch of guidance "
                       "(one returned row)")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="t5-base")
    ap.add_argument("--new-tokens", type=int, default=256)
    ap.add_argument("--options", default="", help="engine options, name=value[,name=value...] (include/mapperhip.h)")
    a = ap.parse_args()
    opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.options.split(",") if kv}
    r = run(a.model, a.new_tokens, options=opts or None)
    if opts:
        r["options"] = opts
    print(json.dumps(r))
