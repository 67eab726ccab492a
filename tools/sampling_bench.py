"""Decode throughput of config 2 under the reference's default USER sampling settings (configs/inference/v32.yaml:
temperature 0.9, top_p 0.9, do_sample) next to greedy: what the device-side top-p search costs per token step."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapperatorinator_amd import Tokenizer
from mapperatorinator_amd.modeling import MapperatorinatorHIP
from mapperatorinator_amd.server import build_sampling
from mapperatorinator_amd.t5_engine import T5_PRESETS
from mh_testing import random_t5_state_dict, synthetic_audio

dev = torch.device("cuda:0")
tok = Tokenizer.benchmark_vocab(src_seq_len=1251)
d = T5_PRESETS["base"]
B, new = 32, 384
model = MapperatorinatorHIP(random_t5_state_dict(d, tok.vocab_size_in, tok.vocab_size_out, seed=0, lm_head_gain=6.0), d,
                            vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out, src_seq_len=1251,
                            tgt_seq_len=512, dtype=torch.bfloat16, device=dev)
eng = model.engine
audio = synthetic_audio(B, 160000, seed=0).to(dev)
prompt = torch.full((B, 1), tok.sos_id, dtype=torch.int32, device=dev)
eos_table = torch.zeros(tok.vocab_size_out, dtype=torch.uint8, device=dev)
out = {}
for name, gk in (("greedy", dict(do_sample=False)), ("top_p 0.9, T 0.9", dict(do_sample=True, top_p=0.9, temperature=0.9, seed=1)),
                 ("top_k 50 + top_p 0.9", dict(do_sample=True, top_p=0.9, top_k=50, temperature=0.9, seed=1))):
    sp, _ = build_sampling(tok, dict(gk, max_length=1 + new), 512)
    eng._enter()
    with eng.on_stream():
        kv = eng.cross_kv(eng.encode_mel(eng.mel(audio)))
        eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            tokens, _, _ = eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
        torch.cuda.synchronize(dev)
    eng._leave()
    dt = (time.perf_counter() - t0) / 3
    out[name] = {"decode_ms": round(dt * 1e3, 2), "us_per_token_step": round(dt * 1e6 / new, 1),
                 "distinct_ids": len(set(tokens.flatten().tolist()))}
# the v30+ processor set: types_first tokenizer, conditional temperatures, lookback renormalisation (logit_processors.py:47-133)
del model, eng
tok2 = Tokenizer.from_json(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tokenizer_types_first.json"))
model = MapperatorinatorHIP(random_t5_state_dict(d, tok2.vocab_size_in, tok2.vocab_size_out, seed=0, lm_head_gain=6.0), d,
                            vocab_size_in=tok2.vocab_size_in, vocab_size_out=tok2.vocab_size_out, src_seq_len=1251,
                            tgt_seq_len=512, dtype=torch.bfloat16, device=dev)
eng = model.engine
prompt = torch.full((B, 1), tok2.sos_id, dtype=torch.int32, device=dev)
eos_table = torch.zeros(tok2.vocab_size_out, dtype=torch.uint8, device=dev)
for name, gk in (("types_first greedy", dict(do_sample=False, types_first=True, lookback_time=500.0)),
                 ("types_first, top_p 0.9, timing temperature 0.1", dict(do_sample=True, top_p=0.9, temperature=0.9, timing_temperature=0.1,
                                                                           mania_column_temperature=0.8, taiko_hit_temperature=0.8,
                                                                           types_first=True, lookback_time=500.0, seed=1,
                                                                           conditional_temperature_per_row=True))):
    sp, _ = build_sampling(tok2, dict(gk, max_length=1 + new), 512)
    eng._enter()
    with eng.on_stream():
        kv = eng.cross_kv(eng.encode_mel(eng.mel(audio)))
        eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            tokens, _, _ = eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
        torch.cuda.synchronize(dev)
    eng._leave()
    dt = (time.perf_counter() - t0) / 3
    out[name] = {"decode_ms": round(dt * 1e3, 2), "us_per_token_step": round(dt * 1e6 / new, 1), "vocab": tok2.vocab_size_out,
                 "distinct_ids": len(set(tokens.flatten().tolist()))}
print(json.dumps(out))
