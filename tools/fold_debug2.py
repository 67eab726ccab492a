"""debug aid: single rows vs the batch under the folded output projection, with the size of the first logit difference"""
import sys

import torch

sys.path.insert(0, ".")
from mapperatorinator_amd import Tokenizer  # noqa: E402
from mapperatorinator_amd.modeling import MapperatorinatorHIP  # noqa: E402
from mapperatorinator_amd.server import build_sampling  # noqa: E402
from mapperatorinator_amd.t5_engine import T5_PRESETS  # noqa: E402
from mapperatorinator_amd.testing import random_t5_state_dict, synthetic_audio  # noqa: E402

src, tgt = 251, 48
tok = Tokenizer.benchmark_vocab(src_seq_len=src)
sd = random_t5_state_dict(T5_PRESETS["small"], tok.vocab_size_in, tok.vocab_size_out, seed=9, lm_head_gain=6.0)
gk = dict(precision="fp32", do_sample=False, num_beams=1, top_p=1.0, top_k=0, max_length=tgt, cfg_scale=1.0, timeshift_bias=0,
          types_first=False, temperature=1.0, lookback_time=0, lookahead_time=0, context_type="map", pad_token_id=0)
B = 40
audio = synthetic_audio(B, 32000, seed=12)
prompt = torch.tensor([[1]] * B)
for fold in (1, 0):
    m = MapperatorinatorHIP(sd, T5_PRESETS["small"], vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out,
                            src_seq_len=src, tgt_seq_len=tgt, dtype=torch.bfloat16, device="cuda", options=dict(decode_fold_oproj=fold))
    sp, eos = build_sampling(tok, gk, tgt)
    full = m.engine.generate(audio, prompt, prompt.ne(0), eos, sp, dump_logits=True)
    bad = []
    for b in range(B):
        for nb in (1, 2):
            rows = [b] if nb == 1 else [b, (b + 1) % B]
            sp1, _ = build_sampling(tok, gk, tgt)
            one = m.engine.generate(audio[rows], prompt[:nb], prompt[:nb].ne(0), eos, sp1, dump_logits=True)
            n = min(one["tokens"].shape[1], full["tokens"].shape[1])
            if not torch.equal(one["tokens"][0, :n], full["tokens"][b, :n]):
                col = int((one["tokens"][0, :n] != full["tokens"][b, :n]).nonzero()[0])
                # logits of the first differing column and of the one before
                d0 = (one["logits"][col, 0] - full["logits"][col, b]).abs().max().item()
                d1 = (one["logits"][1, 0] - full["logits"][1, b]).abs().max().item()
                top = full["logits"][col, b].topk(2).values
                bad.append((b, nb, col, round(d0, 5), round(d1, 7), round(float(top[0] - top[1]), 5)))
    print("fold", fold, "rows whose solo run differs (row, solo batch, column, |dlogit| there, |dlogit| at column 1, top-2 gap):", bad, flush=True)
