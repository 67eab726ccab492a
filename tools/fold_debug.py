"""debug aid: rows of a batch decode with the folded output projection vs the stand-alone GEMVs, and run-to-run determinism"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from mapperatorinator_amd import Tokenizer, _lib  # noqa: E402
from mapperatorinator_amd.modeling import MapperatorinatorHIP  # noqa: E402
from mapperatorinator_amd.server import model_generate  # noqa: E402
from mapperatorinator_amd.t5_engine import T5_PRESETS  # noqa: E402
from mapperatorinator_amd.testing import random_t5_state_dict, synthetic_audio  # noqa: E402

src, tgt = 251, 48
tok = Tokenizer.benchmark_vocab(src_seq_len=src)
sd = random_t5_state_dict(T5_PRESETS["small"], tok.vocab_size_in, tok.vocab_size_out, seed=9, lm_head_gain=6.0)
gk = dict(precision="fp32", do_sample=False, num_beams=1, top_p=1.0, top_k=0, max_length=tgt, cfg_scale=1.0, timeshift_bias=0,
          types_first=False, temperature=1.0, lookback_time=0, lookahead_time=0, context_type="map", pad_token_id=0)
for dtype in (torch.bfloat16, torch.float32):
    for B in (40, 20, 17, 16):
        audio = synthetic_audio(B, 32000, seed=12)
        prompt = torch.tensor([[1]] * B)
        mk = dict(inputs=audio, decoder_input_ids=prompt, decoder_attention_mask=prompt.ne(0))
        out = {}
        for name, opts in (("gemv", dict(decode_fold_oproj=0)), ("fold", dict(decode_fold_oproj=1)), ("fold2", dict(decode_fold_oproj=1)),
                           ("fold_1chain", dict(decode_fold_oproj=1, decode_chains=1))):
            m = MapperatorinatorHIP(sd, T5_PRESETS["small"], vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out,
                                    src_seq_len=src, tgt_seq_len=tgt, dtype=dtype, device="cuda", options=opts)
            out[name] = model_generate(m, tok, mk, gk)[0]
        def diff(a, b):
            n = min(a.shape[1], b.shape[1])
            rows = [(r, int((a[r, :n] != b[r, :n]).nonzero()[0])) for r in range(B) if not torch.equal(a[r, :n], b[r, :n])]
            return rows
        print(dtype, "B", B, "fold vs gemv:", diff(out["fold"], out["gemv"]), "| fold vs fold2:", diff(out["fold"], out["fold2"]),
              "| fold vs fold_1chain:", diff(out["fold"], out["fold_1chain"]), flush=True)
