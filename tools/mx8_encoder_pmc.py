"""The mel + encoder + cross-K/V stage of 32 chunks at BASELINE config 2 dims, once with bf16 operands and once with
MhT5Config.enc_operand_dtype = MH_MX8 -- a small workload for counter passes (rocprofv3 --pmc MfmaUtil around it)."""
import sys

import torch

sys.path.insert(0, ".")
from mapperatorinator_amd import Tokenizer  # noqa: E402
from mapperatorinator_amd.modeling import MapperatorinatorHIP  # noqa: E402
from mapperatorinator_amd.t5_engine import T5_PRESETS  # noqa: E402
from mh_testing import random_t5_state_dict, synthetic_audio  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "base"
src, tgt, B = 1251, 512, 32
tok = Tokenizer.benchmark_vocab(src_seq_len=src)
sd = random_t5_state_dict(T5_PRESETS[size], tok.vocab_size_in, tok.vocab_size_out, seed=0)
audio = synthetic_audio(B, 160000, seed=1).cuda()
for mode in (None, "mx8"):
    m = MapperatorinatorHIP(sd, T5_PRESETS[size], vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out, src_seq_len=src,
                            tgt_seq_len=tgt, dtype=torch.bfloat16, device="cuda", enc_operand_dtype=mode)
    eng = m.engine
    for _ in range(2):
        kv = eng.cross_kv(eng.encode(audio))
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    kv = eng.cross_kv(eng.encode(audio))
    t1.record(); torch.cuda.synchronize()
    print(size, mode or "bf16", "mel + encoder + cross-K/V of 32 chunks:", round(t0.elapsed_time(t1), 2), "ms", flush=True)
    del m, eng, kv
