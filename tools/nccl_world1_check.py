#!/usr/bin/env python
"""The N-GPU data path on the backend it will really use, at the world size a 1-GPU box offers: forms an RCCL ("nccl")
process group of ONE rank, runs `sharded_generate(model_generate on the HIP engine, refine_fn = a DiT refine on the device)`
through the collective (token streams + coordinates in ONE all_gather, tensors assembled on the GPU) and compares with the
plain calls.  Run by tests/test_gpu_t5.py::test_sharded_generate_on_rccl_world_1 in a process of its own (a process group is
process-global state).  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from mapperatorinator_amd import Tokenizer
    from mapperatorinator_amd.dit import BandMask, DiTHIP, create_diffusion
    from mapperatorinator_amd.modeling import MapperatorinatorHIP
    from mapperatorinator_amd.server import model_generate
    from mapperatorinator_amd.sharding import sharded_generate
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mh_testing import (DIT_PRESETS, random_dit_state_dict, random_t5_state_dict, synthetic_audio,
                                              synthetic_dit_inputs)
    src, tgt, B, Tq = 126, 48, 5, 32
    tok = Tokenizer.benchmark_vocab(src_seq_len=src)
    d = T5_PRESETS["tiny"]
    model = MapperatorinatorHIP(random_t5_state_dict(d, tok.vocab_size_in, tok.vocab_size_out, seed=3, lm_head_gain=6.0), d,
                                vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out, src_seq_len=src,
                                tgt_seq_len=tgt, dtype=torch.float32, device=dev)
    depth, hidden, heads = DIT_PRESETS["DiT-XS"] if "DiT-XS" in DIT_PRESETS else DIT_PRESETS["DiT-S"]
    dit = DiTHIP(random_dit_state_dict(depth, hidden, seed=0), depth, hidden, heads, device=dev)
    diff = create_diffusion([4, 0, 0, 0, 0, 0, 0, 0, 0, 0], noise_schedule="squaredcos_cap_v2", diffusion_steps=1000)
    audio = synthetic_audio(B, (src - 1) * 128, seed=2)
    prompt = torch.tensor([[1]] * B)
    gk = dict(precision="fp32", do_sample=False, num_beams=1, top_p=1.0, top_k=0, max_length=tgt, cfg_scale=1.0, timeshift_bias=0,
              types_first=False, temperature=1.0, lookback_time=0, lookahead_time=0, context_type="map", pad_token_id=0)
    parts = [synthetic_dit_inputs(Tq, seed=b) for b in range(B)]
    noise = torch.randn(4, 2 * B, 2, Tq, generator=torch.Generator().manual_seed(0)).to(dev)

    def gen(shard):
        mk = dict(inputs=shard["inputs"], decoder_input_ids=shard["decoder_input_ids"],
                  decoder_attention_mask=shard["decoder_input_ids"].ne(0))
        return model_generate(model, tok, mk, gk)

    def refine(shard, toks):
        lo = shard["_row_offset"]
        n = toks.shape[0]
        sel = parts[lo: lo + n]
        z = torch.cat([p[0][:1] for p in sel] + [p[0][1:] for p in sel]).to(dev)
        c = torch.cat([p[1][:1] for p in sel] + [p[1][1:] for p in sel]).to(dev)
        y = torch.cat([p[2][:1] for p in sel] + [p[2][1:] for p in sel]).to(dev)
        kw = dict(c=c, y=y, cfg_scale=1.0, attn_mask=BandMask(Tq, 128))
        out = diff.p_sample_loop(dit.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=noise)
        return out[:n]                                              # DEVICE tensor: travels without a host bounce

    mk = dict(inputs=audio, decoder_input_ids=prompt)
    toks, lens, stats, coords = sharded_generate(gen, mk, pad_id=0, max_length=tgt, refine_fn=refine)
    plain, _ = gen(dict(mk, _row_offset=0))
    ref_c = refine(dict(mk, _row_offset=0), plain).cpu()
    ok_t = bool(torch.equal(toks[:, : plain.shape[1]], plain) and (toks[:, plain.shape[1]:] == 0).all())
    ok_c = bool(torch.equal(coords, ref_c))
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"backend": "nccl", "world": 1, "tokens_equal": ok_t, "coords_bit_equal": ok_c,
                      "coords_shape": list(coords.shape), "lens": lens.tolist()}), flush=True)
    return 0 if (ok_t and ok_c) else 1


if __name__ == "__main__":
    sys.exit(main())
