#!/usr/bin/env python
"""Headline benchmark: beatmap event-tokens/s of the audio->event hot path on N x MI355X
(BASELINE.json configs[1]: osuT5-base, bf16 storage, batch = 32 ten-second chunks per GPU,
mel + encoder + cross-KV + KV-cached greedy AR decode, all hand-written HIP kernels), plus the
diffusion steps/s per song-chunk of the DiT-S + 100-step DDPM refinement stage (configs[2]) as an
auxiliary figure.

One "step" = one pass of the hot path over one batch of synthetic chunks that already sits in HBM:
  mh_mel -> mh_t5_encode -> mh_t5_cross_kv -> mh_t5_generate (new_tokens greedy tokens per row)
  [-> all_gather of the token streams over RCCL when N > 1].
value = (non-pad generated tokens summed over ALL ranks) / (max over ranks of the timed region / steps).

Contract: python bench.py --gpus N --steps K --warmup W ; for N > 1 launched by torch.distributed.run,
one rank per GPU.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", default="base", choices=["tiny", "small", "base", "large"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32, help="chunks per GPU")
    ap.add_argument("--new-tokens", type=int, default=384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dit", action="store_true")
    return ap.parse_args()


def cpu_baseline(size: str, vocab, seconds_budget: float = 25.0):
    """The CPU oracle (a port of the reference's algorithm: oracle/t5.py, torch-CPU fp32, all host cores)
    on a BOUNDED sample of the same workload.  This is the only place bench.py touches oracle/."""
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mapperatorinator_amd.testing import random_t5_state_dict, synthetic_audio
    from oracle import t5 as ot5
    # small-matrix decode work does not scale past a handful of threads (256 threads on the 2-socket host
    # of the GPU box made it 100x SLOWER); use min(host cores, 16) threads and report that number
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    d = T5_PRESETS[size]
    vin, vout, ts0, ts1 = vocab
    sd = random_t5_state_dict(d, vin, vout, seed=0, lm_head_gain=6.0)
    o = ot5.T5Oracle(sd, d.d_model, d.d_ff, d.n_heads, d.n_enc_layers, d.n_dec_layers)
    Bc = 4
    audio = synthetic_audio(Bc, 160000, seed=0)
    prompt = torch.tensor([[1]] * Bc)
    t0 = time.perf_counter()
    enc = o.encode_audio(audio)
    t_enc = time.perf_counter() - t0
    new = 8
    t1 = time.perf_counter()
    ids = o.generate(enc, prompt, None, [], 1 + new, ts0, ts1, [1])
    t_dec = time.perf_counter() - t1
    if t_enc + t_dec * 4 < seconds_budget:      # cheap enough: take a longer decode sample
        new = 32
        t1 = time.perf_counter()
        ids = o.generate(enc, prompt, None, [], 1 + new, ts0, ts1, [1])
        t_dec = time.perf_counter() - t1
    n_tok = int((ids[:, 1:] != 0).sum())
    # whole-path rate for chunks that decode `full_new` tokens each: encoder cost amortised over them
    full_new = 384
    per_chunk = t_enc / Bc + full_new * (t_dec / n_tok)
    return {"value": full_new / per_chunk, "unit": "event-tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle/t5.py (torch-CPU fp32 restatement of the reference path), osuT5-{size}, batch {Bc} x 10 s "
                      f"chunks: mel+encoder {t_enc:.2f} s, {new} greedy tokens/chunk decoded in {t_dec:.2f} s "
                      f"({n_tok / t_dec:.1f} tok/s decode-only); value = 384 / (encoder s per chunk + 384 x decode s per token)"}


def cpu_dit_baseline(n_steps: int = 6):
    """The DiT-S denoiser of configs[2] through the CPU oracle (oracle/dit.py: the checker, timed here as the `port`
    baseline of the diffusion metric): `n_steps` forward_with_cfg calls at Tq = 128, CFG batch 2."""
    from mapperatorinator_amd.testing import DIT_PRESETS, random_dit_state_dict, synthetic_dit_inputs
    from oracle import dit as odit
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    depth, hidden, heads = DIT_PRESETS["DiT-S"]
    orc = odit.DiTOracle(random_dit_state_dict(depth, hidden, seed=0), depth, hidden, heads)
    z, c, y = synthetic_dit_inputs(128, seed=0)
    t = torch.full((2,), 50, dtype=torch.long)
    orc.forward_with_cfg(z, t, c, y, 1.0)                     # warm-up
    t0 = time.perf_counter()
    for _ in range(n_steps):
        orc.forward_with_cfg(z, t, c, y, 1.0)
    dt = (time.perf_counter() - t0) / n_steps
    return {"value": round(1.0 / dt, 2), "unit": "diffusion steps/s per chunk", "cores": cores, "kind": "port",
            "sample": f"oracle/dit.py DiT-S fp32, Tq=128, CFG batch 2, {n_steps} denoiser calls ({dt * 1e3:.1f} ms each); the "
                      f"DDPM update itself is negligible"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "RANK" in os.environ   # under torch.distributed.run the RCCL path is used even for N=1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from mapperatorinator_amd import Tokenizer, _lib
    from mapperatorinator_amd.modeling import MapperatorinatorHIP
    from mapperatorinator_amd.server import build_sampling
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mapperatorinator_amd.testing import random_t5_state_dict, synthetic_audio

    lib = _lib.load()
    tdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    tok = Tokenizer.benchmark_vocab(src_seq_len=1251)
    ts0 = [v for k, v in tok.event_start.items() if k.name == "TIME_SHIFT"][0]
    ts1 = [v for k, v in tok.event_end.items() if k.name == "TIME_SHIFT"][0]
    dims = T5_PRESETS[args.size]
    B, new = args.batch, args.new_tokens
    tgt_len = max(512, 1 + new)
    sd = random_t5_state_dict(dims, tok.vocab_size_in, tok.vocab_size_out, seed=0, lm_head_gain=6.0)
    model = MapperatorinatorHIP(sd, dims, vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out,
                                src_seq_len=1251, tgt_seq_len=tgt_len, dtype=tdtype, device=dev)
    eng = model.engine
    del sd
    audio = synthetic_audio(B, 160000, seed=rank).to(dev)            # resident in HBM before the timed region
    prompt = torch.full((B, 1), tok.sos_id, dtype=torch.int32, device=dev)
    gk = dict(do_sample=False, num_beams=1, max_length=1 + new, temperature=1.0, context_type="map", pad_token_id=0)
    sp, eos = build_sampling(tok, gk, tgt_len)
    eos_table = torch.zeros(tok.vocab_size_out, dtype=torch.uint8, device=dev)   # random-init: keep rows running
    gathered = torch.empty((world * B, 1 + new), dtype=torch.int32, device=dev) if use_dist else None

    def one_step():
        eng._enter()
        with torch.cuda.stream(eng.stream):
            enc = eng.encode_mel(eng.mel(audio))
            kv = eng.cross_kv(enc)
            tokens, n_out, _ = eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
        eng._leave()
        if use_dist:
            dist.all_gather_into_tensor(gathered, tokens.contiguous())
        return tokens, kv

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        tokens, kv = one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tokens, kv = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    n_tok = int((tokens[:, 1:] != 0).sum().item())
    stat = torch.tensor([elapsed, float(n_tok)], dtype=torch.float64, device=dev)
    if use_dist:
        tmax = stat.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stat.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, n_tok_total = float(tmax[0]), float(tsum[1])
    else:
        n_tok_total = float(n_tok)
    ms_per_step = elapsed / args.steps * 1e3
    value = n_tok_total / (elapsed / args.steps)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HBM-bound decode cross-attention), measured live -------------
    es = 2 if args.dtype == "bf16" else 4
    alg_bytes = B * dims.n_heads * 1251 * 64 * 2 * es            # K and V of one layer, read once per launch
    ms = C.c_float(0.0)
    ws = eng._workspace("dec", lib.mh_t5_decode_workspace_bytes(C.byref(eng.packed.cfg), B))
    reps = 20 * dims.n_dec_layers
    rc = lib.mh_t5_cross_attn_probe(C.byref(eng.packed.cfg), C.byref(eng.packed.w), kv.data_ptr(), B, reps, C.byref(ms),
                                    ws.data_ptr(), ws.numel(), eng.stream.cuda_stream)
    _lib.check(rc, "mh_t5_cross_attn_probe")
    achieved = alg_bytes / (ms.value * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, gfx950
    # correction applied -- profiles/r01_pmc_hbm_traffic.txt); only quoted for the exact workload it was taken on
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_cross_attn.json")
    if os.path.exists(pmc_path) and args.size == "base" and args.dtype == "bf16" and B == 32:
        with open(pmc_path) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    roofline = {"bound": "hbm", "kernel": "dec_cross_attn_q_kernel (cross-attention over the encoder K/V incl. its query projection)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "alg_bytes_per_launch": alg_bytes, "us_per_launch": round(ms.value * 1e3, 2),
                "launches_per_token_step": dims.n_dec_layers}

    # ---- auxiliary: DiT-S + 100-step DDPM, diffusion steps/s per song-chunk (configs[2]) ---------------
    aux = {}
    if not args.no_dit:
        from mapperatorinator_amd.dit import DiTHIP, create_diffusion
        from mapperatorinator_amd.testing import DIT_PRESETS, random_dit_state_dict, synthetic_dit_inputs
        depth, hidden, heads = DIT_PRESETS["DiT-S"]
        dit = DiTHIP(random_dit_state_dict(depth, hidden, seed=0), depth, hidden, heads, device=dev)
        z, c, y = synthetic_dit_inputs(128, seed=0)
        z, c, y = z.to(dev), c.to(dev), y.to(dev)
        diff = create_diffusion([100, 0, 0, 0, 0, 0, 0, 0, 0, 0], noise_schedule="squaredcos_cap_v2",
                                diffusion_steps=1000)
        kw = dict(c=c, y=y, cfg_scale=1.0, attn_mask=None)
        noise = torch.randn(100, *z.shape, device=dev)
        diff.p_sample_loop(dit.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=noise)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        reps_d = 5
        for _ in range(reps_d):
            diff.p_sample_loop(dit.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=noise)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps_d
        aux = {"diffusion_steps_per_s_per_chunk": round(100 / dt, 1), "dit": "DiT-S fp32, Tq=128, CFG batch 2, "
               "100-step DDPM (fused hipGraph loop)", "ms_per_100_steps": round(dt * 1e3, 2)}

    cpu = None
    if not args.no_cpu_baseline and world == 1:   # the CPU port is timed on rank 0 at N=1 only
        cpu = cpu_baseline(args.size, (tok.vocab_size_in, tok.vocab_size_out, ts0, ts1))
        if aux:
            try:
                aux["cpu_baseline"] = cpu_dit_baseline()
            except Exception as e:   # the auxiliary figure must never cost the bench line
                print(f"cpu_dit_baseline failed: {e!r}", file=sys.stderr)

    line = {
        "metric": "beatmap event-tokens/sec (mel + osuT5 encoder + greedy AR decode), whole job",
        "value": round(value, 1), "unit": "event-tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic 16 kHz audio (noise + tones), random-init weights",
        "config": {"workload": f"osuT5-{args.size} {args.dtype}, batch={B} x 10 s chunks per GPU, {new} greedy tokens "
                               f"per chunk, mel+encoder+cross-KV+AR decode (BASELINE configs[1])",
                   "chunks_per_gpu": B, "new_tokens": new, "src_frames": 1251, "vocab": tok.vocab_size_out,
                   "parallelism": f"chunk-sharded x{world}, all_gather of token streams"},
        "roofline": roofline, "cpu_baseline": cpu, "aux": aux,
    }
    print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
