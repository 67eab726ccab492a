#!/usr/bin/env python
"""TEST / MEASUREMENT INFRASTRUCTURE: the REFERENCE's own CPU path, timed.

Imports the unmodified reference from /root/reference through oracle/ref_harness.py (stubs for the uninstallable packages,
the restated nnAudio mel -- the same harness that generates tests/golden/) and times, on this container's host cores:
  config 1   `model_generate` (osuT5/osuT5/inference/server.py:83-156), osuT5-small fp32, ONE 10 s chunk, 128 greedy tokens
             (BASELINE configs[0]: "CPU reference path");
  config 2'  the same at osuT5-base dims, a batch of 4 chunks x 128 tokens (the headline batch is 32 x 384: the CPU figure is a
             per-token rate, the sample is bounded);
  DiT-S      `diffusion.p_sample_loop(model.forward_with_cfg, ...)` (gaussian_diffusion.py:469-561), Tq = 128, CFG batch 2,
             a bounded number of steps.
Writes one JSON record (profiles/r04_cpu_reference.json) that bench.py quotes as `cpu_baseline.reference_recorded`: the
reference cannot travel to the GPU box (pure Python + transformers, /root/reference is absent there), so its number is
recorded HERE, reproducibly, and the live CPU leg of bench.py stays the oracle port.

    python oracle/time_reference.py [--threads N] [--out profiles/r04_cpu_reference.json]"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, 16))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_cpu_reference.json"))
    ap.add_argument("--dit-steps", type=int, default=20)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from mh_testing import synthetic_audio, synthetic_dit_inputs
    from oracle import dit as odit
    from oracle import ref_harness as rh
    rec = {"what": "the unmodified reference (/root/reference) on this container's CPU, via oracle/ref_harness.py",
           "host": {"cpu_count": os.cpu_count(), "threads_used": args.threads, "machine": platform.machine(), "torch": torch.__version__},
           "runs": {}}

    def t5_run(size, batch, new_tokens):
        model, tok, _ = rh.build_reference_t5(size, src_seq_len=1251, tgt_seq_len=512, lm_head_gain=6.0)
        audio = synthetic_audio(batch, 160000, seed=0)
        prompt = torch.full((batch, 1), int(tok.sos_id), dtype=torch.long)
        gk = rh.default_generate_kwargs(1 + new_tokens)
        # random-init rows may stop at an EOS-set id before `new_tokens`: the rate counts the tokens really generated, as
        # the reference's own statistics do (server.py:50-69)
        rh.reference_encode(model, audio[:1])             # warm-up: first-call costs (thread pool, mel tables) are not the path's
        t0 = time.perf_counter()
        enc = rh.reference_encode(model, audio)
        t_enc = time.perf_counter() - t0
        t1 = time.perf_counter()
        ids, stats = rh.reference_generate(model, tok, audio, prompt, gk)
        t_all = time.perf_counter() - t1          # (encodes again inside: reference_generate = encoder + model_generate)
        n_tok = int(stats["generated_tokens"])
        return {"model": f"google/t5-v1_1-{size} dims under the reference wrapper, fp32, eager attention", "batch": batch,
                "new_tokens_asked": new_tokens, "generated_tokens": n_tok, "encoder_seconds": round(t_enc, 3),
                "encoder_plus_model_generate_seconds": round(t_all, 3),
                "tokens_per_second_end_to_end": round(n_tok / t_all, 2),
                "model_generate_seconds": round(float(stats["elapsed_seconds"]), 3),
                "reference_stats_tokens_per_second": round(float(stats["tokens_per_second"]), 2)}

    rec["runs"]["config1_small_1x128"] = t5_run("small", 1, 128)
    print(json.dumps(rec["runs"]["config1_small_1x128"]), flush=True)
    rec["runs"]["config2_base_4x128"] = t5_run("base", 4, 128)
    print(json.dumps(rec["runs"]["config2_base_4x128"]), flush=True)

    dit = rh.build_reference_dit("DiT-S")
    diff = rh.reference_diffusion()
    z, c, y = synthetic_dit_inputs(128, seed=0)
    mask = odit.band_mask(128, 128)
    t = torch.full((2,), 50, dtype=torch.long)
    with torch.no_grad():
        dit.forward_with_cfg(z, t, c, y, 1.0, attn_mask=mask)          # warm-up
        t0 = time.perf_counter()
        for _ in range(args.dit_steps):
            dit.forward_with_cfg(z, t, c, y, 1.0, attn_mask=mask)
        dt = (time.perf_counter() - t0) / args.dit_steps
    g = torch.Generator().manual_seed(0)
    noise = [torch.randn(z.shape, generator=g) for _ in range(100)]
    t0 = time.perf_counter()
    rh.reference_ddpm(dit, diff, z, c, y, 1.0, mask, noise)
    t_loop = time.perf_counter() - t0
    rec["runs"]["dit_s_tq128"] = {"model": "osu_diffusion DiT-S fp32, Tq = 128, CFG batch 2", "forward_with_cfg_ms": round(dt * 1e3, 2),
                                  "p_sample_loop_100_steps_seconds": round(t_loop, 3), "diffusion_steps_per_second_per_chunk": round(100 / t_loop, 2)}
    print(json.dumps(rec["runs"]["dit_s_tq128"]), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
