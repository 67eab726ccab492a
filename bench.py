#!/usr/bin/env python
"""Headline benchmark: beatmap event-tokens/s of the audio->event hot path on N x MI355X
(BASELINE.json configs[1]: osuT5-base, bf16 storage, batch = 32 ten-second chunks per GPU,
mel + encoder + cross-KV + KV-cached greedy AR decode, all hand-written HIP kernels).

One "step" = one pass of the hot path over one batch of synthetic chunks that already sits in HBM:
  mh_mel -> mh_t5_encode -> mh_t5_cross_kv -> mh_t5_generate (new_tokens greedy tokens per row)
  [-> all_gather of the token streams over RCCL when N > 1].
value = (non-pad generated tokens summed over ALL ranks) / (max over ranks of the timed region / steps).

Besides the contract fields the JSON line carries
  roofline      the dominant kernel (decode cross-attention) AS THE TIMED REGION LAUNCHES IT: rows per launch and
                in-situ microseconds from device-side timestamps taken in an EXTRA decode pass (never the timed one),
                next to the stand-alone full-batch probe and to the step-level figure (SURVEY 8d bytes per token step /
                measured decode time per token step);
  cpu_baseline  the CPU oracle (a port, kind "port") on a bounded sample of the same workload;
  aux           config 1 (osuT5-small, 1 chunk, 128 tokens: GPU and CPU port), config 3 (T5 + DiT-S 100-step DDPM
                refine of all 32 chunks as one denoiser batch: diffusion steps/s per chunk, end-to-end chunks/s), config 5
                (osuT5-large / base over 32 three-minute songs with resident cross K/V, bf16 and e4m3; DiT-B), each with
                its own roofline_step, the end-to-end rate through `model_generate` including H2D / D2H, per-stage ms.

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

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# bench.py owns its process, so it makes the EXPLICIT runtime opt-in INTEGRATION.md 3 describes, before the HIP runtime
# initialises (the package import itself never touches os.environ): hipGraph replay through the runtime's classic per-node
# submission.  An explicit value in the environment wins; `aux.runtime_flag_ab` re-measures the headline under the runtime's
# default path in a child process, so the line shows both.
import mapperatorinator_amd  # noqa: E402
RUNTIME = mapperatorinator_amd.configure_runtime()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
F32_MFMA_PEAK_TF = 157.3  # exact-f32 MFMA (= fp32 vector) peak, same guide
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA peak, same guide
MX8_MFMA_PEAK_TF = 5000.0   # dense MX-fp8 MFMA peak (v_mfma_scale_f32_*_f8f6f4), same guide
SRC_FRAMES, N_SAMPLES = 1251, 160000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE under a launcher, else 1")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", default="base", choices=["tiny", "small", "base", "large"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32, help="chunks per GPU")
    ap.add_argument("--new-tokens", type=int, default=384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dit", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip config 1 / end-to-end / in-situ passes")
    ap.add_argument("--no-config5", action="store_true", help="skip the long-song (osuT5-large, 3 min songs) and DiT-B aux lines")
    ap.add_argument("--headline-only", action="store_true",
                    help="time the headline region only and print {value, ms_per_step, runtime_env}: the child run of aux.runtime_flag_ab")
    ap.add_argument("--no-runtime-ab", action="store_true", help="skip the child run under the runtime's default graph-replay path")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="only form the process group (RCCL on a GPU box, gloo without one), all_gather the rank ids, print "
                         "{selftest, n_gpus, rccl_ranks} and exit: the launch path of --gpus N without the workload")
    return ap.parse_args()


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks_if_needed(args) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher.  The process re-executes itself
    under `torch.distributed.run --nproc-per-node N` (one rank per GPU, rendezvous on 127.0.0.1) and exits with the job's
    status, so `--gpus N` ALWAYS means N ranks -- whether the driver wrapped the command in torchrun or not."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def selftest_launch(args, world: int, rank: int, local_rank: int) -> None:
    """The launch path alone: group formation on the backend the box offers, one all_gather, one line from rank 0."""
    on_gpu = torch.cuda.is_available()
    if on_gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = dict(device_id=dev) if on_gpu else {}
        dist.init_process_group("nccl" if on_gpu else "gloo", rank=rank, world_size=world, **kw)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        mine = torch.tensor([rank], dtype=torch.int32, device=dev)
        allr = torch.empty(world, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        ranks = allr.cpu().tolist()
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks = [0]
    if rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": world, "rccl_ranks": ranks,
                          "backend": "nccl (RCCL)" if on_gpu else "gloo"}), flush=True)


# ---------------------------------------------------------------------------------------------------------
# CPU baselines (the ONLY places bench.py touches oracle/: the checker timed as the `port` baseline)
# ---------------------------------------------------------------------------------------------------------
def _cpu_threads():
    # small-matrix decode work does not scale past a handful of threads (256 threads on the 2-socket host of the
    # GPU box made it 100x SLOWER); use min(host cores, 16) threads and report that number
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    return cores


def cpu_baseline(size: str, vocab, seconds_budget: float = 25.0):
    """The CPU oracle (a port of the reference's algorithm: oracle/t5.py, torch-CPU fp32) on a BOUNDED sample of the
    headline workload."""
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mh_testing import random_t5_state_dict, synthetic_audio
    from oracle import t5 as ot5
    cores = _cpu_threads()
    d = T5_PRESETS[size]
    vin, vout, ts0, ts1 = vocab
    sd = random_t5_state_dict(d, vin, vout, seed=0, lm_head_gain=6.0)
    o = ot5.T5Oracle(sd, d.d_model, d.d_ff, d.n_heads, d.n_enc_layers, d.n_dec_layers)
    Bc = 4
    audio = synthetic_audio(Bc, N_SAMPLES, seed=0)
    prompt = torch.tensor([[1]] * Bc)
    t0 = time.perf_counter()
    enc = o.encode_audio(audio)
    t_enc = time.perf_counter() - t0
    # 128 tokens per chunk (the decode rate must not be a 32-step estimate) unless the host is so slow that the
    # sample would not fit `seconds_budget`: 8 probe tokens price a step, then as many as the budget allows (>= 32)
    t1 = time.perf_counter()
    o.generate(enc, prompt, None, [], 1 + 8, ts0, ts1, [1])
    per_step = (time.perf_counter() - t1) / 8
    new = int(max(32, min(128, (seconds_budget - t_enc) / max(per_step, 1e-6))))
    t1 = time.perf_counter()
    ids = o.generate(enc, prompt, None, [], 1 + new, ts0, ts1, [1])
    t_dec = time.perf_counter() - t1
    n_tok = int((ids[:, 1:] != 0).sum())
    full_new = 384
    per_chunk = t_enc / Bc + full_new * (t_dec / n_tok)
    out = {"value": full_new / per_chunk, "unit": "event-tokens/s", "cores": cores, "kind": "port",
           "sample": f"oracle/t5.py (torch-CPU fp32 restatement of the reference path), osuT5-{size}, batch {Bc} x 10 s "
                     f"chunks: mel+encoder {t_enc:.2f} s, {new} greedy tokens/chunk decoded in {t_dec:.2f} s "
                     f"({n_tok / t_dec:.1f} tok/s decode-only); value = 384 / (encoder s per chunk + 384 x decode s per token)"}
    # the REFERENCE itself cannot run on the GPU box (pure Python + transformers; /root/reference is absent there): its CPU
    # numbers are recorded by oracle/time_reference.py in the build container and quoted here from the committed record
    try:
        with open(os.path.join(ROOT, "profiles", "r04_cpu_reference.json")) as f:
            rr = json.load(f)
        run = rr["runs"]["config2_base_4x128" if size == "base" else "config1_small_1x128"]
        out["reference_recorded"] = {"value": run["reference_stats_tokens_per_second"], "unit": "event-tokens/s (model_generate only, the reference's own stats)",
                                     "end_to_end_tokens_per_s": run["tokens_per_second_end_to_end"], "cores": rr["host"]["threads_used"],
                                     "kind": "reference", "where": "build container CPU, not this box", "source": "profiles/r04_cpu_reference.json "
                                     "(oracle/time_reference.py: the unmodified reference's model_generate via oracle/ref_harness.py)",
                                     "sample": f"{run['model']}, batch {run['batch']} x {run['new_tokens_asked']} greedy tokens"}
    except Exception as e:   # the record is evidence, not a dependency of the line
        print(f"reference CPU record not quoted: {e!r}", file=sys.stderr)
    return out


def cpu_config1(vocab, new_tokens: int = 128):
    """BASELINE configs[0] (osuT5-small, ONE 10 s chunk, greedy, 128 new tokens) through the CPU port."""
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mh_testing import random_t5_state_dict, synthetic_audio
    from oracle import t5 as ot5
    cores = _cpu_threads()
    d = T5_PRESETS["small"]
    vin, vout, ts0, ts1 = vocab
    o = ot5.T5Oracle(random_t5_state_dict(d, vin, vout, seed=0, lm_head_gain=6.0), d.d_model, d.d_ff, d.n_heads,
                     d.n_enc_layers, d.n_dec_layers)
    audio = synthetic_audio(1, N_SAMPLES, seed=0)
    t0 = time.perf_counter()
    enc = o.encode_audio(audio)
    ids = o.generate(enc, torch.tensor([[1]]), None, [], 1 + new_tokens, ts0, ts1, [1])
    dt = time.perf_counter() - t0
    return {"value": round(int((ids[:, 1:] != 0).sum()) / dt, 1), "unit": "event-tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle/t5.py, osuT5-small fp32, 1 chunk, {new_tokens} greedy tokens, mel+encoder+decode in {dt:.2f} s"}


def cpu_dit_baseline(n_steps: int = 6):
    """The DiT-S denoiser of configs[2] through the CPU oracle: `n_steps` forward_with_cfg calls at Tq = 128, CFG batch 2."""
    from mh_testing import DIT_PRESETS, random_dit_state_dict, synthetic_dit_inputs
    from oracle import dit as odit
    cores = _cpu_threads()
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


# ---------------------------------------------------------------------------------------------------------
def decode_step_bytes(dims, B: int, vocab_out: int, es: int, t_avg: float) -> float:
    """SURVEY.md 8d algorithmic bytes of ONE token step: decoder weights + lm_head, read once per step, + per row the
    cross-attention K/V of every layer and the self-attention K/V written so far (t_avg positions)."""
    d, dff, inner, L = dims.d_model, dims.d_ff, dims.inner, dims.n_dec_layers
    per_layer = (3 * inner * d + inner * d) + (inner * d + inner * d) + (2 * dff * d + d * dff)   # self qkv+o, cross q+o, ffn
    weights = L * per_layer * es + d * vocab_out * es
    cross = L * 2 * dims.n_heads * SRC_FRAMES * 64 * es
    self_kv = L * 2 * dims.n_heads * t_avg * 64 * es
    return weights + B * (cross + self_kv)


def dit_flops_per_step(depth: int, D: int, N: int, T: int, band: int = 0) -> float:
    """N * [depth * (2 T 12 D^2 + 4 T^2 D) + 2 T 528 D] (dense attention count)."""
    return N * (depth * (2.0 * T * 12 * D * D + 4.0 * T * T * D) + 2.0 * T * 528 * D)


def main():
    args = parse()
    if args.gpus is None:                          # not given: follow the launcher (`torchrun --nproc-per-node=N bench.py` runs N ranks)
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    spawn_ranks_if_needed(args)                    # `--gpus N` without a launcher: re-execute under torch.distributed.run
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher formed WORLD_SIZE={world}; n_gpus must never "
                         f"disagree with --gpus")
    if args.selftest_launch:
        return selftest_launch(args, world, rank, local_rank)
    use_dist = world > 1 or "RANK" in os.environ   # under torch.distributed.run the RCCL path is used even for N=1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from mapperatorinator_amd import Tokenizer, _lib
    from mapperatorinator_amd.modeling import MapperatorinatorHIP
    from mapperatorinator_amd.server import build_sampling, model_generate
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mh_testing import random_t5_state_dict, synthetic_audio

    lib = _lib.load()
    tdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    tok = Tokenizer.benchmark_vocab(src_seq_len=SRC_FRAMES)
    ts0 = [v for k, v in tok.event_start.items() if k.name == "TIME_SHIFT"][0]
    ts1 = [v for k, v in tok.event_end.items() if k.name == "TIME_SHIFT"][0]
    vocab = (tok.vocab_size_in, tok.vocab_size_out, ts0, ts1)
    dims = T5_PRESETS[args.size]
    B, new = args.batch, args.new_tokens
    tgt_len = max(512, 1 + new)
    sd = random_t5_state_dict(dims, tok.vocab_size_in, tok.vocab_size_out, seed=0, lm_head_gain=6.0)
    model = MapperatorinatorHIP(sd, dims, vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out,
                                src_seq_len=SRC_FRAMES, tgt_seq_len=tgt_len, dtype=tdtype, device=dev)
    eng = model.engine
    del sd
    audio_host = synthetic_audio(B, N_SAMPLES, seed=rank)
    audio = audio_host.to(dev)                                             # resident in HBM before the timed region
    prompt = torch.full((B, 1), tok.sos_id, dtype=torch.int32, device=dev)
    gk = dict(do_sample=False, num_beams=1, max_length=1 + new, temperature=1.0, context_type="map", pad_token_id=0)
    sp, eos = build_sampling(tok, gk, tgt_len)
    eos_table = torch.zeros(tok.vocab_size_out, dtype=torch.uint8, device=dev)   # random-init: keep rows running
    gathered = torch.empty((world * B, 1 + new), dtype=torch.int32, device=dev) if use_dist else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def one_step(record=False, gather=True):
        eng._enter()
        with torch.cuda.stream(eng.stream):
            if record:
                ev[0].record(eng.stream)
            enc = eng.encode_mel(eng.mel(audio))
            if record:
                ev[1].record(eng.stream)
            kv = eng.cross_kv(enc)
            if record:
                ev[2].record(eng.stream)
            tokens, n_out, _ = eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
            if record:
                ev[3].record(eng.stream)
        eng._leave()
        if use_dist and gather:
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

    runtime_env = {k: os.environ.get(k) for k in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG")}
    if args.headline_only:
        if rank == 0:
            print(json.dumps({"headline_only": True, "value": round(value, 1), "ms_per_step": round(ms_per_step, 2), "steps": args.steps,
                              "runtime_env": runtime_env}), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-stage milliseconds (one extra pass with events on the engine's stream; not the timed region) --------
    tokens, kv = one_step(record=True)
    fence()
    stage = {"mel_encoder_ms": round(ev[0].elapsed_time(ev[1]), 3), "cross_kv_ms": round(ev[1].elapsed_time(ev[2]), 3),
             "decode_ms": round(ev[2].elapsed_time(ev[3]), 3)}
    decode_us_per_step = stage["decode_ms"] * 1e3 / new

    # ---- diffusion stage (configs[2] / [3]): every rank refines ITS chunks as one denoiser batch, coordinates are
    # all-gathered next to the tokens -------------------------------------------------------------------
    aux = {"stage_ms": stage,
           # HIP runtime flags this process ran under (mapperatorinator_amd.configure_runtime() at the top of this file)
           "runtime_env": runtime_env, "runtime_configure": RUNTIME}
    if not args.no_dit:
        from mapperatorinator_amd.dit import BandMask, DiTHIP, create_diffusion
        from mh_testing import DIT_PRESETS, random_dit_state_dict, synthetic_dit_inputs
        depth, hidden, heads = DIT_PRESETS["DiT-S"]
        dit = DiTHIP(random_dit_state_dict(depth, hidden, seed=0), depth, hidden, heads, device=dev)
        Tq = 128
        parts = [synthetic_dit_inputs(Tq, seed=rank * B + b) for b in range(B)]
        z = torch.cat([p[0][:1] for p in parts] + [p[0][1:] for p in parts]).to(dev)      # [cond rows | null rows]
        c = torch.cat([p[1][:1] for p in parts] + [p[1][1:] for p in parts]).to(dev)
        y = torch.cat([p[2][:1] for p in parts] + [p[2][1:] for p in parts]).to(dev)
        diff = create_diffusion([100, 0, 0, 0, 0, 0, 0, 0, 0, 0], noise_schedule="squaredcos_cap_v2", diffusion_steps=1000)
        coords_all = torch.empty((world * B, 2, Tq), dtype=torch.float32, device=dev) if use_dist else None

        def dit_stage(zz, cc, yy, gather):
            kw = dict(c=cc, y=yy, cfg_scale=1.0, attn_mask=BandMask(Tq, 128))
            noise = torch.randn(100, *zz.shape, device=dev)
            out = diff.p_sample_loop(dit.forward_with_cfg, zz.shape, zz, model_kwargs=kw, step_noise=noise)
            coords = out[: zz.shape[0] // 2].contiguous()
            if gather and use_dist:
                dist.all_gather_into_tensor(coords_all, coords)
            return coords

        def timed(fn, reps):
            fn()
            fence()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            fence()
            return (time.perf_counter() - t) / reps

        dt_one = timed(lambda: dit_stage(z[[0, B]], c[[0, B]], y[[0, B]], False), 3)     # the reference's shape: one chunk
        dt_all = timed(lambda: dit_stage(z, c, y, True), 2)
        if use_dist:
            tt = torch.tensor([dt_all], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_all = float(tt[0])
        flops = dit_flops_per_step(depth, hidden, 2 * B, Tq) * 100
        aux["diffusion"] = {
            "dit": "DiT-S fp32 (exact-f32 MFMA), Tq=128, 100-step DDPM as one replayed hipGraph",
            "one_chunk": {"ms_per_100_steps": round(dt_one * 1e3, 2), "steps_per_s_per_chunk": round(100 / dt_one, 1)},
            "batched": {"chunks_per_gpu": B, "denoiser_batch": 2 * B, "ms_per_100_steps": round(dt_all * 1e3, 2),
                        "steps_per_s_per_chunk": round(100 * B / dt_all, 1),
                        "tflops": round(flops / dt_all / 1e12, 1), "frac_of_f32_mfma_peak": round(flops / dt_all / 1e12 / F32_MFMA_PEAK_TF, 3),
                        "frac_of_bf16_pipe": round(3 * flops / dt_all / 1e12 / BF16_MFMA_PEAK_TF, 4),
                        "frac_note": "the big GEMMs run as three bf16 MFMA passes (bf16 x 3): the pipe they occupy is the bf16 "
                                     "one, so the honest fraction is 3 x fp32-equivalent flops / 2500 TFLOP/s",
                        "coords_all_gather": bool(use_dist)},
        }
        aux["diffusion_steps_per_s_per_chunk"] = aux["diffusion"]["batched"]["steps_per_s_per_chunk"]
        if not args.no_config5 and rank == 0 and world == 1:   # (single-GPU extras: the other ranks of a multi-GPU run would only wait)
            # BASELINE configs[4]: DiT-B (osu_diffusion/utils/models.py:392), the same 32 chunks as one denoiser batch
            db, hb, nb = DIT_PRESETS["DiT-B"]
            dit_b = DiTHIP(random_dit_state_dict(db, hb, seed=0), db, hb, nb, device=dev)

            def dit_b_stage():
                kw = dict(c=c, y=y, cfg_scale=1.0, attn_mask=BandMask(Tq, 128))
                return diff.p_sample_loop(dit_b.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=torch.randn(100, *z.shape, device=dev))
            dt_b = timed(dit_b_stage, 1)
            fl_b = dit_flops_per_step(db, hb, 2 * B, Tq) * 100

            def dit_b_one():      # the reference's own call shape (diffusion_pipeline.py:243-252): ONE chunk per p_sample_loop, CFG batch 2
                kw1 = dict(c=c[[0, B]], y=y[[0, B]], cfg_scale=1.0, attn_mask=BandMask(Tq, 128))
                return diff.p_sample_loop(dit_b.forward_with_cfg, z[[0, B]].shape, z[[0, B]], model_kwargs=kw1,
                                          step_noise=torch.randn(100, 2, *z.shape[1:], device=dev))
            dt_b1 = timed(dit_b_one, 3)
            fl_b1 = dit_flops_per_step(db, hb, 2, Tq) * 100
            aux["diffusion"]["one_chunk_dit_b"] = {
                "dit": "DiT-B (the released diffusion checkpoint's size, configs/diffusion/v1.yaml:11) fp32 exact-f32 MFMA, one chunk, Tq=128",
                "ms_per_100_steps": round(dt_b1 * 1e3, 2), "steps_per_s_per_chunk": round(100 / dt_b1, 1),
                "tflops": round(fl_b1 / dt_b1 / 1e12, 1), "frac_of_f32_mfma_peak": round(fl_b1 / dt_b1 / 1e12 / F32_MFMA_PEAK_TF, 3)}
            aux["config5_dit_b"] = {
                "dit": "DiT-B fp32 semantics (big GEMMs bf16 x 3 on the matrix cores), Tq=128, 100-step DDPM, one replayed hipGraph",
                "chunks": B, "denoiser_batch": 2 * B, "ms_per_100_steps": round(dt_b * 1e3, 2),
                "steps_per_s_per_chunk": round(100 * B / dt_b, 1),
                "roofline_step": {"bound": "mfma", "alg_flops_per_step": fl_b / 100, "achieved": round(fl_b / dt_b / 1e12, 1),
                                  "unit": "TFLOP/s", "peak": BF16_MFMA_PEAK_TF,
                                  "frac": round(3 * fl_b / dt_b / 1e12 / BF16_MFMA_PEAK_TF, 4),
                                  "how": "fp32-equivalent flops x 3 (three bf16 MFMA passes per product) / dense bf16 peak"}}
            del dit_b
            # the reduced-precision mode configs[4] names (MhDiTConfig.operand_dtype = MH_BF16): block GEMMs and attention on
            # bf16 operands, one MFMA pass; parity = error bounds vs the fp32 reference golden (tests/test_gpu_dit.py)
            lowp = {}
            for pname in ("DiT-S", "DiT-B"):
                dd, hh, nn = DIT_PRESETS[pname]
                dl = DiTHIP(random_dit_state_dict(dd, hh, seed=0), dd, hh, nn, device=dev, operand_dtype=torch.bfloat16)

                def lowp_stage():
                    kw = dict(c=c, y=y, cfg_scale=1.0, attn_mask=BandMask(Tq, 128))
                    return diff.p_sample_loop(dl.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=torch.randn(100, *z.shape, device=dev))
                dtl = timed(lowp_stage, 2)
                fl = dit_flops_per_step(dd, hh, 2 * B, Tq) * 100
                lowp[pname] = {"ms_per_100_steps": round(dtl * 1e3, 2), "steps_per_s_per_chunk": round(100 * B / dtl, 1),
                               "tflops": round(fl / dtl / 1e12, 1), "frac_of_bf16_mfma_peak": round(fl / dtl / 1e12 / BF16_MFMA_PEAK_TF, 4)}
                del dl
            aux["config5_dit_bf16_operands"] = dict(lowp, chunks=B, note="32 chunks x 100 DDPM steps, block GEMMs + attention on bf16 operands "
                                                    "(fp32 residual stream / LayerNorm / softmax / DDPM update); NOT the parity mode: eps within "
                                                    "5e-3 of scale and one p_sample step within 0.03 px of the fp32 reference golden")
            # ---- BASELINE configs[4] "fp8 MFMA": the same denoiser batch with the block GEMMs on MX-fp8 operands
            # (v_mfma_scale_f32_16x16x128_f8f6f4), and the T5 encoder + cross-K/V projection of the headline batch in the MX mode
            fp8 = {}
            for pname in ("DiT-S", "DiT-B"):
                dd, hh, nn = DIT_PRESETS[pname]
                d8 = DiTHIP(random_dit_state_dict(dd, hh, seed=0), dd, hh, nn, device=dev, operand_dtype="mx8")

                def mx_stage():
                    kw = dict(c=c, y=y, cfg_scale=1.0, attn_mask=BandMask(Tq, 128))
                    return diff.p_sample_loop(d8.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=torch.randn(100, *z.shape, device=dev))
                dt8 = timed(mx_stage, 2)
                fl = dit_flops_per_step(dd, hh, 2 * B, Tq) * 100
                gemm_fl = 2 * B * (dd * 2.0 * Tq * 12 * hh * hh) * 100                    # the four block projections (the MX part)
                fp8[pname] = {"ms_per_100_steps": round(dt8 * 1e3, 2), "steps_per_s_per_chunk": round(100 * B / dt8, 1),
                              "tflops": round(fl / dt8 / 1e12, 1), "frac_of_mx8_mfma_peak": round(gemm_fl / dt8 / 1e12 / MX8_MFMA_PEAK_TF, 4),
                              "vs_bf16_operands": round(lowp[pname]["ms_per_100_steps"] / (dt8 * 1e3), 3)}
                del d8
            aux["config5_fp8"] = {"dit": dict(fp8, chunks=B, note="32 chunks x 100 DDPM steps, the four block projections of every DiT block on MX-fp8 "
                                              "operands (OCP e4m3 + E8M0 per 32 k; LayerNorm-modulate writes the operand, attention output and GELU "
                                              "hidden quantised by a pass of their own), attention on bf16 operands; frac_of_mx8_mfma_peak = flops "
                                              "of those projections / whole-loop time / 5 PFLOP/s; NOT a parity mode: error bounds in "
                                              "tests/test_gpu_dit.py::test_mx8_operand_mode_error_bounds")}
        # configs[2] end to end: T5 path + diffusion refine of the same chunks, whole job
        aux["config3_end_to_end"] = {"chunks": world * B, "seconds": round(ms_per_step / 1e3 + dt_all, 4),
                                     "chunks_per_s": round(world * B / (ms_per_step / 1e3 + dt_all), 2),
                                     "t5_ms": round(ms_per_step, 2), "diffusion_ms": round(dt_all * 1e3, 2)}

    if rank != 0:
        if use_dist:
            dist.barrier()                  # rank 0 prints its line (its single-rank extras take a few seconds), then every
            dist.destroy_process_group()    # rank leaves the group together
        return

    # ---- roofline of the dominant kernel (HBM-bound decode cross-attention) ------------------------------------
    es = 2 if args.dtype == "bf16" else 4
    bytes_per_row = dims.n_heads * SRC_FRAMES * 64 * 2 * es                      # K and V of one layer, one chunk
    # (a) in situ: device-side wall-clock stamps of every launch of one extra decode pass
    n_chains = lib.mh_t5_decode_chains(B)
    rows_per_launch = -(-B // n_chains)
    ring, L = 64, dims.n_dec_layers
    in_situ_us = None
    if not args.no_extras:
        tbuf = torch.zeros((n_chains, ring, L, 2), dtype=torch.int64, device=dev)
        tbuf[..., 0] = -1                                                          # = UINT64_MAX for atomicMin
        _lib.check(lib.mh_t5_decode_timing(tbuf.data_ptr(), ring), "mh_t5_decode_timing")
        try:
            one_step(gather=False)      # (rank 0 only: no collective in here)
            torch.cuda.synchronize(dev)
        finally:
            _lib.check(lib.mh_t5_decode_timing(None, 0), "mh_t5_decode_timing")
        t = tbuf.cpu().numpy().view("uint64")
        okm = t[..., 1] > t[..., 0]
        khz = lib.mh_wall_clock_khz() or 100000                                       # 100 MHz on gfx950
        if okm.any():
            in_situ_us = float((t[..., 1][okm] - t[..., 0][okm]).astype("float64").mean()) / (khz / 1e3)
    # (b) stand-alone probe: the same kernel back to back at the full batch on the engine's stream (HIP events)
    ms = C.c_float(0.0)
    ws = eng._workspace("dec", lib.mh_t5_decode_workspace_bytes(C.byref(eng.packed.cfg), B))
    rc = lib.mh_t5_cross_attn_probe(C.byref(eng.packed.cfg), C.byref(eng.packed.w), kv.data_ptr(), B, 20 * L, C.byref(ms),
                                    ws.data_ptr(), ws.numel(), eng.stream.cuda_stream)
    _lib.check(rc, "mh_t5_cross_attn_probe")
    probe_gbs = B * bytes_per_row / (ms.value * 1e-3) / 1e9
    # (c) the whole token step against the bytes it must move
    step_bytes = decode_step_bytes(dims, B, tok.vocab_size_out, es, t_avg=(1 + new) / 2.0)
    step_gbs = step_bytes / (decode_us_per_step * 1e-6) / 1e9
    use_us = in_situ_us if in_situ_us else ms.value * 1e3
    use_rows = rows_per_launch if in_situ_us else B
    achieved = use_rows * bytes_per_row / (use_us * 1e-6) / 1e9
    pmc_record = {}
    pmc_file = "r06_pmc_cross_attn.json"      # the latest counter passes (tools/profile_bench_pmc.sh: the headline loop of this command; earlier rounds: r04 / r05_pmc_cross_attn.json)
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pmc_file)) as fh:
            pmc_record = json.load(fh)
        pmc_record["rows_per_launch"] = 32 // int(pmc_record.get("step", {}).get("decode_chains", 1))
    except (OSError, ValueError):
        pmc_record = {}
    roofline = {
        "bound": "hbm", "kernel": "dec_cross_attn_q_kernel (decode cross-attention over the encoder K/V incl. its query projection)",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        # PMC FETCH_SIZE / WRITE_SIZE need rocprofv3 around the process: not measurable from inside this run.  The committed
        # record of the counter passes of THIS command in the timed configuration (two 16-row chains) is quoted instead.
        "traffic": pmc_record.get("hbm_bytes_per_launch") if pmc_record.get("rows_per_launch") == use_rows else None,
        "traffic_source": ("profiles/" + pmc_file + " / r06_pmc_hbm_traffic.txt: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                           "(separate passes) around `bench.py --headline-only` (this command's timed loop; a counter pass around the whole "
                           "default command did not finish in 42 minutes), builder-run, two 16-row decode chains fed by one launcher thread "
                           "(MH_DECODE_LAUNCH_THREADS=0); bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 per launch (gfx950 correction of "
                           "MI355X_MICROARCH.md); the counter passes cannot run inside this process"),
        "traffic_over_algorithmic": (round(pmc_record["hbm_bytes_per_launch"] / (use_rows * bytes_per_row), 4)
                                     if pmc_record.get("rows_per_launch") == use_rows else None),
        "step_traffic": pmc_record.get("step"),     # all decode kernels of a token step: PMC bytes vs SURVEY 8d bytes
        "how": ("in situ: mean (last workgroup end - first workgroup start) over the launches of one extra decode pass, "
                "device wall clock; the other chain's kernels run beside it" if in_situ_us else "stand-alone probe"),
        "rows_per_launch": use_rows, "alg_bytes_per_launch": use_rows * bytes_per_row, "us_per_launch": round(use_us, 2),
        "launches_per_token_step": L * n_chains, "decode_chains": n_chains,
        "probe": {"rows_per_launch": B, "alg_bytes_per_launch": B * bytes_per_row, "us_per_launch": round(ms.value * 1e3, 2),
                  "achieved": round(probe_gbs, 1), "frac": round(probe_gbs / HBM_PEAK_GBS, 4),
                  "how": "the same kernel back to back at the full batch, HIP events on the engine's stream"},
        "step": {"alg_bytes_per_token_step": int(step_bytes), "us_per_token_step": round(decode_us_per_step, 1),
                 "achieved": round(step_gbs, 1), "frac": round(step_gbs / HBM_PEAK_GBS, 4),
                 "how": "SURVEY 8d bytes (decoder weights + lm_head + B x (cross K/V + self K/V at the mean position)) / "
                        "(decode stage time / new tokens)"},
    }

    # ---- the same batch with the token steps streaming the e4m3 copy of the cross K / V (BASELINE configs[4] direction) ----
    if not args.no_extras and args.dtype == "bf16" and rank == 0:
        def fp8_step():
            eng._enter()
            with torch.cuda.stream(eng.stream):
                kv_ = eng.cross_kv(eng.encode_mel(eng.mel(audio)))
                kv8 = eng.cross_kv_fp8(kv_)
                t_, _, _ = eng.decode(kv_, prompt, None, eos_table, sp, poll_every=64, kv_fp8=kv8)
            eng._leave()
            return t_
        fp8_step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            t8 = fp8_step()
        torch.cuda.synchronize(dev)
        dt8 = (time.perf_counter() - t0) / args.steps
        aux["fp8_cross_kv"] = {"tokens_per_s": round(int((t8[:, 1:] != 0).sum().item()) / dt8, 1), "ms_per_step": round(dt8 * 1e3, 2),
                               "note": "same workload, the token steps stream an OCP e4m3 copy of the cross-attention K / V (one "
                                       "scale per layer, k|v, row, head; quantisation pass inside the timed step); NOT the parity "
                                       "mode and not `value`",
                               "same_tokens_as_bf16_kv": round(float((t8 == tokens).float().mean().item()), 4)}

    # ---- 64 chunks per call (the engine's maximum: two 32-row chains): more rows under the same dependent-kernel chain ----
    if not args.no_extras and rank == 0 and B == 32:
        try:
            audio64 = torch.cat([audio, audio.flip(0)], 0)
            prompt64 = torch.cat([prompt, prompt], 0)

            def step64():
                eng._enter()
                with torch.cuda.stream(eng.stream):
                    kv_ = eng.cross_kv(eng.encode_mel(eng.mel(audio64)))
                    t_, _, _ = eng.decode(kv_, prompt64, None, eos_table, sp, poll_every=64)
                eng._leave()
                return t_
            step64()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(2):
                t64 = step64()
            torch.cuda.synchronize(dev)
            dt64 = (time.perf_counter() - t0) / 2
            aux["rows64"] = {"tokens_per_s": round(int((t64[:, 1:] != 0).sum().item()) / dt64, 1), "ms_per_step": round(dt64 * 1e3, 2),
                             "chunks_per_call": 64, "first_half_same_tokens_as_headline": bool(torch.equal(t64[:B], tokens)),
                             "note": "NOT `value` (BASELINE configs[1] is 32 chunks per call): the same model and chunk shape with 64 chunks per "
                                     "call -- two 32-row decode chains.  The token step is a chain of dependent kernels whose time grows "
                                     "slowly with the rows it carries (profiles/r06_small_batch_decode.txt), so a caller with >= 64 windows "
                                     "in hand gets more tokens per second at ~1.75 x the per-token latency"}
            del audio64, t64
        except Exception as e:  # noqa: BLE001
            aux["rows64"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- the reference's default USER settings: sampling with temperature 0.9 / top_p 0.9 (configs/inference/v32.yaml:12-13) ----
    if not args.no_extras and rank == 0:
        sp_s, _ = build_sampling(tok, dict(gk, do_sample=True, top_p=0.9, temperature=0.9, seed=1), tgt_len)

        def sample_step():
            eng._enter()
            with torch.cuda.stream(eng.stream):
                kv_ = eng.cross_kv(eng.encode_mel(eng.mel(audio)))
                t_, _, _ = eng.decode(kv_, prompt, None, eos_table, sp_s, poll_every=64)
            eng._leave()
            return t_
        sample_step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ts_ = sample_step()
        torch.cuda.synchronize(dev)
        dts = (time.perf_counter() - t0) / args.steps
        aux["sampling_top_p"] = {"tokens_per_s": round(int((ts_[:, 1:] != 0).sum().item()) / dts, 1), "ms_per_step": round(dts * 1e3, 2),
                                 "note": "same workload with do_sample, temperature 0.9, top_p 0.9 (the reference's default user settings); "
                                         "thresholds and the draw on the device, per row"}

    # ---- configs[0] on the GPU, and the end-to-end rate through the reference-shaped boundary ---------------------
    if not args.no_extras:
        e2e_audio = audio_host                                                       # HOST tensors, as server.py:86 receives them
        mk = dict(inputs=e2e_audio, decoder_input_ids=torch.full((B, 1), tok.sos_id, dtype=torch.long))
        # model_generate builds its EOS set from the tokenizer; random-init rows may stop early, so the rate is tokens / s
        model_generate(model, tok, mk, gk)                                          # warm-up
        t0 = time.perf_counter()
        ids, st = model_generate(model, tok, mk, gk)
        dt = time.perf_counter() - t0
        aux["e2e_model_generate"] = {"tokens_per_s": round(st["generated_tokens"] / dt, 1), "seconds": round(dt, 4),
                                     "generated_tokens": st["generated_tokens"],
                                     "includes": "H2D of 32 x 640 KB audio, mel, encoder, cross-KV, decode, D2H of the ids, host "
                                                 "bookkeeping of mapperatorinator_amd.server.model_generate (EOS set active)"}
        d1 = T5_PRESETS["small"]
        m1 = MapperatorinatorHIP(random_t5_state_dict(d1, tok.vocab_size_in, tok.vocab_size_out, seed=0, lm_head_gain=6.0), d1,
                                 vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out, src_seq_len=SRC_FRAMES,
                                 tgt_seq_len=512, dtype=torch.float32, device=dev)
        a1 = synthetic_audio(1, N_SAMPLES, seed=0).to(dev)
        p1 = torch.full((1, 1), tok.sos_id, dtype=torch.int32, device=dev)
        sp1, _ = build_sampling(tok, dict(gk, max_length=129), 512)

        def cfg1():
            e1 = m1.engine
            e1._enter()
            with torch.cuda.stream(e1.stream):
                e1.decode(e1.cross_kv(e1.encode_mel(e1.mel(a1))), p1, None, eos_table, sp1, poll_every=64)
            e1._leave()
            torch.cuda.synchronize(dev)
        cfg1()
        t0 = time.perf_counter()
        for _ in range(3):
            cfg1()
        dt1 = (time.perf_counter() - t0) / 3
        aux["config1"] = {"workload": "osuT5-small fp32 (the bit-exact storage mode), 1 x 10 s chunk, 128 greedy tokens, "
                                      "mel+encoder+decode (BASELINE configs[0])",
                          "gpu_tokens_per_s": round(128 / dt1, 1), "gpu_ms": round(dt1 * 1e3, 2)}
        del m1

    # ---- the Whisper-family backbones of the released checkpoints at their own chunk sizes: V32 'OliBomby/varwhisper-small' (2048
    # log-mel frames), V30 / V31 'Tiger14n/ropewhisper-small' (4096 frames, 80 mels + 3 x 128 conditioning channels into conv1),
    # V28 / V29 'openai/whisper-small' (1024 nnAudio-mel frames behind encoder_embedder; library arch 2) -- batch 32, greedy.
    # V29 + DiT-B is the one released pairing whose inference config runs BOTH stages (configs/inference/v29.yaml:7-8 with
    # generate_positions inherited true): its chunks/s is the T5-side time of the V29 line + the DiT-B batch of aux.config5_dit_b.
    if not args.no_extras and not args.no_config5 and world == 1:
        try:
            import importlib.util
            spec_sb = importlib.util.spec_from_file_location("small_batch_decode", os.path.join(ROOT, "tools", "small_batch_decode.py"))
            sbd = importlib.util.module_from_spec(spec_sb)
            spec_sb.loader.exec_module(sbd)
            fam_lines = {}
            for mname, release in (("varwhisper-small", "V32"), ("ropewhisper-small", "V30 / V31"), ("whisper-small", "V28 / V29")):
                mw, tokw, dv, fr = sbd.build(mname, tgt_len, dev)
                ew = mw.engine
                aw = synthetic_audio(B, (fr - 1) * 128, seed=5).to(dev)
                pw = torch.full((B, 1), tokw.sos_id, dtype=torch.int32, device=dev)
                spw, _ = build_sampling(tokw, dict(gk), tgt_len)
                eos_w = torch.zeros(tokw.vocab_size_out, dtype=torch.uint8, device=dev)
                rbw = None
                if ew.packed.cond_channels:      # (the V30 wiring: difficulty / mapper / song-position vectors as conv1 channels)
                    rbw = torch.randn(B, ew.packed.cond_channels, generator=torch.Generator().manual_seed(1)).to(dev)

                def vw_step():
                    ew._enter()
                    with torch.cuda.stream(ew.stream):
                        t_, _, _ = ew.decode(ew.cross_kv(ew.encode_mel(ew.mel(aw), row_bias=rbw)), pw, None, eos_w, spw, poll_every=64)
                    ew._leave()
                    return t_
                vw_step()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    tw = vw_step()
                torch.cuda.synchronize(dev)
                dtw = (time.perf_counter() - t0) / args.steps
                fam_lines[mname] = {"release": release, "kind": ew.kind, "mel_frames": fr, "encoder_positions": ew.src_len,
                                    "tokens_per_s": round(int((tw[:, 1:] != 0).sum().item()) / dtw, 1), "ms_per_step": round(dtw * 1e3, 2)}
                del mw, ew
                torch.cuda.empty_cache()
            aux["whisper_family"] = dict(fam_lines, workload=f"backbone-small dims (d 768, 12 + 12 layers) bf16, batch={B} chunks of the release's own "
                                         f"length, {new} greedy tokens per chunk, mel + conv front-end + encoder + cross-K/V + AR decode, "
                                         "synthetic audio, random-init weights in the reference's parameter layout")
            if "config5_dit_b" in aux:
                t5s, dbs = fam_lines["whisper-small"]["ms_per_step"] / 1e3, aux["config5_dit_b"]["ms_per_100_steps"] / 1e3
                aux["released_pairing_v29_dit_b"] = {
                    "workload": f"BASELINE configs[2] on a RELEASED pairing: 'openai/whisper-small' event model (V29) + DiT-B (osu-diffusion-v2), "
                                f"{B} chunks: event tokens, then 100-step DDPM refine of all chunks as one denoiser batch",
                    "t5_ms": round(t5s * 1e3, 2), "diffusion_ms": round(dbs * 1e3, 2), "chunks_per_s": round(B / (t5s + dbs), 2),
                    "event_tokens_per_s": fam_lines["whisper-small"]["tokens_per_s"],
                    "diffusion_steps_per_s_per_chunk": aux["config5_dit_b"]["steps_per_s_per_chunk"]}
            # ---- the reference's DEFAULT call shape (configs/inference/default.yaml:54 `parallel: false`): one window per call ----
            for mname in ("t5-base", "varwhisper-small"):
                r = sbd.run(mname, new_tokens=256, device=str(dev))
                for key in ("b1", "b2_cfg"):
                    r[key]["workload"] = r["workload"]
                    aux.setdefault("decode_" + key, {})[mname] = r[key]
            # beam search as the reference's timing pass runs it (2 beams, super_timing_generator.py:28): the per-token bookkeeping in ONE
            # kernel (mh_beam_step) against the torch-op form of round 5
            spec_bb = importlib.util.spec_from_file_location("beam_bench", os.path.join(ROOT, "tools", "beam_bench.py"))
            bbm = importlib.util.module_from_spec(spec_bb)
            spec_bb.loader.exec_module(bbm)
            aux["beam2"] = {"one_window": bbm.run(chunks=1, beams=2, device=str(dev)), "eight_windows": bbm.run(chunks=8, beams=2, device=str(dev))}
            aux["decode_b1"]["cpu_reference_tokens_per_s"] = 44.0      # the unmodified reference at this shape, other host
            aux["decode_b1"]["note"] = ("one row: the token step is 74 dependent kernels of 5-13 us (profiles/r06_small_batch_decode.txt); "
                                        "roofline_step = SURVEY 8d bytes of one step / measured step time / 8 TB/s")
        except Exception as e:   # an auxiliary figure must never cost the bench line -- but it must not vanish silently either
            aux.setdefault("errors", []).append(f"whisper_family / small batch: {e!r}")
            print(f"whisper-family pass failed: {e!r}", file=sys.stderr)

    # ---- BASELINE configs[4] "fp8 MFMA", T5 side: mel + encoder + cross-K/V of the headline batch with MX-fp8 operands ----
    if not args.no_extras and not args.no_config5 and world == 1 and args.dtype == "bf16":
        try:
            enc_lines = {}
            for sz in ("base", "large"):
                dz = T5_PRESETS[sz]
                per = {}
                for mode in (None, "mx8"):
                    mm = MapperatorinatorHIP(random_t5_state_dict(dz, tok.vocab_size_in, tok.vocab_size_out, seed=0, lm_head_gain=6.0), dz,
                                             vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out, src_seq_len=SRC_FRAMES,
                                             tgt_seq_len=tgt_len, dtype=torch.bfloat16, device=dev, enc_operand_dtype=mode)
                    em = mm.engine

                    def enc_step():
                        em._enter()
                        with torch.cuda.stream(em.stream):
                            kv_ = em.cross_kv(em.encode_mel(em.mel(audio)))
                        em._leave()
                        return kv_
                    enc_step()
                    torch.cuda.synchronize(dev)
                    # median of per-iteration times: ONE stalled launch in a 5-iteration average made this line read 20 ms for a
                    # 14 ms stage once (round 5; the same build measures 12.5 ms three times in a row in tools/mx8_encoder_pmc.py)
                    its = []
                    for _ in range(7):
                        t0 = time.perf_counter()
                        enc_step()
                        torch.cuda.synchronize(dev)
                        its.append(time.perf_counter() - t0)
                    per[mode or "bf16"] = sorted(its)[len(its) // 2]
                    del mm, em
                Lr = B * SRC_FRAMES
                gemm_fl = 2.0 * Lr * (dz.n_enc_layers * (4 * dz.d_model * dz.inner + 3 * dz.d_model * dz.d_ff) + dz.n_dec_layers * 2 * dz.inner * dz.d_model)
                enc_lines[sz] = {"bf16_ms": round(per["bf16"] * 1e3, 2), "mx8_ms": round(per["mx8"] * 1e3, 2), "speedup": round(per["bf16"] / per["mx8"], 3),
                                 "projection_tflops_mx8": round(gemm_fl / per["mx8"] / 1e12, 1),
                                 "frac_of_mx8_mfma_peak": round(gemm_fl / per["mx8"] / 1e12 / MX8_MFMA_PEAK_TF, 4)}
            aux.setdefault("config5_fp8", {})["t5_encoder"] = dict(
                enc_lines, chunks=B, note="mel + encoder + cross-K/V projection of 32 chunks; mx8 = MhT5Config.enc_operand_dtype MH_MX8 (block "
                "projections + cross-K/V projection on MX-fp8 operands, attention bf16); frac = flops of those projections / whole-stage time "
                "(mel, attention, norms and quantiser passes included) / 5 PFLOP/s; NOT a parity mode: teacher-forced agreement against the fp32 "
                "reference goldens in tests/test_gpu_t5.py::test_mx8_encoder_teacher_forced_on_the_reference_fp32_run")
        except Exception as e:
            aux.setdefault("errors", []).append(f"config5_fp8.t5_encoder: {e!r}")
            print(f"config 5 fp8 encoder pass failed: {e!r}", file=sys.stderr)

    # ---- BASELINE configs[4]: whole 3-minute songs, KV-cached, through the window scheduler (tools/long_song_bench.py) ----
    if not args.no_extras and not args.no_config5 and world == 1:
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("long_song_bench", os.path.join(ROOT, "tools", "long_song_bench.py"))
            lsb = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(lsb)
            large_runs = lsb.run("large", songs=32, windows=18, fp8_kv=(False, True), device=str(dev), collect_tokens=True)
            large_bf16_tokens = large_runs[0].pop("_tokens")
            large_runs[1].pop("_tokens", None)
            aux["config5_long_songs"] = {
                "large": large_runs,
                "base": lsb.run("base", songs=32, windows=18, fp8_kv=(False,), device=str(dev)),
                "note": "32 songs x 18 windows (3 min each): every window encoded up front, its cross K/V resident in HBM, wave w "
                        "decodes window w of all songs; window w's prompt carries the last 32 tokens of window w-1"}
            # configs[4] as ONE workload (tools/config5_full.py): the same songs with MX-fp8 encoder operands + e4m3 cross K/V, the
            # agreement of its tokens with the bf16 run above, then DiT-B (MX-fp8 projections) over every window + its px error vs fp32
            try:
                spec5 = importlib.util.spec_from_file_location("config5_full", os.path.join(ROOT, "tools", "config5_full.py"))
                c5 = importlib.util.module_from_spec(spec5)
                spec5.loader.exec_module(c5)
                torch.cuda.empty_cache()
                aux["config5_full"] = c5.run(songs=32, windows=18, device=str(dev), bf16_tokens=large_bf16_tokens, bf16_line=dict(large_runs[0]))
            except Exception as e:
                aux.setdefault("errors", []).append(f"config5_full: {e!r}")
                print(f"config 5 full pass failed: {e!r}", file=sys.stderr)
        except Exception as e:
            aux.setdefault("errors", []).append(f"config5_long_songs: {e!r}")
            print(f"config 5 long-song pass failed: {e!r}", file=sys.stderr)

    # ---- the headline under the OTHER graph-replay setting of the HIP runtime (a child process: the runtime reads the flag once) ----
    if not args.no_extras and not args.no_runtime_ab and world == 1:
        try:
            import subprocess
            var = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
            other = "1" if os.environ.get(var, "1") == "0" else "0"
            env = dict(os.environ, **{var: other})
            torch.cuda.synchronize(dev)
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--headline-only", "--steps", str(args.steps), "--warmup", "1",
                                 "--size", args.size, "--dtype", args.dtype, "--batch", str(B), "--new-tokens", str(new)],
                                env=env, capture_output=True, text=True, timeout=600)
            child = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
            aux["runtime_flag_ab"] = {
                "this_process": {var: os.environ.get(var), "value": round(value, 1), "ms_per_step": round(ms_per_step, 2)},
                "child_process": {var: child["runtime_env"][var], "value": child["value"], "ms_per_step": child["ms_per_step"]},
                "note": "same workload, same build; '0' = classic per-node graph submission (mapperatorinator_amd.configure_runtime(), "
                        "explicit opt-in), '1' / unset = ROCm 7.2.0's default captured-AQL-packet replay.  Tokens are identical"}
        except Exception as e:
            aux.setdefault("errors", []).append(f"runtime_flag_ab: {e!r}")
            print(f"runtime flag A/B failed: {e!r}", file=sys.stderr)

    cpu = None
    if not args.no_cpu_baseline and world == 1:   # the CPU port is timed on rank 0 at N=1 only
        cpu = cpu_baseline(args.size, vocab)
        for key, fn in (("config1", lambda: cpu_config1(vocab)), ("diffusion", cpu_dit_baseline)):
            if key in aux:
                try:
                    aux[key]["cpu_baseline"] = fn()
                except Exception as e:
                    aux.setdefault("errors", []).append(f"cpu_baseline[{key}]: {e!r}")
                    print(f"cpu baseline for {key} failed: {e!r}", file=sys.stderr)

    line = {
        "metric": "beatmap event-tokens/sec (mel + osuT5 encoder + greedy AR decode), whole job",
        "value": round(value, 1), "unit": "event-tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic 16 kHz audio (noise + tones), random-init weights",
        "config": {"workload": f"osuT5-{args.size} {args.dtype}, batch={B} x 10 s chunks per GPU, {new} greedy tokens "
                               f"per chunk, mel+encoder+cross-KV+AR decode (BASELINE configs[1])",
                   "chunks_per_gpu": B, "new_tokens": new, "src_frames": SRC_FRAMES, "vocab": tok.vocab_size_out,
                   "parallelism": f"chunk-sharded x{world}, all_gather of token streams (+ diffusion coordinates in aux)"},
        "rccl_ranks": dist.get_world_size() if use_dist else 0,   # size of the RCCL group the token all_gather ran over (0: no group)
        "roofline": roofline, "cpu_baseline": cpu, "aux": aux,
    }
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
