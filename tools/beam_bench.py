"""Beam search throughput (`num_beams = 2`: the reference's timing pass, super_timing_generator.py:28; cache reorder per step,
inference/cache_utils.py:16-20): G windows x 2 beams through mapperatorinator_amd.beam.beam_search on osuT5-base bf16.
Prints one JSON line: tokens/s of the returned hypotheses and ms per beam step.    python tools/beam_bench.py [--chunks 1] [--beams 2]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(chunks=1, beams=2, new_tokens=128, device="cuda:0", model_tuple=None):
    from mapperatorinator_amd.server import build_sampling
    from mh_testing import synthetic_audio_varied
    import importlib.util
    spec = importlib.util.spec_from_file_location("small_batch_decode", os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_batch_decode.py"))
    sbd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sbd)
    dev = torch.device(device)
    tgt = 1 + new_tokens
    model, tok, dims, frames = model_tuple or sbd.build("t5-base", tgt, dev)
    eng = model.engine
    audio = synthetic_audio_varied(chunks, (frames - 1) * 128, seed=5).to(dev)
    prompt = torch.full((chunks, 1), tok.sos_id, dtype=torch.long)
    gk = dict(do_sample=False, num_beams=beams, max_length=tgt, temperature=1.0, context_type="map", pad_token_id=0)
    sp, eos = build_sampling(tok, gk, tgt)
    eos = []          # random-init weights: keep every hypothesis running to max_length
    res = {}
    outs = {}
    for name, uk in (("beam_step_kernel", None), ("torch_op_bookkeeping", False)):
        times = []
        for r in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            out = eng.generate_beam(audio, prompt, None, eos, sp, beams, use_kernel=uk)
            torch.cuda.synchronize(dev)
            times.append(time.perf_counter() - t0)
        dt = sorted(times[1:])[0]
        n = int(out["tokens"].shape[1]) - 1
        outs[name] = out["tokens"]
        res[name] = {"seconds": round(dt, 4), "ms_per_beam_step": round(dt * 1e3 / max(n, 1), 3), "tokens_per_s": round(chunks * n / dt, 1)}
    res["same_ids"] = bool(torch.equal(outs["beam_step_kernel"], outs["torch_op_bookkeeping"]))
    res["workload"] = (f"osuT5-base bf16, {chunks} window(s) x {beams} beams, {n} steps: mel + encoder + beam search (per token mh_t5_step -> "
                       "mh_beam_step -> mh_t5_reorder_cache; torch_op_bookkeeping = the ~40 ATen launches per token of round 5)")
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--beams", type=int, default=2)
    a = ap.parse_args()
    print(json.dumps(run(a.chunks, a.beams)))
