"""Experiment: overlap the encoder stage (mel -> encoder -> cross K/V) of batch i + 1 with the token loop of batch i (two HIP
streams).  The token loop is a chain of small dependent kernels that leaves most CUs idle; the encoder stage is 6 % of a batch.
    python tools/pipeline_bench.py [--steps 8] [--side-priority 0]"""
import argparse, importlib.util, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapperatorinator_amd
mapperatorinator_amd.configure_runtime()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--new-tokens", type=int, default=384)
    a = ap.parse_args()
    from mapperatorinator_amd.server import build_sampling
    from mh_testing import synthetic_audio_varied
    spec = importlib.util.spec_from_file_location("sbd", os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_batch_decode.py"))
    sbd = importlib.util.module_from_spec(spec); spec.loader.exec_module(sbd)
    dev = torch.device("cuda:0")
    B, new = 32, a.new_tokens
    tgt = 1 + new
    model, tok, dims, frames = sbd.build("t5-base", tgt, dev)
    eng = model.engine
    audio = synthetic_audio_varied(B, (frames - 1) * 128, seed=5).to(dev)
    prompt = torch.full((B, 1), tok.sos_id, dtype=torch.int32, device=dev)
    sp, _ = build_sampling(tok, dict(do_sample=False, num_beams=1, max_length=tgt, temperature=1.0, context_type="map", pad_token_id=0), tgt)
    eos_table = torch.zeros(tok.vocab_size_out, dtype=torch.uint8, device=dev)
    main_stream = eng.stream
    res = {}

    def enc_stage(stream):
        eng.stream = stream
        try:
            with torch.cuda.stream(stream):
                kv = eng.cross_kv(eng.encode_mel(eng.mel(audio)))
        finally:
            eng.stream = main_stream
        e = torch.cuda.Event()
        e.record(stream)
        return kv, e

    def sequential(n):
        for _ in range(n):
            kv, e = enc_stage(main_stream)
            with torch.cuda.stream(main_stream):
                tokens, _, _ = eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
        return tokens

    def pipelined(n, side):
        kv, e = enc_stage(side)
        for i in range(n):
            nxt = enc_stage(side) if i + 1 < n else None          # enqueued now, runs under the token loop below
            main_stream.wait_event(e)
            with torch.cuda.stream(main_stream):
                tokens, _, _ = eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
            if nxt:
                kv, e = nxt
        return tokens

    for name, fn in (("sequential", lambda n: sequential(n)),
                     ("pipelined_side_default_priority", lambda n: pipelined(n, torch.cuda.Stream(dev))),
                     ("pipelined_side_low_priority", lambda n: pipelined(n, torch.cuda.Stream(dev, priority=0))),
                     ("sequential_again", lambda n: sequential(n))):
        t_ref = fn(2)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        t = fn(a.steps)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / a.steps
        res[name] = {"ms_per_batch": round(dt * 1e3, 2), "tokens_per_s": round(B * new / dt, 1), "checksum": int(t.sum().item())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
