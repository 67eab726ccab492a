"""How the token loop scales with rows in flight and row chains (osuT5-base bf16, 1251 frames, cross K/V resident, 256 tokens):
    python tools/batch_scaling.py"""
import importlib.util, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapperatorinator_amd
mapperatorinator_amd.configure_runtime()
from mapperatorinator_amd import _lib  # noqa: E402


def main():
    from mapperatorinator_amd.server import build_sampling
    from mh_testing import synthetic_audio_varied
    spec = importlib.util.spec_from_file_location("sbd", os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_batch_decode.py"))
    sbd = importlib.util.module_from_spec(spec); spec.loader.exec_module(sbd)
    dev = torch.device("cuda:0")
    new = 256
    tgt = 1 + new
    model, tok, dims, frames = sbd.build("t5-base", tgt, dev)
    eng = model.engine
    sp, _ = build_sampling(tok, dict(do_sample=False, num_beams=1, max_length=tgt, temperature=1.0, context_type="map", pad_token_id=0), tgt)
    eos_table = torch.zeros(tok.vocab_size_out, dtype=torch.uint8, device=dev)
    for B, chains in ((16, 1), (32, 2), (48, 3), (64, 2), (64, 4)):
        old = _lib.set_option("decode_chains", chains)
        audio = synthetic_audio_varied(B, (frames - 1) * 128, seed=5).to(dev)
        prompt = torch.full((B, 1), tok.sos_id, dtype=torch.int32, device=dev)
        with eng.on_stream():
            kv = eng.cross_kv(eng.encode_mel(eng.mel(audio)))
            for _ in range(2):
                eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(3):
                eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
            torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / 3
        print(json.dumps({"rows": B, "chains": chains, "us_per_step": round(dt / new * 1e6, 1), "decode_tokens_per_s": round(B * new / dt, 1)}), flush=True)
        _lib.set_option("decode_chains", old)
        del kv


if __name__ == "__main__":
    main()
