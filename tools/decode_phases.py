#!/usr/bin/env python
"""Where does the time inside the decode kernels go?  Runs the bench workload (osuT5-base, bf16) on the PROFILING build
of the library (`make prof`: in-kernel phase stamps, thread 0 of workgroup 0) and prints, per kernel, the average
shader-clock ticks from the kernel's first instruction to every stamp.  The cross-attention kernel is also timed in wall
clock (mh_t5_decode_timing), which calibrates ticks per microsecond.

  python tools/decode_phases.py [--batch 16] [--chains 1] [--tokens 192]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MAPPERHIP_LIB"] = os.path.join(ROOT, "mapperatorinator_amd", "lib", "libmapperhip_prof.so")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

STAMPS = {
    "gemv": ["loads issued", "products done (this wave)", "all waves done", "stores issued"],
    "self": ["q/k/v projected", "cached keys attended", "partials merged"],
    "cross": ["row normalised", "query projected", "keys streamed", "partials merged"],
}
EPI = {0: "STORE", 1: "QKV", 2: "GEGLU(wi)", 3: "RESID(o/co/wo)", 4: "LOGITS"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--chains", type=int, default=1)
    ap.add_argument("--tokens", type=int, default=192)
    args = ap.parse_args()
    from mapperatorinator_amd import Tokenizer, _lib
    from mapperatorinator_amd.modeling import MapperatorinatorHIP
    from mapperatorinator_amd.server import build_sampling
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mh_testing import random_t5_state_dict, synthetic_audio
    lib = _lib.load()
    lib.mh_debug_phase_stamps.restype = C.c_int
    lib.mh_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int]
    _lib.set_option("decode_chains", args.chains)
    dev = torch.device("cuda", 0)
    tok = Tokenizer.benchmark_vocab(src_seq_len=1251)
    dims = T5_PRESETS["base"]
    B, new = args.batch, args.tokens
    model = MapperatorinatorHIP(random_t5_state_dict(dims, tok.vocab_size_in, tok.vocab_size_out, seed=0, lm_head_gain=6.0), dims,
                                vocab_size_in=tok.vocab_size_in, vocab_size_out=tok.vocab_size_out, src_seq_len=1251,
                                tgt_seq_len=512, dtype=torch.bfloat16, device=dev)
    eng = model.engine
    audio = synthetic_audio(B, 160000, seed=0).to(dev)
    prompt = torch.full((B, 1), tok.sos_id, dtype=torch.int32, device=dev)
    sp, _ = build_sampling(tok, dict(do_sample=False, num_beams=1, max_length=1 + new, temperature=1.0, context_type="map",
                                     pad_token_id=0), 512)
    eos_table = torch.zeros(tok.vocab_size_out, dtype=torch.uint8, device=dev)
    eng._enter()
    with torch.cuda.stream(eng.stream):
        kv = eng.cross_kv(eng.encode_mel(eng.mel(audio)))
        eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)           # warm-up
    eng._leave()
    torch.cuda.synchronize()
    ring, L = 64, dims.n_dec_layers
    tbuf = torch.empty((args.chains, ring, L, 2), dtype=torch.int64, device=dev)
    tbuf[..., 0] = -1      # UINT64_MAX
    tbuf[..., 1] = 0
    _lib.check(lib.mh_t5_decode_timing(tbuf.data_ptr(), ring), "timing on")
    _lib.check(lib.mh_debug_phase_stamps(None, 1), "stamps reset")
    eng._enter()
    with torch.cuda.stream(eng.stream):
        eng.decode(kv, prompt, None, eos_table, sp, poll_every=64)
    eng._leave()
    torch.cuda.synchronize()
    _lib.check(lib.mh_t5_decode_timing(None, 0), "timing off")
    raw = (C.c_uint64 * (16 * 16 * 2))()
    _lib.check(lib.mh_debug_phase_stamps(raw, 0), "stamps read")
    st = torch.tensor(list(raw), dtype=torch.float64).reshape(16, 16, 2)
    rate_khz = torch.cuda.get_device_properties(0).clock_rate if hasattr(torch.cuda.get_device_properties(0), "clock_rate") else 0
    t = tbuf.cpu()
    ok = (t[..., 1] > 0) & (t[..., 0] > 0)
    wall_ticks = (t[..., 1] - t[..., 0])[ok].double().mean().item()      # 100 MHz wall clock ticks
    print(f"workload: osuT5-base bf16, B={B}, chains={args.chains}, {new} tokens; cross-attention in situ (wall clock, "
          f"{int(ok.sum())} launches): {wall_ticks / 100.0:.2f} us")
    cross_last = st[11, 3, 0] / max(st[11, 3, 1], 1)
    tpu = cross_last / (wall_ticks / 100.0) if wall_ticks > 0 else float("nan")
    print(f"calibration: cross kernel's last stamp {cross_last:.0f} ticks ~ its wall time -> {tpu:.0f} ticks/us (upper bound: the "
          f"stamp is workgroup 0's, the wall time is the whole grid's)")
    for kid in range(16):
        if st[kid, :, 1].sum() == 0:
            continue
        if kid < 10:
            name, labels = f"gemv {EPI[kid // 2]}{' 8 waves' if kid % 2 else ''}", STAMPS["gemv"]
        elif kid == 10:
            name, labels = "self-attention + q/k/v projection", STAMPS["self"]
        elif kid == 11:
            name, labels = "cross-attention + q projection", STAMPS["cross"]
        else:
            name, labels = f"kernel {kid}", [f"stamp {i}" for i in range(16)]
        print(f"{name}:")
        prev = 0.0
        for i, lab in enumerate(labels):
            n = st[kid, i, 1].item()
            if n == 0:
                continue
            avg = st[kid, i, 0].item() / n
            print(f"    {lab:32s} {avg:9.0f} ticks  (+{avg - prev:8.0f})  = {avg / tpu:6.2f} us  [{int(n)} samples]")
            prev = avg


if __name__ == "__main__":
    main()
