"""Times the encoder stage of the headline shape alone (osuT5-base bf16, 32 chunks x 1251 frames): mel, encoder, cross K/V, HIP events.
    python tools/encoder_stage_bench.py [t5-base|t5-large]"""
import importlib.util, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from mh_testing import synthetic_audio_varied
    spec = importlib.util.spec_from_file_location("sbd", os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_batch_decode.py"))
    sbd = importlib.util.module_from_spec(spec); spec.loader.exec_module(sbd)
    dev = torch.device("cuda:0")
    name = sys.argv[1] if len(sys.argv) > 1 else "t5-base"
    model, tok, dims, frames = sbd.build(name, 385, dev)
    eng = model.engine
    audio = synthetic_audio_varied(32, (frames - 1) * 128, seed=5).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = [0.0, 0.0, 0.0]
    reps = 10
    with eng.on_stream():
        for r in range(reps + 2):
            ev[0].record(eng.stream)
            mel = eng.mel(audio)
            ev[1].record(eng.stream)
            enc = eng.encode_mel(mel)
            ev[2].record(eng.stream)
            kv = eng.cross_kv(enc)
            ev[3].record(eng.stream)
            torch.cuda.synchronize(dev)
            if r >= 2:
                for i in range(3):
                    acc[i] += ev[i].elapsed_time(ev[i + 1])
    d = eng.dims
    L = eng.packed.src_len
    fl = 32 * L * (d.n_enc_layers * (2.0 * d.d_model * (3 * d.n_heads * 64) + 2.0 * d.n_heads * 64 * d.d_model + 2.0 * d.d_model * 2 * d.d_ff + 2.0 * d.d_ff * d.d_model
                                    + 4.0 * L * 64 * d.n_heads) + 2.0 * 388 * d.d_model)
    enc_ms = acc[1] / reps
    print(json.dumps({"model": name, "mel_ms": round(acc[0] / reps, 3), "encoder_ms": round(enc_ms, 3), "cross_kv_ms": round(acc[2] / reps, 3),
                      "encoder_tflops": round(fl / enc_ms / 1e9, 1), "encoder_frac_of_bf16_peak": round(fl / enc_ms / 1e9 / 2500.0, 3)}))


if __name__ == "__main__":
    main()
