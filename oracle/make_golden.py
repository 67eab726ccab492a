"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz|json by running the IMPORTED REFERENCE
(this container only: needs /root/reference).  Run:  python -m oracle.make_golden

Every fixture is an output of reference code (reference `model_generate`, `Mapperatorinator`,
HF T5 as wired by the reference, osu_diffusion `DiT` / `create_diffusion`, reference `Tokenizer`)
on weights regenerated from a numpy seed (mapperatorinator_amd/testing.py), so the GPU box can
rebuild the same weights and compare the HIP path against what the reference produced here.
The mel frontend inside these runs is oracle/mel.py (nnAudio is not installable: parity unpinned
for that one stage, see oracle/mel.py).
"""
from __future__ import annotations

import copy
import json
import os

import numpy as np
import torch

from mapperatorinator_amd.t5_engine import T5_PRESETS
from mh_testing import (DIT_PRESETS, DIVERSE_GAINS, boost_timed_rows, random_dit_state_dict,
                                          random_t5_state_dict, synthetic_audio, synthetic_audio_varied,
                                          synthetic_dit_inputs)

from . import dit as odit
from . import mel as omel
from . import ref_harness as rh

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

T5_CASES = {
    # name: size, samples, src frames, tgt_len, weight seed, lm_head gain, audio seed, prompts (left-padded with 0),
    #       audio kind, weight gains, record the per-step processed scores?
    "t5_tiny": dict(size="tiny", ns=32000, src=251, tgt=48, wseed=11, gain=6.0, aseed=4,
                    prompts=[[0, 0, 1], [1, 40, 700], [0, 1, 9]], audio="stationary", gains=None, scores=False),
    "t5_small": dict(size="small", ns=160000, src=1251, tgt=96, wseed=3, gain=4.0, aseed=1, prompts=[[0, 1], [1, 5]],
                     audio="varied", gains="diverse", scores=False),
    # BASELINE configs[1] dims (osuT5-base, 1251 frames): ragged prompts, 133 new tokens per row, per-step scores
    "t5_base": dict(size="base", ns=160000, src=1251, tgt=136, wseed=5, gain=4.0, aseed=2,
                    prompts=[[0, 0, 1], [1, 40, 700], [0, 1, 9], [1, 5, 1003]], audio="varied", gains="diverse",
                    scores=True),
    # BASELINE configs[4] dims ("osuT5-large" = google/t5-v1_1-large through the same wrapper): 2 ragged rows, 69 new tokens
    # one full 16-row decode chain of the headline shape (bench.py runs two of them): 16 ragged rows, 197 new tokens each, fp32 and bf16
    "t5_base_wide": dict(size="base", ns=160000, src=1251, tgt=200, wseed=13, gain=4.0, aseed=6,
                         prompts=[[0, 0, 1], [1, 40, 700], [0, 1, 9], [1, 5, 1003], [0, 0, 1], [0, 1, 77], [1, 300, 301], [0, 1, 1500],
                                  [1, 9, 10], [0, 0, 1], [0, 1, 64], [1, 800, 1800], [0, 1, 3], [1, 21, 22], [0, 0, 1], [0, 1, 1200]],
                         audio="varied", gains="diverse", scores=True),
    "t5_large": dict(size="large", ns=160000, src=1251, tgt=72, wseed=7, gain=4.0, aseed=3,
                     prompts=[[0, 1, 40], [1, 9, 700]], audio="varied", gains="diverse", scores=True),
}
TOPK = 16   # per (step, row): the TOPK largest processed scores + their ids + the row's logsumexp


def case_audio(c, batch):
    return (synthetic_audio_varied if c["audio"] == "varied" else synthetic_audio)(batch, c["ns"], seed=c["aseed"])


def case_weights(c, tok):
    return random_t5_state_dict(T5_PRESETS[c["size"]], tok.vocab_size_in, tok.vocab_size_out, seed=c["wseed"],
                                lm_head_gain=c["gain"], gains=DIVERSE_GAINS if c["gains"] == "diverse" else None)


def topk_scores(rec):
    """list of (B, V) processed scores -> (steps, B, TOPK) values, ids, (steps, B) logsumexp"""
    s = torch.stack(rec).float()
    v, i = s.topk(TOPK, dim=-1)
    return v.numpy(), i.numpy().astype(np.int32), torch.logsumexp(s, -1).numpy()


def t5_case(name):
    c = T5_CASES[name]
    size, ns, src, tgt, prompts = c["size"], c["ns"], c["src"], c["tgt"], c["prompts"]
    model, tok, _ = rh.build_reference_t5(size, src_seq_len=src, tgt_seq_len=tgt)
    sd = case_weights(c, tok)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res
    audio = case_audio(c, len(prompts))
    prompt = torch.tensor(prompts)
    mask = prompt.ne(0)
    with torch.no_grad():
        mel = model.spectrogram(audio)
    enc = rh.reference_encode(model, audio)
    rec = [] if c["scores"] else None
    ids, stats = rh.reference_generate(model, tok, audio, prompt, rh.default_generate_kwargs(tgt), mask, record_scores=rec)
    # a second run with processors switched on: temperature + timeshift bias + lookahead EOS window
    ids2, _ = rh.reference_generate(model, tok, audio, prompt,
                                    rh.default_generate_kwargs(tgt, temperature=0.7, timeshift_bias=0.35,
                                                               lookahead_time=3000), mask)
    extra = {}
    if rec is not None:
        v, i, lse = topk_scores(rec)
        gap = v[..., 0] - v[..., 1]
        extra = dict(top_vals=v, top_ids=i, lse=lse)
        print(name, "steps", len(rec), "top-2 gap min / median", float(gap.min()), float(np.median(gap)))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        vocab_in=tok.vocab_size_in, vocab_out=tok.vocab_size_out, n_samples=ns, src_len=src, tgt_len=tgt,
        weight_seed=c["wseed"], lm_head_gain=c["gain"], audio_seed=c["aseed"], audio_kind=c["audio"],
        gains=c["gains"] or "", prompt=prompt.numpy(),
        mel_slice=mel[:, ::37, ::29].numpy(), mel_sum=mel.double().sum().item(),
        enc_slice=enc[:, ::53, ::17].numpy(), enc_abs_mean=enc.abs().double().mean().item(),
        ids=ids.numpy(), ids_processors=ids2.numpy(), **extra,
    )
    print(name, "ids", ids.shape, "distinct", len(set(ids.flatten().tolist())), "distinct(processors run)",
          len(set(ids2.flatten().tolist())), "tok/s(ref,cpu)", stats["tokens_per_second"])


COND_CASE = dict(size="tiny", ns=32000, src=251, tgt=40, wseed=3, gain=4.0, aseed=7, cond_dim=16, num_mappers=11, cseed=1,
                 prompts=[[0, 1], [1, 40], [0, 1]], difficulty=[2.5, 6.1, 9.0], mapper_idx=[3, -1, 10],
                 song_position=[[0.0, 0.1], [0.45, 0.5], [0.9, 1.0]])


def t5_conditioning_case(name="t5_tiny_cond"):
    """The wrapper's conditioning embedders (difficulty, mapper style, song position; modeling_mapperatorinator.py:104-128,
    395-414) on the reference: the per-row conditioning vectors its own modules produce, the encoder states with them
    concatenated to the mel frames, and the greedy ids -- plus the ids WITHOUT conditioning, to show it matters."""
    from mh_testing import add_random_conditioning
    c = COND_CASE
    model, tok, _ = rh.build_reference_t5(c["size"], src_seq_len=c["src"], tgt_seq_len=c["tgt"],
                                          cond=dict(cond_dim=c["cond_dim"], num_mappers=c["num_mappers"]))
    sd = random_t5_state_dict(T5_PRESETS[c["size"]], tok.vocab_size_in, tok.vocab_size_out, seed=c["wseed"], lm_head_gain=c["gain"])
    add_random_conditioning(sd, T5_PRESETS[c["size"]].d_model, 388, c["cond_dim"], c["num_mappers"], seed=c["cseed"])
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if "embedder" in k], res
    audio = synthetic_audio(len(c["prompts"]), c["ns"], seed=c["aseed"])
    prompt = torch.tensor(c["prompts"])
    diff, mp, sp = torch.tensor(c["difficulty"]), torch.tensor(c["mapper_idx"]), torch.tensor(c["song_position"])
    cond = rh.reference_cond_vectors(model, diff, mp, sp)
    enc = rh.reference_encode(model, audio, cond)
    ids, _ = rh.reference_generate(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"]), prompt.ne(0), cond=cond)
    ids0, _ = rh.reference_generate(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"]), prompt.ne(0),
                                    cond=torch.zeros_like(cond))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        vocab_in=tok.vocab_size_in, vocab_out=tok.vocab_size_out, n_samples=c["ns"], src_len=c["src"], tgt_len=c["tgt"],
        weight_seed=c["wseed"], lm_head_gain=c["gain"], audio_seed=c["aseed"], cond_dim=c["cond_dim"],
        num_mappers=c["num_mappers"], cond_seed=c["cseed"], prompt=prompt.numpy(), difficulty=diff.numpy(),
        mapper_idx=mp.numpy(), song_position=sp.numpy(), cond_vectors=cond.numpy(),
        enc_slice=enc[:, ::13, ::7].numpy(), enc_abs_mean=enc.abs().double().mean().item(), ids=ids.numpy(), ids_zero_cond=ids0.numpy())
    print(name, "ids", ids.shape, "positions that differ from the zero-conditioning run",
          int((ids != ids0[:, :ids.shape[1]]).sum()) if ids0.shape == ids.shape else "shape differs", "cond |mean|", float(cond.abs().mean()))


def t5_bf16_reference_case(name="t5_base"):
    """The reference itself in bfloat16 (`model.to(torch.bfloat16)`, the precision switch of
    osuT5/osuT5/utils/model_utils.py:375-376) on the weights / audio / prompts of `name`: free-running greedy ids,
    the top scores of every step, and how often the fp32 reference -- teacher-forced on those ids -- picks the same
    token.  The GPU test runs the HIP bf16 path teacher-forced on the same ids and asserts its agreement."""
    c = T5_CASES[name]
    tgt, prompts = c["tgt"], c["prompts"]
    model, tok, _ = rh.build_reference_t5(c["size"], src_seq_len=c["src"], tgt_seq_len=tgt)
    model.load_state_dict(case_weights(c, tok), strict=False)
    audio = case_audio(c, len(prompts))
    prompt = torch.tensor(prompts)
    m16 = copy.deepcopy(model).to(torch.bfloat16)
    rec16 = []
    ids16, _ = rh.reference_generate(m16, tok, audio, prompt, rh.default_generate_kwargs(tgt, precision="bf16"), prompt.ne(0),
                                     record_scores=rec16)
    v, i, lse = topk_scores(rec16)
    # the CPU oracle teacher-forced on the bf16 run's ids: on which steps does it decide differently?  (steps after a
    # row's EOS are pads and do not count)
    from oracle import t5 as ot5
    d = T5_PRESETS[c["size"]]
    P = prompt.shape[1]
    gap = torch.from_numpy(v[..., 0] - v[..., 1]).T                     # (B, steps)
    ts0, ts1 = rh.ts_range(tok)
    agree = {}
    for tag, rounding in (("fp32", None), ("bf16_contract", "bf16")):
        o = ot5.T5Oracle(case_weights(c, tok), d.d_model, d.d_ff, d.n_heads, d.n_enc_layers, d.n_dec_layers, rounding=rounding)
        _, sc = o.generate(o.encode_audio(audio), prompt, prompt.ne(0), [], ids16.shape[1], ts0, ts1, [tok.sos_id],
                           forced=ids16, return_logits=True)
        pick = torch.stack([s.argmax(-1) for s in sc], 1)              # (B, steps)
        want = ids16[:, P:P + pick.shape[1]]
        live, ok = want.ne(0), pick == want
        dec = live & (gap[:, :pick.shape[1]] >= BF16_DECISIVE_GAP)
        agree[tag] = (ok[live].float().mean().item(), ok[dec].float().mean().item(), int(dec.sum()))
    print(name, "bf16 reference: ids", tuple(ids16.shape), "distinct", len(set(ids16.flatten().tolist())),
          "pads per row", (ids16 == 0).sum(1).tolist(), "top-2 gap median", float(gap.median()),
          "| teacher-forced top-1 agreement with the bf16 reference (all live steps, steps with gap >= "
          f"{BF16_DECISIVE_GAP}, their count): fp32 oracle {agree['fp32']}, bf16-contract oracle {agree['bf16_contract']}")
    np.savez_compressed(os.path.join(OUT, name + "_bf16ref.npz"), ids=ids16.numpy(), top_vals=v, top_ids=i, lse=lse,
                        decisive_gap=BF16_DECISIVE_GAP, agree_fp32_oracle=np.array(agree["fp32"]),
                        agree_bf16_contract_oracle=np.array(agree["bf16_contract"]))


BF16_DECISIVE_GAP = 0.5   # bf16 logits of magnitude 8..16 are spaced 0.0625..0.125: below this a "decision" is rounding


TF_CASE = dict(src=251, tgt=48, n_samples=32000, weight_seed=21, lm_head_gain=6.0, timed_gain=2.0, audio_seed=5,
               prompt=[[0, 0, 3], [3, 40, 2068], [0, 3, 9]], negative=[[0, 0, 1], [0, 1, 2069], [0, 0, 3]])
TF_RUNS = {
    # ConditionalTemperature + LookbackBias(types_first=True)
    "tf": dict(types_first=True, temperature=0.9, timing_temperature=0.5, mania_column_temperature=0.6,
               taiko_hit_temperature=0.7, lookback_time=500),
    # classifier-free guidance alone
    "cfg": dict(cfg_scale=2.5),
    # everything at once
    "all": dict(cfg_scale=1.7, types_first=True, temperature=1.1, timing_temperature=0.6, mania_column_temperature=0.8,
                taiko_hit_temperature=0.5, lookback_time=800, lookahead_time=400, timeshift_bias=0.2),
}


def types_first_case():
    """Reference `model_generate` with the types_first processors and classifier-free guidance: ids + the processed
    scores of every step (observed at HF's LogitsProcessorList), on a tokenizer that has every token family."""
    c = TF_CASE
    model, tok, _ = rh.build_reference_t5("tiny", src_seq_len=c["src"], tgt_seq_len=c["tgt"], types_first=True)
    with open(os.path.join(OUT, "tokenizer_types_first.json"), "w") as f:
        json.dump(tok.state_dict(), f)
    sd = random_t5_state_dict(T5_PRESETS["tiny"], tok.vocab_size_in, tok.vocab_size_out, seed=c["weight_seed"],
                              lm_head_gain=c["lm_head_gain"])
    boost_timed_rows(sd, tok, c["timed_gain"])
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res
    audio = synthetic_audio(len(c["prompt"]), c["n_samples"], seed=c["audio_seed"])
    prompt, neg = torch.tensor(c["prompt"]), torch.tensor(c["negative"])
    out = dict(vocab_in=tok.vocab_size_in, vocab_out=tok.vocab_size_out, prompt=prompt.numpy(), negative=neg.numpy(),
               runs=json.dumps(TF_RUNS),
               **{k: v for k, v in c.items() if k not in ("prompt", "negative")})
    for name, over in TF_RUNS.items():
        rec = []
        ids, _ = rh.reference_generate(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"], **over),
                                       prompt.ne(0), negative_prompt=neg if "cfg_scale" in over else None,
                                       record_scores=rec)
        out["ids_" + name] = ids.numpy()
        out["scores_" + name] = torch.stack(rec).numpy()
        timed_hits = sum(int(((ids[:, 3:-1] >= 2058) & (ids[:, 3:-1] < 2080)).sum()) for _ in [0])
        print("types_first run", name, ids.shape, "steps", len(rec), "timed ids emitted", timed_hits,
              "row0 scroll-speed ids", int(((ids[0] >= 882) & (ids[0] < 1883)).sum()))
    np.savez_compressed(os.path.join(OUT, "t5_tiny_tf.npz"), **out)


VW_CASES = {
    # name: backbone dims (mapperatorinator_amd.whisper_engine.VARWHISPER_PRESETS), log-mel frames per chunk, tgt_len, ...
    "vw_test": dict(size="test", frames=250, tgt=48, wseed=3, gain=5.0, aseed=4, prompts=[[0, 0, 1], [1, 40, 700], [0, 1, 9]],
                    bias=True),
    "vw_test_nobias": dict(size="test", frames=250, tgt=40, wseed=5, gain=5.0, aseed=6, prompts=[[0, 1], [1, 9]], bias=False),
    # the released V32 backbone: 'OliBomby/varwhisper-small' (whisper-small dims) at its own chunk size, data.src_seq_len 2048
    "vw_small": dict(size="small", frames=2048, tgt=72, wseed=9, gain=4.0, aseed=7, prompts=[[0, 1, 40], [1, 9, 700]], bias=True),
}


def vw_case(name):
    """The Whisper-family backbone on the REFERENCE: `Mapperatorinator` over VarWhisperForConditionalGeneration
    (custom_transformers/modeling_varwhisper.py) as configs/model/varwhisper_*_v3.yaml wire it, through the reference's own
    `model_generate`: log-mel slice, encoder states, greedy ids, the 16 best processed scores of every step."""
    from mh_testing import random_varwhisper_state_dict
    from mapperatorinator_amd.whisper_engine import VARWHISPER_PRESETS
    c = VW_CASES[name]
    d = VARWHISPER_PRESETS[c["size"]]
    over = dict(d_model=d.d_model, encoder_layers=d.n_enc_layers, decoder_layers=d.n_dec_layers, encoder_attention_heads=d.n_heads,
                decoder_attention_heads=d.n_heads, encoder_ffn_dim=d.d_ff, decoder_ffn_dim=d.d_ff)
    model, tok, _ = rh.build_reference_varwhisper("small", src_seq_len=c["frames"], tgt_seq_len=c["tgt"], overwrite=over,
                                                  attention_bias=c["bias"])
    sd = random_varwhisper_state_dict(d.d_model, d.n_heads, d.n_enc_layers, d.n_dec_layers, d.d_ff, tok.vocab_size_in,
                                      tok.vocab_size_out, seed=c["wseed"], head_gain=c["gain"], attention_bias=c["bias"],
                                      gains={"decoder_embedder": 0.5})
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if "loss_fn" not in k], res
    ns = (c["frames"] - 1) * 128
    audio = synthetic_audio_varied(len(c["prompts"]), ns, seed=c["aseed"])
    prompt = torch.tensor(c["prompts"])
    with torch.no_grad():
        mel = model.spectrogram(audio)
    enc = rh.reference_encode_whisper(model, audio)
    rec = []
    ids, stats = rh.reference_generate_whisper(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"]), record_scores=rec)
    ids2, _ = rh.reference_generate_whisper(model, tok, audio, prompt,
                                            rh.default_generate_kwargs(c["tgt"], temperature=0.7, timeshift_bias=0.35, lookahead_time=3000))
    v, i, lse = topk_scores(rec)
    gap = v[..., 0] - v[..., 1]
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), size=c["size"], vocab_in=tok.vocab_size_in, vocab_out=tok.vocab_size_out, n_samples=ns,
        in_frames=c["frames"], tgt_len=c["tgt"], weight_seed=c["wseed"], head_gain=c["gain"], audio_seed=c["aseed"],
        attention_bias=c["bias"], prompt=prompt.numpy(), mel_slice=mel[:, ::37, ::11].numpy(), mel_sum=mel.double().sum().item(),
        enc_slice=enc[:, ::29, ::17].numpy(), enc_abs_mean=enc.abs().double().mean().item(), ids=ids.numpy(),
        ids_processors=ids2.numpy(), top_vals=v, top_ids=i, lse=lse)
    print(name, "ids", tuple(ids.shape), "distinct", len(set(ids.flatten().tolist())), "top-2 gap min / median", float(gap.min()),
          float(np.median(gap)), "tok/s(ref,cpu)", stats["tokens_per_second"])


WF_CASES = {
    # the other two Whisper-family backbones (oracle/whisper_family.py).  kind "rope": 'Tiger14n/ropewhisper-*' (V30 / V31), "hf":
    # 'openai/whisper-*' (V28 / V29); dims = mapperatorinator_amd.whisper_engine.VARWHISPER_PRESETS (openai/whisper-* dims)
    "rw_test": dict(kind="rope", size="test", frames=250, tgt=48, n_mels=80, wseed=3, gain=5.0, aseed=4,
                    prompts=[[0, 0, 1], [1, 40, 700], [0, 1, 9]], cond=None),
    # conditioning embedders as conv1 channels (configs/model/whisper_small_v2.yaml:9-13)
    "rw_test_cond": dict(kind="rope", size="test", frames=250, tgt=40, n_mels=80, wseed=5, gain=5.0, aseed=6, prompts=[[0, 1], [1, 9], [0, 1]],
                         cond=dict(cond_dim=16, num_mappers=11, cseed=2, difficulty=[2.5, 6.1, 9.0], mapper_idx=[3, -1, 10],
                                   song_position=[[0.0, 0.1], [0.45, 0.5], [0.9, 1.0]])),
    # the released V30 backbone: 'Tiger14n/ropewhisper-small' (whisper-small dims) at its own chunk size (data.src_seq_len 4096 ->
    # 2048 encoder positions), 80 mels + 3 x 128 conditioning channels
    "rw_small": dict(kind="rope", size="small", frames=4096, tgt=40, n_mels=80, wseed=9, gain=4.0, aseed=7, prompts=[[0, 1, 40], [1, 9, 700]],
                     cond=dict(cond_dim=128, num_mappers=11, cseed=3, difficulty=[4.2, 7.7], mapper_idx=[5, -1], song_position=[[0.2, 0.3], [0.8, 0.9]])),
    "hfw_test": dict(kind="hf", size="test", frames=250, tgt=48, n_mels=388, wseed=17, gain=5.0, aseed=8,
                     prompts=[[0, 0, 1], [1, 40, 700], [0, 1, 9]], cond=None),
    # the released V29 backbone: 'openai/whisper-small' at its own chunk size (data.src_seq_len 1024 -> 512 encoder positions)
    "hfw_small": dict(kind="hf", size="small", frames=1024, tgt=72, n_mels=388, wseed=15, gain=4.0, aseed=9, prompts=[[0, 1, 40], [1, 9, 700]],
                      cond=None),
}


def wf_weights(c, tok):
    """state dict of a WF_CASES entry from its seeds (shared with tests/conftest.py:wf_golden_case)"""
    from mh_testing import add_random_cond_embedders, random_whisper_family_state_dict
    from mapperatorinator_amd.whisper_engine import VARWHISPER_PRESETS
    d = VARWHISPER_PRESETS[c["size"]]
    cd = c["cond"]
    sd = random_whisper_family_state_dict(c["kind"], d.d_model, d.n_heads, d.n_enc_layers, d.n_dec_layers, d.d_ff, tok.vocab_size_in,
                                          tok.vocab_size_out, c["n_mels"], src_positions=c["frames"] // 2, tgt_positions=c["tgt"],
                                          cond_size=3 * cd["cond_dim"] if cd else 0, seed=c["wseed"], head_gain=c["gain"],
                                          gains={"decoder_embedder": 0.5})
    if cd:
        add_random_cond_embedders(sd, cd["cond_dim"], cd["num_mappers"], seed=cd["cseed"])
    return sd


def wf_case(name):
    """'Tiger14n/ropewhisper-*' / 'openai/whisper-*' behind the REFERENCE's wrapper, through its own `model_generate`: front-end
    slice, encoder states, greedy ids (ragged left-padded prompts), the 16 best processed scores of every step."""
    from mapperatorinator_amd.whisper_engine import VARWHISPER_PRESETS
    c = WF_CASES[name]
    d = VARWHISPER_PRESETS[c["size"]]
    over = dict(d_model=d.d_model, encoder_layers=d.n_enc_layers, decoder_layers=d.n_dec_layers, encoder_attention_heads=d.n_heads,
                decoder_attention_heads=d.n_heads, encoder_ffn_dim=d.d_ff, decoder_ffn_dim=d.d_ff)
    cd = c["cond"]
    model, tok, _ = rh.build_reference_whisper_family(c["kind"], src_seq_len=c["frames"], tgt_seq_len=c["tgt"], n_mels=c["n_mels"], overwrite=over,
                                                      cond=dict(cond_dim=cd["cond_dim"], num_mappers=cd["num_mappers"]) if cd else None)
    sd = wf_weights(c, tok)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if "loss_fn" not in k], res
    ns = (c["frames"] - 1) * 128
    audio = synthetic_audio_varied(len(c["prompts"]), ns, seed=c["aseed"])
    prompt = torch.tensor(c["prompts"])
    cond = None
    extra = {}
    if cd:
        diff, mp, sp = torch.tensor(cd["difficulty"]), torch.tensor(cd["mapper_idx"]), torch.tensor(cd["song_position"])
        cond = rh.reference_cond_vectors(model, diff, mp, sp)
        extra = dict(difficulty=diff.numpy(), mapper_idx=mp.numpy(), song_position=sp.numpy(), cond_vectors=cond.numpy(),
                     cond_dim=cd["cond_dim"], num_mappers=cd["num_mappers"], cond_seed=cd["cseed"])
    with torch.no_grad():
        mel = model.spectrogram(audio)
    enc = rh.reference_encode_whisper_family(model, audio, cond)
    rec = []
    ids, stats = rh.reference_generate_whisper_family(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"]), record_scores=rec, cond=cond)
    ids2, _ = rh.reference_generate_whisper_family(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"], temperature=0.7, timeshift_bias=0.35,
                                                                                                        lookahead_time=3000), cond=cond)
    if c["kind"] == "hf":   # transformers 4.57's mask-derived decoder positions (the reference's pin), through the same reference objects
        idsm, _ = rh.reference_generate_whisper_family(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"]), positions_from_mask=True)
        extra["ids_mask_positions"] = idsm.numpy()
    if cd:   # the run WITHOUT conditioning, to show it matters
        ids0, _ = rh.reference_generate_whisper_family(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"]), cond=torch.zeros_like(cond))
        extra["ids_zero_cond"] = ids0.numpy()
    v, i, lse = topk_scores(rec)
    gap = v[..., 0] - v[..., 1]
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), kind=c["kind"], size=c["size"], vocab_in=tok.vocab_size_in, vocab_out=tok.vocab_size_out, n_samples=ns,
        in_frames=c["frames"], tgt_len=c["tgt"], n_mels=c["n_mels"], weight_seed=c["wseed"], head_gain=c["gain"], audio_seed=c["aseed"],
        prompt=prompt.numpy(), mel_slice=mel[:, ::37, ::11].numpy(), mel_sum=mel.double().sum().item(),
        enc_slice=enc[:, ::29, ::17].numpy(), enc_abs_mean=enc.abs().double().mean().item(), ids=ids.numpy(),
        ids_processors=ids2.numpy(), top_vals=v, top_ids=i, lse=lse, **extra)
    print(name, "ids", tuple(ids.shape), "distinct", len(set(ids.flatten().tolist())), "top-2 gap min / median", float(gap.min()),
          float(np.median(gap)), "tok/s(ref,cpu)", stats["tokens_per_second"],
          ("| positions differing from zero conditioning: %s" % (int((ids.numpy() != extra["ids_zero_cond"]).sum())
                                                                 if ids.numpy().shape == extra["ids_zero_cond"].shape else "shape differs")) if cd else "")


BEAM_CASE = dict(src=251, tgt=40, ns=32000, wseed=13, gain=1.5, aseed=6, prompts=[[0, 0, 1], [1, 40, 700], [0, 1, 9]],
                 negative=[[0, 0, 1], [0, 1, 701], [0, 0, 1]],
                 runs={"b2": dict(num_beams=2), "b3": dict(num_beams=3),
                       "b2p": dict(num_beams=2, lookahead_time=500, temperature=0.8, timeshift_bias=0.3),
                       "b3p": dict(num_beams=3, lookahead_time=700, lookback_time=300, temperature=1.3),
                       # classifier-free guidance UNDER beams (the timing pass: processor.py:709 halves its batch for it): the doubled
                       # rows of prepare_inputs_for_generation + `beam_idx.repeat(2)` of MapperatorinatorCache.reorder_cache
                       "b2g": dict(num_beams=2, cfg_scale=2.0),
                       "b3g": dict(num_beams=3, cfg_scale=1.5, temperature=0.8, timeshift_bias=0.3, lookahead_time=500),
                       # beam-sample (do_sample under beams, processor.py:147-160): HF's top-k / top-p warpers behind the list with
                       # min_tokens_to_keep = #eos + 1, K continuations drawn without replacement by testing.SeededMultinomial
                       "b2s": dict(num_beams=2, do_sample=True, temperature=0.9, top_p=0.9),
                       "b3s": dict(num_beams=3, do_sample=True, top_k=64, lookahead_time=500, timeshift_bias=0.3),
                       "b2sg": dict(num_beams=2, do_sample=True, cfg_scale=1.5, top_p=0.95, top_k=40, temperature=1.1)},
                 sample_seeds={"b2s": 1001, "b3s": 1002, "b2sg": 1003})


def beam_case(name="t5_tiny_beam"):
    """`num_beams > 1` on the reference: its `model_generate` -> HF beam search with `MapperatorinatorCache.reorder_cache`
    (inference/cache_utils.py:16-20), 2 and 3 beams, with and without processors / EOS windows (hypotheses of different
    lengths), next to the greedy ids of the same inputs."""
    c = BEAM_CASE
    model, tok, _ = rh.build_reference_t5("tiny", src_seq_len=c["src"], tgt_seq_len=c["tgt"])
    sd = random_t5_state_dict(T5_PRESETS["tiny"], tok.vocab_size_in, tok.vocab_size_out, seed=c["wseed"], lm_head_gain=c["gain"],
                              gains=DIVERSE_GAINS)
    model.load_state_dict(sd, strict=False)
    audio = synthetic_audio_varied(len(c["prompts"]), c["ns"], seed=c["aseed"])
    prompt, neg = torch.tensor(c["prompts"]), torch.tensor(c["negative"])
    out = dict(vocab_in=tok.vocab_size_in, vocab_out=tok.vocab_size_out, prompt=prompt.numpy(), negative=neg.numpy(), runs=json.dumps(c["runs"]),
               sample_seeds=json.dumps(c["sample_seeds"]),
               **{k: v for k, v in c.items() if k not in ("prompts", "runs", "negative", "sample_seeds")})
    from mh_testing import SeededMultinomial
    for tag, kw in c["runs"].items():
        ng = neg if kw.get("cfg_scale", 1.0) > 1.0 else None
        real_multinomial = torch.multinomial
        if kw.get("do_sample"):
            torch.multinomial = SeededMultinomial(c["sample_seeds"][tag])
        try:
            ids, _ = rh.reference_generate(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"], **kw), prompt.ne(0), negative_prompt=ng)
            if kw.get("do_sample"):
                print("   ", tag, "multinomial calls", torch.multinomial.calls)
        finally:
            torch.multinomial = real_multinomial
        greedy, _ = rh.reference_generate(model, tok, audio, prompt, rh.default_generate_kwargs(c["tgt"], **dict(kw, num_beams=1, do_sample=False)),
                                          prompt.ne(0), negative_prompt=ng)
        out["ids_" + tag], out["greedy_" + tag] = ids.numpy(), greedy.numpy()
        w = min(ids.shape[1], greedy.shape[1])
        print(name, tag, tuple(ids.shape), "greedy", tuple(greedy.shape), "positions where beams and greedy differ",
              int((ids[:, :w] != greedy[:, :w]).sum()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def tokenizer_case():
    _, tok, _ = rh.build_reference_t5("small", src_seq_len=1251, tgt_seq_len=64)
    with open(os.path.join(OUT, "tokenizer_benchmark_vocab.json"), "w") as f:
        json.dump(tok.state_dict(), f)
    # HF bucket function as the reference's backbone uses it
    from transformers.models.t5.modeling_t5 import T5Attention
    rel = torch.arange(-1300, 1301)
    np.savez_compressed(os.path.join(OUT, "t5_buckets.npz"), rel=rel.numpy(),
                        bidirectional=T5Attention._relative_position_bucket(rel, True, 32, 128).numpy(),
                        unidirectional=T5Attention._relative_position_bucket(rel, False, 32, 128).numpy())


def dit_case(name, preset, T, wseed, iseed, cfg_scale, loop_steps=100):
    """loop_steps < 100: the reference loop is run over that many injected draws only (the long DiT-B window: the
    per-step arithmetic is what is pinned, a full 100-step trajectory of a random denoiser is chaotic anyway)"""
    depth, hidden, heads = DIT_PRESETS[preset]
    sd = random_dit_state_dict(depth, hidden, seed=wseed)
    rh.ref_shims.install()
    from osu_diffusion.utils.models import DiT
    ref = DiT(context_size=272, hidden_size=hidden, depth=depth, num_heads=heads, class_size=300).eval()
    ref.load_state_dict(sd, strict=True)
    z, c, y = synthetic_dit_inputs(T, seed=iseed)
    mask = odit.band_mask(T, 128)
    diff = rh.reference_diffusion()
    eps = {}
    for tv in (99, 50, 0):
        with torch.no_grad():
            eps[tv] = ref.forward_with_cfg(z, torch.full((2,), tv, dtype=torch.long), c, y, cfg_scale, attn_mask=mask)
    noise = torch.from_numpy(np.random.default_rng(500 + iseed).standard_normal((100, *z.shape)).astype(np.float32))
    # one reference p_sample from z at loop index 57, and the full 100-step loop
    from osu_diffusion.utils.diffusion import gaussian_diffusion as gd
    orig = gd.th.randn_like
    gd.th.randn_like = lambda v: noise[0]
    try:
        with torch.no_grad():
            one = diff.p_sample(ref.forward_with_cfg, z, torch.full((2,), 57, dtype=torch.long), clip_denoised=True,
                                model_kwargs=dict(c=c, y=y, cfg_scale=cfg_scale, attn_mask=mask, key_padding_mask=None))
    finally:
        gd.th.randn_like = orig
    if loop_steps >= 100:
        full = rh.reference_ddpm(ref, diff, z, c, y, cfg_scale, mask, list(noise))
    else:   # the last `loop_steps` iterations (indices loop_steps-1 .. 0) of the reference loop from z
        full = z
        gd.th.randn_like = lambda v: pending.pop(0)
        pending = list(noise[:loop_steps])
        try:
            with torch.no_grad():
                for i in range(loop_steps - 1, -1, -1):
                    full = diff.p_sample(ref.forward_with_cfg, full, torch.full((2,), i, dtype=torch.long), clip_denoised=True,
                                         model_kwargs=dict(c=c, y=y, cfg_scale=cfg_scale, attn_mask=mask, key_padding_mask=None))["sample"]
        finally:
            gd.th.randn_like = orig
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), preset=preset, T=T, weight_seed=wseed, input_seed=iseed, cfg_scale=cfg_scale,
        loop_steps=loop_steps, eps_t99=eps[99].numpy(), eps_t50=eps[50].numpy(), eps_t0=eps[0].numpy(),
        p_sample_i57=one["sample"].numpy(), p_sample_i57_x0=one["pred_xstart"].numpy(), sample_100=full.numpy(),
        timestep_map=np.array(diff.timestep_map), betas=diff.betas,
        posterior_log_variance_clipped=diff.posterior_log_variance_clipped,
        posterior_mean_coef1=diff.posterior_mean_coef1, posterior_mean_coef2=diff.posterior_mean_coef2,
        sqrt_recip_alphas_cumprod=diff.sqrt_recip_alphas_cumprod,
        sqrt_recipm1_alphas_cumprod=diff.sqrt_recipm1_alphas_cumprod,
    )
    print(name, "eps scale", eps[50].abs().max().item(), "sample range", full.min().item(), full.max().item())


PIPE_CASE = dict(preset="DiT-XS", T=300, weight_seed=31, point_seed=7, noise_seed=11, classes=[3, 40, 250],
                 null_classes=[3, 299],
                 knobs=dict(timesteps=[12, 0, 0, 0, 0, 0, 0, 0, 0, 0], seq_len=32, max_seq_len=160, overlap_buffer=16,
                            cfg_scale=1.5, refine_iters=2))


def pipeline_case():
    """Reference `DiffisionPipeline.generate` (window loop, in-paint masks with start/end time, refine steps) on
    synthetic hit objects, gaussian draws injected from a numpy stream (one draw per p_sample call, in call order)."""
    from mapperatorinator_amd.diffusion_pipeline import points_to_sequence
    from mh_testing import pipeline_windows, synthetic_hit_objects, synthetic_sliders
    c = PIPE_CASE
    depth, hidden, heads = DIT_PRESETS[c["preset"]]
    sd = random_dit_state_dict(depth, hidden, seed=c["weight_seed"])
    rh.ref_shims.install()
    from osu_diffusion.utils.models import DiT
    ref = DiT(context_size=272, hidden_size=hidden, depth=depth, num_heads=heads, class_size=300).eval()
    ref.load_state_dict(sd, strict=True)
    x, y, times, dist, typ = synthetic_hit_objects(c["T"], c["point_seed"])
    seq_x, seq_o, seq_c = points_to_sequence(x, y, times, dist, typ)
    # the conditioning assembly itself against the reference's function
    from osu_diffusion import timestep_embedding as ref_te
    assert torch.equal(seq_c[:128], ref_te(seq_o * 0.1, 128).T) and torch.equal(seq_c[128:256], ref_te(torch.from_numpy(dist), 128).T)
    cv, ucv = torch.zeros(300), torch.zeros(300)
    cv[c["classes"]] = 1
    ucv[c["null_classes"]] = 1
    k = c["knobs"]
    start_time, end_time = float(times[20]), float(times[280])
    out = {}
    # "short": 2 DDPM steps + 1 refine step per window -- errors cannot compound, pins the window / mask logic tightly
    # "sliders": the slider end re-projection of `denoised_fn` on synthetic sliders (mh_testing)
    sliders = synthetic_sliders(c["T"], c["point_seed"] + 1)
    for tag, kk, seed in (("", k, c["noise_seed"]),
                          ("_short", dict(k, timesteps=[2] + [0] * 9, refine_iters=1), c["noise_seed"] + 1),
                          ("_sliders", k, c["noise_seed"] + 2),
                          ("_sliders_short", dict(k, timesteps=[2] + [0] * 9, refine_iters=1), c["noise_seed"] + 3),
                          # pad_sequence: every window padded to max_seq_len, the pad positions attendable (:186-193)
                          ("_pad_short", dict(k, timesteps=[2] + [0] * 9, refine_iters=1, pad_sequence=True), c["noise_seed"] + 4),
                          ("_pad_sliders", dict(k, pad_sequence=True), c["noise_seed"] + 5)):
        rng = np.random.default_rng(seed)
        noise = []
        for (a, b) in pipeline_windows(c["T"], kk["max_seq_len"], kk["overlap_buffer"]):
            width = kk["max_seq_len"] if kk.get("pad_sequence") else b - a
            for _ in range(kk["timesteps"][0] + kk["refine_iters"]):
                noise.append(torch.from_numpy(rng.standard_normal((2, 2, width)).astype(np.float32)))
        out["positions" + tag] = rh.reference_pipeline_positions(ref, seq_x, seq_o, seq_c, cv, ucv, noise,
                                                                 start_time=start_time, end_time=end_time,
                                                                 sliders=sliders if "sliders" in tag else (), **kk).numpy()
        if "short" not in tag:
            # the fp32 noise floor of the long runs: the SAME reference pipeline and draws around a second, independent fp32
            # CPU implementation of the denoiser (oracle/dit.py).  Where these two disagree a third fp32 implementation (the
            # device) cannot be asked to agree better -- the GPU gate is a multiple of this spread, not a constant.
            from oracle import dit as odit
            alt = odit.DiTOracle(sd, depth, hidden, heads)
            out["positions" + tag + "_alt"] = rh.reference_pipeline_positions(alt, seq_x, seq_o, seq_c, cv, ucv, list(noise),
                                                                              start_time=start_time, end_time=end_time,
                                                                              sliders=sliders if "sliders" in tag else (), **kk).numpy()
            d = np.abs(out["positions" + tag + "_alt"] - out["positions" + tag]).max(0)
            print("  floor", tag or "_full", "max", d.max(), "median", np.median(d), "p90", np.quantile(d, 0.9))
    pos = torch.from_numpy(out["positions"])
    np.savez_compressed(os.path.join(OUT, "dit_pipeline.npz"), case=json.dumps(c), start_time=start_time,
                        end_time=end_time, seq_c_slice=seq_c[:, ::37].numpy(), **out)
    print("dit_pipeline positions", tuple(pos.shape), "range", pos.min().item(), pos.max().item(),
          "moved", (pos[0] - torch.stack([torch.from_numpy(x), torch.from_numpy(y)])).abs().mean().item())


WHISPER_CASE = dict(seed=77, d=128, c_in=96, L=150)


def whisper_weights(c):
    """conv1 / conv2 weights + input of the front-end case, from the numpy seed (same draw order as the tests)"""
    rng = np.random.default_rng(c["seed"])
    d, cin, Ln = c["d"], c["c_in"], c["L"]
    rnd = lambda shape, std: torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32))
    w1, b1, w2, b2 = rnd((d, cin, 3), 0.08), rnd((d,), 0.05), rnd((d, d, 3), 0.06), rnd((d,), 0.05)
    return w1, b1, w2, b2, rnd((2, cin, Ln), 1.0)


def whisper_frontend_case():
    """Row a3: the conv front-end as the REFERENCE's `VarWhisperEncoder.forward` computes it
    (custom_transformers/modeling_varwhisper.py:779-780,813-816: conv1 k3 p1 -> gelu -> conv2 k3 s2 p1 -> gelu -> permute,
    no position table), captured at the input of its final norm with zero encoder layers; and the installed HF
    `WhisperEncoder` (the stock backbone `get_backbone_model` wires for openai/whisper names,
    modeling_mapperatorinator.py:35-39), which adds its sinusoid `embed_positions`."""
    c = WHISPER_CASE
    w1, b1, w2, b2, x = whisper_weights(c)
    rh.ref_shims.install()
    from osuT5.osuT5.model.custom_transformers import modeling_varwhisper as mv
    from transformers.models.whisper.modeling_whisper import WhisperConfig, WhisperEncoder
    out = {}
    for tag, enc in (("var", mv.VarWhisperEncoder(mv.VarWhisperConfig(
                         d_model=c["d"], num_mel_bins=c["c_in"], encoder_layers=0, decoder_layers=0,
                         encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256))),
                     ("hf", WhisperEncoder(WhisperConfig(
                         d_model=c["d"], num_mel_bins=c["c_in"], encoder_layers=0, decoder_layers=0,
                         encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256,
                         max_source_positions=c["L"] // 2)))):
        enc = enc.eval()
        with torch.no_grad():
            enc.conv1.weight.copy_(w1); enc.conv1.bias.copy_(b1); enc.conv2.weight.copy_(w2); enc.conv2.bias.copy_(b2)
        cap = {}
        enc.layer_norm.register_forward_pre_hook(lambda m, inp: cap.__setitem__("x", inp[0].detach().clone()))
        with torch.no_grad():
            enc(x)
        out[tag] = cap["x"]
        if tag == "hf":
            out["pos"] = enc.embed_positions.weight.detach().clone()
    np.savez_compressed(os.path.join(OUT, "whisper_frontend.npz"), out=out["hf"].numpy(), pos=out["pos"].numpy(),
                        out_var=out["var"].numpy(), **c)
    print("whisper_frontend", tuple(out["var"].shape), "hf - var - pos max abs",
          (out["hf"] - out["var"] - out["pos"]).abs().max().item())


def mel_case():
    a = synthetic_audio(2, 16000, seed=9)
    m = omel.mel_spectrogram(a)
    np.savez_compressed(os.path.join(OUT, "mel_oracle.npz"), audio_seed=9, n_samples=16000, mel=m.numpy())


def random_slider_cases(n_cases: int, seed: int):
    """(curve code, float32 control points, length) triples covering every curve type, repeated points (red anchors),
    degenerate (all points equal) and nearly collinear perfect curves"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_cases):
        ct = int(rng.integers(0, 4))
        ncp = 3 if (ct == 1 and rng.random() < 0.7) else int(rng.integers(2, 10))
        base = rng.uniform(50, 450, 2)
        cps = (base + rng.normal(0, rng.uniform(5, 150), (ncp, 2))).astype(np.float32)
        if rng.random() < 0.3 and ncp > 3:
            k = int(rng.integers(1, ncp - 1))
            cps[k + 1] = cps[k]
        if rng.random() < 0.04:
            cps[:] = cps[0]
        if ct == 1 and ncp == 3 and rng.random() < 0.2:
            cps[1] = (cps[0] + cps[2]) / 2 + rng.normal(0, 1e-3, 2).astype(np.float32)
        # what `to_positions` makes of normalised coordinates, so that a run from the normalised values sees THESE points
        v = torch.from_numpy(cps) / torch.tensor((512, 384)) * 2 - 1
        cps = (((v + 1) / 2) * torch.tensor((512, 384))).numpy()
        out.append((ct, cps, float(rng.uniform(0, 600)), v.numpy()))
    return out


def sliders_case():
    """End points of the reference's own SliderPath (slider_path.py) on random sliders: what `denoised_fn` writes into
    x2[slider.end_index] (diffusion_pipeline.py:213-219), float32 like x2."""
    rh.ref_shims.install()
    from osuT5.osuT5.inference.slider_path import SliderPath
    names = ["Linear", "PerfectCurve", "Catmull", "Bezier"]
    cases = random_slider_cases(800, 20240)
    types, cp_off, cps_all, v_all, lengths, ends, moved = [], [0], [], [], [], [], []
    for ct, cps, length, v in cases:
        path = SliderPath(names[ct], cps)
        total = path.get_distance()
        end = None if total == 0 else np.asarray(path.position_at(length / total))
        types.append(ct)
        cps_all.append(cps)
        v_all.append(v)
        cp_off.append(cp_off[-1] + len(cps))
        lengths.append(length)
        moved.append(end is not None)
        ends.append(np.zeros(2, np.float32) if end is None else end.astype(np.float32))
    np.savez_compressed(os.path.join(OUT, "sliders.npz"), type=np.array(types, np.int32), cp_off=np.array(cp_off, np.int32),
                        cps=np.concatenate(cps_all), v=np.concatenate(v_all), length=np.array(lengths, np.float64), end=np.stack(ends),
                        moved=np.array(moved))
    print("sliders", len(cases), "cases, moved", int(np.sum(moved)), "by type", np.bincount(types))


# (seed, objects, types_first, with_positions): the host code either side of the diffusion stage
EVENT_CASES = [(0, 22, False, False), (1, 22, True, False), (2, 22, False, True), (3, 22, True, True), (4, 60, False, False),
               (5, 60, True, False)]
CURVE_CODES = {"Bezier": 0, "PerfectCurve": 1, "Catmull": 2}
GEN_CASE = dict(preset="DiT-XS", weight_seed=33, event_seed=12, objects=70, tokenizer_seed=8, noise_seed=21,
                config=dict(beatmap_id=None, difficulty=5.3, mapper_id=None, circle_size=4.2, slider_multiplier=1.7,
                            descriptors=["d1", "d0"], negative_descriptors=["d2"]),
                knobs=dict(timesteps=[2, 0, 0, 0, 0, 0, 0, 0, 0, 0], seq_len=32, max_seq_len=96, overlap_buffer=12,
                           cfg_scale=1.5, refine_iters=1))


def events_case():
    """Row a14's host code run through the REFERENCE: `update_event_times`, `events_to_sequence` (get_groups inside),
    `events_with_pos`, and one whole `DiffisionPipeline.generate` (events in, events out, nothing replaced) on the event
    streams of `mh_testing.synthetic_event_stream`."""
    import types
    from mh_testing import (pipeline_windows, synthetic_diffusion_tokenizer_state, synthetic_event_stream,
                                              synthetic_timing)
    rh.ref_shims.install()
    import diffusion_pipeline as dp
    from osuT5.osuT5.dataset.data_utils import update_event_times
    from osuT5.osuT5.tokenizer import Event, EventType
    out = {}
    for seed, n_obj, tf, wp in EVENT_CASES:
        ev = synthetic_event_stream(n_obj, seed, types_first=tf, with_positions=wp)
        rev = [Event(EventType[e.type.name], e.value) for e in ev]
        times = []
        update_event_times(rev, times, types_first=tf)
        pipe = object.__new__(dp.DiffisionPipeline)
        pipe.types_first, pipe.has_sv = tf, True
        seq_x, seq_o, seq_c, n, seq_indices, sliders = pipe.events_to_sequence(rev, synthetic_timing(seed), 1.4)
        pos = torch.from_numpy(np.random.default_rng(seed).uniform(0, 512, (2, n)).astype(np.float32))
        placed = dp.DiffisionPipeline.events_with_pos(rev, pos, seq_indices)
        names = sorted({e.type.name for e in placed})
        k = f"s{seed}_"
        out.update({k + "times": np.array(times, np.int64), k + "seq_x": seq_x.numpy(), k + "seq_o": seq_o.numpy(),
                    # full conditioning for the short streams; type rows + every 7th column of the embeddings for the long ones
                    **({k + "seq_c": seq_c.numpy()} if n_obj <= 30 else {k + "seq_c_types": seq_c[256:].numpy(), k + "seq_c_cols": seq_c[:, ::7].numpy()}),
                    k + "seq_indices": np.array([seq_indices[i] for i in range(len(rev))], np.int32),
                    k + "slider_off": np.cumsum([0] + [len(sl.seq_indices) for sl in sliders]).astype(np.int32),
                    k + "slider_idx": np.concatenate([sl.seq_indices for sl in sliders] + [np.zeros(0, np.int64)]).astype(np.int32),
                    k + "slider_end": np.array([sl.end_index for sl in sliders], np.int32),
                    k + "slider_curve": np.array([CURVE_CODES[sl.curve_type] for sl in sliders], np.int32),
                    k + "slider_length": np.array([sl.length for sl in sliders], np.float64),
                    k + "placed_names": np.array(names), k + "placed_type": np.array([names.index(e.type.name) for e in placed], np.int32),
                    k + "placed_value": np.array([e.value for e in placed], np.int64)})
        print("events", seed, "events", len(rev), "points", n, "sliders", len(sliders))

    g = GEN_CASE
    depth, hidden, heads = DIT_PRESETS[g["preset"]]
    state = synthetic_diffusion_tokenizer_state(g["tokenizer_seed"])
    from osu_diffusion.utils.tokenizer import Tokenizer
    tok = Tokenizer()
    tok.load_state_dict(state)
    sd = random_dit_state_dict(depth, hidden, seed=g["weight_seed"], class_size=tok.num_tokens)
    from osu_diffusion.utils.models import DiT
    ref = DiT(context_size=272, hidden_size=hidden, depth=depth, num_heads=heads, class_size=tok.num_tokens).eval()
    ref.load_state_dict(sd, strict=True)
    ev = synthetic_event_stream(g["objects"], g["event_seed"])
    timing = synthetic_timing(g["event_seed"])
    pipe = object.__new__(dp.DiffisionPipeline)
    pipe.types_first, pipe.has_sv = False, True
    n = pipe.events_to_sequence([Event(EventType[e.type.name], e.value) for e in ev], timing, 1.0)[3]
    kk = g["knobs"]
    rng = np.random.default_rng(g["noise_seed"])
    noise = [torch.from_numpy(rng.standard_normal((2, 2, b - a)).astype(np.float32))
             for (a, b) in pipeline_windows(n, kk["max_seq_len"], kk["overlap_buffer"])
             for _ in range(kk["timesteps"][0] + kk["refine_iters"])]
    placed = rh.reference_pipeline_generate(ref, ev, types.SimpleNamespace(**g["config"]), timing, state, noise, **kk)
    names = sorted({nm for nm, _ in placed})
    out.update(gen_case=json.dumps(g), gen_points=n, gen_names=np.array(names),
               gen_type=np.array([names.index(nm) for nm, _ in placed], np.int32), gen_value=np.array([v for _, v in placed], np.int64))
    print("generate: points", n, "events", len(ev), "->", len(placed))
    np.savez_compressed(os.path.join(OUT, "events_to_sequence.npz"), cases=json.dumps(EVENT_CASES), **out)


def main(only=None):
    """`python -m oracle.make_golden` regenerates everything; `python -m oracle.make_golden NAME ...` only the named
    fixtures (t5_tiny, t5_small, t5_base, t5_large, vw_test, vw_test_nobias, vw_small, rw_test, rw_test_cond, rw_small, hfw_test,
    hfw_small, t5_base_wide, t5_base_bf16ref, t5_base_wide_bf16ref, t5_tiny_cond,
    t5_tiny_tf, dit_xs, dit_s, dit_b, dit_b_1024, dit_pipeline, sliders, events, whisper_frontend, mel_oracle, tokenizer)."""
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cases = {"tokenizer": tokenizer_case, "mel_oracle": mel_case, "whisper_frontend": whisper_frontend_case}
    for name in T5_CASES:
        cases[name] = (lambda n: (lambda: t5_case(n)))(name)
    for name in VW_CASES:
        cases[name] = (lambda n: (lambda: vw_case(n)))(name)
    for name in WF_CASES:
        cases[name] = (lambda n: (lambda: wf_case(n)))(name)
    cases.update({
        "t5_base_bf16ref": lambda: t5_bf16_reference_case("t5_base"),
        "t5_base_wide_bf16ref": lambda: t5_bf16_reference_case("t5_base_wide"),
        "t5_tiny_cond": t5_conditioning_case,
        "t5_tiny_tf": types_first_case,
        "dit_xs": lambda: dit_case("dit_xs", "DiT-XS", 96, 21, 5, 1.5),
        "dit_s": lambda: dit_case("dit_s", "DiT-S", 160, 1, 2, 2.0),
        # BASELINE configs[4]: DiT-B (osu_diffusion/utils/models.py:392) at a chunk-sized and at a full 1024-point window
        "dit_b": lambda: dit_case("dit_b", "DiT-B", 256, 4, 6, 1.5),
        "dit_b_1024": lambda: dit_case("dit_b_1024", "DiT-B", 1024, 4, 8, 2.0, loop_steps=3),
        "dit_pipeline": pipeline_case,
        "t5_tiny_beam": beam_case,
        "sliders": sliders_case,
        "events": events_case,
    })
    for name, fn in cases.items():
        if only and name not in only:
            continue
        fn()


if __name__ == "__main__":
    import sys
    main(set(sys.argv[1:]) or None)
