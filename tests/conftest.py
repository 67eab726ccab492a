import os
import sys

# The CPU suite's oracle runs are multi-threaded torch; on a shared host (this container is a VM whose cores come and go) OpenMP's
# default spin-waiting turns a descheduled worker into minutes of busy waiting for everyone else (seen twice: the t5_base oracle case
# making no progress at 600 % CPU).  Passive waiting costs a few percent on an idle host and cannot livelock.  Before torch is imported.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("KMP_BLOCKTIME", "0")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def ts_range(tok):
    s = [v for k, v in tok.event_start.items() if k.name == "TIME_SHIFT"][0]
    e = [v for k, v in tok.event_end.items() if k.name == "TIME_SHIFT"][0]
    return s, e


def t5_golden_case(name):
    """tests/golden/<name>.npz (oracle/make_golden.py:t5_case) -> (golden, size, tok, state_dict, audio, src, tgt):
    weights and audio are regenerated from the seeds / kinds recorded in the fixture."""
    import numpy as np

    from mapperatorinator_amd import Tokenizer
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mh_testing import DIVERSE_GAINS, random_t5_state_dict, synthetic_audio, synthetic_audio_varied
    g = np.load(f"{GOLDEN}/{name}.npz")
    size = name.split("_")[1]
    src, tgt = int(g["src_len"]), int(g["tgt_len"])
    tok = Tokenizer.benchmark_vocab(src_seq_len=src)
    assert tok.vocab_size_out == int(g["vocab_out"]) and tok.vocab_size_in == int(g["vocab_in"])
    gains = DIVERSE_GAINS if ("gains" in g.files and str(g["gains"]) == "diverse") else None
    sd = random_t5_state_dict(T5_PRESETS[size], tok.vocab_size_in, tok.vocab_size_out, seed=int(g["weight_seed"]),
                              lm_head_gain=float(g["lm_head_gain"]), gains=gains)
    varied = "audio_kind" in g.files and str(g["audio_kind"]) == "varied"
    audio = (synthetic_audio_varied if varied else synthetic_audio)(g["prompt"].shape[0], int(g["n_samples"]),
                                                                    seed=int(g["audio_seed"]))
    return g, size, tok, sd, audio, src, tgt


def vw_golden_case(name):
    """tests/golden/vw_*.npz (oracle/make_golden.py:vw_case) -> (golden, dims, tok, state_dict, audio): the Whisper-family
    backbone on the reference; weights and audio regenerated from the recorded seeds."""
    import numpy as np

    from mapperatorinator_amd import Tokenizer
    from mh_testing import random_varwhisper_state_dict, synthetic_audio_varied
    from mapperatorinator_amd.whisper_engine import VARWHISPER_PRESETS
    g = np.load(f"{GOLDEN}/{name}.npz")
    d = VARWHISPER_PRESETS[str(g["size"])]
    tok = Tokenizer.benchmark_vocab(src_seq_len=int(g["in_frames"]))
    assert tok.vocab_size_out == int(g["vocab_out"]) and tok.vocab_size_in == int(g["vocab_in"])
    sd = random_varwhisper_state_dict(d.d_model, d.n_heads, d.n_enc_layers, d.n_dec_layers, d.d_ff, tok.vocab_size_in,
                                      tok.vocab_size_out, seed=int(g["weight_seed"]), head_gain=float(g["head_gain"]),
                                      attention_bias=bool(g["attention_bias"]), gains={"decoder_embedder": 0.5})
    audio = synthetic_audio_varied(g["prompt"].shape[0], int(g["n_samples"]), seed=int(g["audio_seed"]))
    return g, d, tok, sd, audio


def wf_golden_case(name):
    """tests/golden/rw_*.npz / hfw_*.npz (oracle/make_golden.py:wf_case) -> (golden, kind, dims, tok, state_dict, audio, cond inputs):
    'Tiger14n/ropewhisper-*' / 'openai/whisper-*' on the reference; weights and audio regenerated from the recorded seeds.  The
    cond inputs are dict(difficulty, mapper_idx, song_position) tensors or None."""
    import numpy as np
    import torch

    from mapperatorinator_amd import Tokenizer
    from mh_testing import add_random_cond_embedders, random_whisper_family_state_dict, synthetic_audio_varied
    from mapperatorinator_amd.whisper_engine import VARWHISPER_PRESETS
    g = np.load(f"{GOLDEN}/{name}.npz")
    kind = str(g["kind"])
    d = VARWHISPER_PRESETS[str(g["size"])]
    frames, tgt = int(g["in_frames"]), int(g["tgt_len"])
    tok = Tokenizer.benchmark_vocab(src_seq_len=frames)
    assert tok.vocab_size_out == int(g["vocab_out"]) and tok.vocab_size_in == int(g["vocab_in"])
    has_cond = "cond_dim" in g.files
    cdim = int(g["cond_dim"]) if has_cond else 0
    sd = random_whisper_family_state_dict(kind, d.d_model, d.n_heads, d.n_enc_layers, d.n_dec_layers, d.d_ff, tok.vocab_size_in,
                                          tok.vocab_size_out, int(g["n_mels"]), src_positions=frames // 2, tgt_positions=tgt,
                                          cond_size=3 * cdim, seed=int(g["weight_seed"]), head_gain=float(g["head_gain"]),
                                          gains={"decoder_embedder": 0.5})
    cond = None
    if has_cond:
        add_random_cond_embedders(sd, cdim, int(g["num_mappers"]), seed=int(g["cond_seed"]))
        cond = dict(difficulty=torch.from_numpy(g["difficulty"]), mapper_idx=torch.from_numpy(g["mapper_idx"]),
                    song_position=torch.from_numpy(g["song_position"]))
    audio = synthetic_audio_varied(g["prompt"].shape[0], int(g["n_samples"]), seed=int(g["audio_seed"]))
    return g, kind, d, tok, sd, audio, cond


def assert_topk_scores_match(dumped, g, P, tol):
    """dumped: fp32 (cols, B, V) processed scores of a run (index = produced column); g: a fixture holding the
    reference's per-step `top_vals` / `top_ids` (steps, B, K) and `lse` (steps, B).  The K best ids of every step must
    be the same set in the same order wherever the reference separates them by more than 2*tol, values and the
    row's logsumexp within tol."""
    import numpy as np
    import torch
    tv, ti, lse = torch.from_numpy(g["top_vals"]), torch.from_numpy(g["top_ids"]).long(), torch.from_numpy(g["lse"])
    steps, B, K = tv.shape
    worst = 0.0
    for i in range(steps):
        s = dumped[P + i].float().cpu()
        got = s.gather(-1, ti[i])
        worst = max(worst, (got - tv[i]).abs().max().item(), (torch.logsumexp(s, -1) - lse[i]).abs().max().item())
        gv, gi = s.topk(K, dim=-1)
        sep = (tv[i][:, :-1] - tv[i][:, 1:]) > 2 * tol           # rank r clearly above rank r+1 in the reference
        clear = torch.cat([sep, torch.zeros(B, 1, dtype=torch.bool)], 1)
        clear[:, 1:] &= sep                                        # ... and clearly below rank r-1
        assert torch.equal(gi[clear], ti[i][clear]), f"step {i}: ranking differs from the reference"
    assert worst < tol, worst
    return worst


def types_first_case():
    """tests/golden/t5_tiny_tf.npz: the reference's `model_generate` under the types_first processors and
    classifier-free guidance (oracle/make_golden.py:types_first_case).  Returns (golden, tok, sd, audio, tgt, runs)."""
    import json

    import numpy as np

    from mapperatorinator_amd import Tokenizer
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mh_testing import boost_timed_rows, random_t5_state_dict, synthetic_audio
    g = np.load(f"{GOLDEN}/t5_tiny_tf.npz")
    tok = Tokenizer.from_json(f"{GOLDEN}/tokenizer_types_first.json")
    assert tok.vocab_size_out == int(g["vocab_out"]) and tok.vocab_size_in == int(g["vocab_in"])
    sd = random_t5_state_dict(T5_PRESETS["tiny"], tok.vocab_size_in, tok.vocab_size_out, seed=int(g["weight_seed"]),
                              lm_head_gain=float(g["lm_head_gain"]))
    boost_timed_rows(sd, tok, float(g["timed_gain"]))
    audio = synthetic_audio(g["prompt"].shape[0], int(g["n_samples"]), seed=int(g["audio_seed"]))
    return g, tok, sd, audio, int(g["tgt"]), json.loads(str(g["runs"]))


def oracle_processor_kwargs(sp):
    """MhSampling (server.build_sampling) -> the keyword arguments of oracle.t5.T5Oracle.generate."""
    import numpy as np

    from mapperatorinator_amd.server import FLAG_COND0, FLAG_LOOKBACK_EOS, FLAG_TIMED
    fl = sp.host_tok_flags
    rules = [(sp.cond_temp[j], set(np.nonzero(fl & (FLAG_COND0 << j))[0].tolist()), sp.cond_offset[j])
             for j in range(sp.n_cond)]
    ltf = None
    if sp.lookback_types_first:
        ltf = dict(eos_ids=np.nonzero(fl & FLAG_LOOKBACK_EOS)[0].tolist(), timed_ids=np.nonzero(fl & FLAG_TIMED)[0].tolist())
    return dict(temperature=sp.temperature, timeshift_bias=sp.timeshift_bias, lookback_mask_end=sp.lookback_mask_end,
                cfg_scale=sp.cfg_scale, cond_rules=rules, lookback_types_first=ltf)


def assert_scores_close(got, want, tol, eos_extra_slot=None):
    """processed scores: same -inf pattern, finite entries within `tol`.  `eos_extra_slot` (LookbackBias
    types_first): that id holds log(clip((s-1)*p_eos/p_event, 0, 1)) whose argument is a difference of nearly equal
    fp32 numbers -- compared as a probability instead; the renormalised scores are log(probability), where torch's
    exp underflows to -inf below about -87 .. -103 while the log-domain form on the device stays finite: scores under
    -80 (probability < 2e-35) count as -inf on both sides."""
    import torch
    got, want = got.clone(), want.clone()
    if eos_extra_slot is not None:
        assert (got[:, eos_extra_slot].exp() - want[:, eos_extra_slot].exp()).abs().max().item() < 1e-5
        got[:, eos_extra_slot] = 0
        want[:, eos_extra_slot] = 0
        got[got < -80] = float("-inf")
        want[want < -80] = float("-inf")
    fa, fb = torch.isfinite(got), torch.isfinite(want)
    assert torch.equal(fa, fb), (fa != fb).nonzero()[:5]
    err = (got[fa] - want[fa]).abs().max().item()
    assert err < tol, err
    return err
