"""not gpu: host logic + the C-ABI library loads and exports every symbol include/mapperhip.h declares
(no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, ts_range
from mapperatorinator_amd import ContextType, Event, EventType, Tokenizer, _lib


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mapperhip.h")).read()
    declared = set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mh_abi_version() == _lib.ABI_VERSION
    assert lib.mh_last_error() is not None


def test_argument_validation_without_gpu():
    """Entry points validate before touching the device: bad descriptors return MH_ERR_ARG + a message."""
    lib = _lib.load()
    g = _lib.MhGemm()
    assert lib.mh_gemm(C.byref(g), None) == -1
    assert b"null operand" in lib.mh_last_error()
    assert lib.mh_gemm(None, None) == -1
    cfg = _lib.MhT5Config(128, 32, 256, 2, 2, 2, 10, 10, 388, 416, 251, 48, 0, 1e-6)
    assert lib.mh_t5_encode(C.byref(cfg), None, None, 1, None, None, None, 0, None) == -1
    assert b"d_kv" in lib.mh_last_error()
    with pytest.raises(RuntimeError, match="status -1"):
        _lib.check(-1, "x")
    assert lib.mh_t5_encode_workspace_bytes(C.byref(cfg), 0) == -1
    cfg.d_kv = 64
    assert lib.mh_t5_encode_workspace_bytes(C.byref(cfg), 2) > 0
    assert lib.mh_t5_decode_workspace_bytes(C.byref(cfg), 2) > 0
    dc = _lib.MhDiTConfig(128, 2, 2, 272, 300, 2, 128, 256, 544, 300)
    assert lib.mh_dit_workspace_bytes(C.byref(dc), 2, 96) > 0


def test_struct_layouts_match_header_sizes():
    # pointer arrays of MH_MAX_LAYERS entries, ints packed as in C
    assert C.sizeof(_lib.MhT5Config) == 14 * 4 + 5 * 4 + 4 + 8 + 4 + 4    # ABI 5: arch, attn_scale, in_frames, local_every, local_window; ABI 7: enc_operand_dtype; ABI 8: options; ABI 10: dec_pos_from_mask + tail pad
    assert C.sizeof(_lib.MhDiTConfig) == 11 * 4 + 4 + 8                    # 4B pad before the ABI 8 options pointer
    base = 4 * 8 + 16 * 4 + 3 * 4 + 4 + 8                                # 4B pad before the uint64 seed
    assert C.sizeof(_lib.MhSampling) == base + 9 * 4 + 4 + 8 + 2 * 4 + 8   # ABI 2 tail: 9 words, pad, tok_flags; ABI 3: 2 words; ABI 4: cross_kv_fp8
    assert C.sizeof(_lib.MhT5Weights) == 8 * (5 + 6 * 32 + 1 + 5 * 32 + 1 + 4 * 32 + 2) + 8 * (4 + 4 * 32 + 3 * 32 + 1 + 3 * 32 + 4) + 8 * (8 * 32 + 2) + 8 * (5 * 32 + 4)   # + ABI 5 (arch 1) + ABI 7 (MX-fp8 copies) + ABI 10 (arch 2: LayerNorm biases, position tables)
    assert C.sizeof(_lib.MhDiTWeights) == 8 * (12 + 10 * 32 + 4 + 1 + 4 * 32 + 4 * 32 + 8 * 32)   # + the pre-split (bf16 x 3), the bf16 and (ABI 7) the MX-fp8 copies


def test_option_sets_override_and_fall_through_without_gpu():
    """ABI 8: an MhOptionSet starts empty (falls through to the process-wide value), holds its own overrides, refuses unknown
    names, and workspace sizing follows the set named by the config (decode_chains changes nothing there, but the call must
    resolve options through the set: mh_t5_decode_chains_cfg)."""
    lib = _lib.load()
    a, b = _lib.OptionSet(dict(decode_chains=1)), _lib.OptionSet()
    assert a["decode_chains"] == 1 and b["decode_chains"] == lib.mh_get_option(b"decode_chains")
    b["decode_chains"] = 3
    assert (a["decode_chains"], b["decode_chains"]) == (1, 3)
    old = _lib.set_option("gemm_splitk_tiles", 77)
    try:
        assert a["gemm_splitk_tiles"] == 77                 # no override: the process-wide value
        a["gemm_splitk_tiles"] = 5
        assert a["gemm_splitk_tiles"] == 5 and b["gemm_splitk_tiles"] == 77 and lib.mh_get_option(b"gemm_splitk_tiles") == 77
        a.clear("gemm_splitk_tiles")
        assert a["gemm_splitk_tiles"] == 77
    finally:
        _lib.set_option("gemm_splitk_tiles", old)
    with pytest.raises(RuntimeError, match="unknown option"):
        a["nope"] = 1
    assert lib.mh_options_get(a.handle, b"nope") == -1
    cfg = _lib.MhT5Config(128, 64, 256, 2, 2, 2, 10, 10, 388, 416, 251, 48, 0, 1e-6)
    assert lib.mh_t5_decode_chains_cfg(C.byref(cfg), 32) == lib.mh_t5_decode_chains(32)
    cfg.options = a.handle
    assert lib.mh_t5_decode_chains_cfg(C.byref(cfg), 32) == 1
    cfg.options = b.handle
    assert lib.mh_t5_decode_chains_cfg(C.byref(cfg), 32) == 3
    b.clear()
    assert lib.mh_t5_decode_chains_cfg(C.byref(cfg), 32) == lib.mh_t5_decode_chains(32)


def test_no_cpu_fallback():
    from mapperatorinator_amd.mel import MelSpectrogram
    m = MelSpectrogram()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4096))
    if not torch.cuda.is_available():
        from mapperatorinator_amd.t5_engine import T5Engine, T5_PRESETS
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            T5Engine({}, T5_PRESETS["tiny"], 10, 10)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mapperatorinator_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
    assert not re.search(r"^\s*(from|import)\s+oracle", open(os.path.join(ROOT, "bench.py")).read().split("def cpu_baseline")[0], re.M)


def test_tokenizer_api():
    tok = Tokenizer.benchmark_vocab()
    assert (tok.pad_id, tok.sos_id, tok.eos_id) == (0, 1, 2)
    ts0, ts1 = ts_range(tok)
    assert (ts0, ts1) == (3, 1004)
    for ev in (Event(EventType.TIME_SHIFT, 0), Event(EventType.TIME_SHIFT, 1000), Event(EventType.DISTANCE, 640),
               Event(EventType.MEASURE, 0), Event(EventType.HITSOUND, 71)):
        assert tok.decode(tok.encode(ev)) == ev
    with pytest.raises(ValueError):
        tok.encode(Event(EventType.TIME_SHIFT, 1001))
    with pytest.raises(ValueError):
        tok.decode(tok.vocab_size_in)
    assert tok.event_type_range(EventType.SNAPPING) == (1004, 1020)
    t2 = Tokenizer.from_ranges([(EventType.TIME_SHIFT, -512, 512), (EventType.SNAPPING, 0, 16)],
                               [(EventType.DIFFICULTY, 0, 9)], [ContextType.MAP, "gd"])
    assert t2.context_sos == {ContextType.MAP: 3, ContextType.GD: 5} and t2.context_eos[ContextType.GD] == 6
    assert t2.event_start[EventType.TIME_SHIFT] == 7 and t2.vocab_size_in == t2.vocab_size_out + 10
    t3 = Tokenizer().load_state_dict(t2.state_dict())
    assert t3.state_dict() == t2.state_dict()
    assert repr(Event(EventType.TIME_SHIFT, 12)) == "t12"


def test_eos_set_and_sampling_translation():
    from mapperatorinator_amd.server import build_sampling, get_eos_token_id
    tok = Tokenizer.from_ranges([(EventType.TIME_SHIFT, 0, 1000), (EventType.SNAPPING, 0, 16)], (), [ContextType.MAP])
    ts0, ts1 = ts_range(tok)
    assert get_eos_token_id(tok) == [2]
    assert get_eos_token_id(tok, context_type=ContextType.MAP) == [2, 4]
    ids = get_eos_token_id(tok, lookback_time=50, lookahead_time=30, context_type="map")
    assert ids == [2, 4] + list(range(ts0, ts0 + 5)) + list(range(ts1 - 3, ts1))
    sp, eos = build_sampling(tok, dict(do_sample=True, top_k=5, top_p=0.9, temperature=0.8, timeshift_bias=0.2,
                                       lookback_time=120, max_length=77, context_type="map", pad_token_id=0), 512)
    assert (sp.do_sample, sp.top_k, sp.max_length, sp.ts_start, sp.ts_end) == (1, 5, 77, ts0, ts1)
    assert abs(sp.top_p - 0.9) < 1e-6 and abs(sp.temperature - 0.8) < 1e-6
    assert sp.n_sos == 2 and list(sp.sos_ids[:2]) == [1, 3]
    assert sp.lookback_mask_end == ts0 + 12
    assert eos[:2] == [2, 4]
    assert sp.cfg_scale == 1.0 and sp.n_cond == 0 and sp.lookback_types_first == 0 and sp.host_tok_flags is None
    assert build_sampling(tok, dict(num_beams=2), 512)[0].num_beams == 2       # beam search: mapperatorinator_amd/beam.py
    bs = build_sampling(tok, dict(num_beams=2, do_sample=True, top_k=30, top_p=0.9), 512)[0]      # beam-sample (round 5)
    assert (bs.num_beams, bs.do_sample, bs.top_k) == (2, 1, 30) and abs(bs.top_p - 0.9) < 1e-6


def test_types_first_sampling_translation():
    """ConditionalTemperature rules + the LookbackBias(types_first) tables, on the reference's own tokenizer state
    (tests/golden/tokenizer_types_first.json; rule construction: logit_processors.py:59-73, tables :99-108)."""
    from conftest import GOLDEN
    from mapperatorinator_amd.server import (FLAG_COND0, FLAG_LOOKBACK_EOS, FLAG_TIMED, build_sampling,
                                             get_beat_type_tokens, get_mania_type_tokens, get_scroll_speed_tokens)
    tok = Tokenizer.from_json(f"{GOLDEN}/tokenizer_types_first.json")
    assert get_beat_type_tokens(tok) == (2068, 2069, 2070)
    assert get_mania_type_tokens(tok) == (2058, 2073, 2074)
    assert get_scroll_speed_tokens(tok) == tuple(range(882, 1883))
    kw = dict(types_first=True, temperature=0.9, timing_temperature=0.5, mania_column_temperature=0.9,
              taiko_hit_temperature=0.7, lookback_time=500, cfg_scale=2.0, context_type="map")
    sp, eos = build_sampling(tok, kw, 64)
    assert sp.cfg_scale == 2.0 and sp.lookback_types_first == 1
    # the mania rule is dropped (its temperature equals the default), the others keep their order and offsets
    assert sp.n_cond == 2 and [round(sp.cond_temp[j], 6) for j in range(2)] == [0.5, 0.7]
    assert [sp.cond_offset[j] for j in range(2)] == [1, 1]
    fl = sp.host_tok_flags
    assert fl.shape == (tok.vocab_size_out,)
    assert np.nonzero(fl & FLAG_COND0)[0].tolist() == [2068, 2069, 2070]
    assert np.nonzero(fl & (FLAG_COND0 << 1))[0].tolist() == list(range(882, 1883))
    assert np.nonzero(fl & FLAG_LOOKBACK_EOS)[0].tolist() == [2, 4]
    timed = np.nonzero(fl & FLAG_TIMED)[0].tolist()
    assert 2058 in timed and 2068 in timed and 2079 in timed and 2062 not in timed and 5 not in timed
    assert sp.lookback_mask_end == sp.ts_start + 50
    # types_first=False: conditional temperatures are ignored (reference prints a warning and drops them)
    sp2, _ = build_sampling(tok, dict(kw, types_first=False), 64)
    assert sp2.n_cond == 0 and sp2.lookback_types_first == 0 and sp2.host_tok_flags is None


def test_rel_bias_tables():
    from mapperatorinator_amd.t5_engine import T5_PRESETS, rel_bias_tables
    d = T5_PRESETS["tiny"]
    enc_tab = torch.arange(32 * 2, dtype=torch.float32).reshape(32, 2)
    dec_tab = -enc_tab
    e, dcd = rel_bias_tables(enc_tab, dec_tab, 50, 20, d)
    assert e.shape == (2, 99) and dcd.shape == (2, 20)
    assert e[0, 49].item() == enc_tab[0, 0]                # rel 0 -> bucket 0
    assert e[1, 49 + 3].item() == enc_tab[16 + 3, 1]       # k > q -> upper half
    assert e[0, 49 - 3].item() == enc_tab[3, 0]
    assert dcd[0, 5].item() == dec_tab[5, 0] and dcd[1, 16].item() == dec_tab[16, 1]


def test_band_mask_recovery_and_inpaint():
    from mapperatorinator_amd.dit import DiTHIP, InpaintSpec
    from oracle.dit import band_mask
    f = DiTHIP.band_from_mask
    self = type("X", (), {"_band_cache": {}})()
    assert f(self, None, 10) == (0, 0)
    assert f(self, band_mask(300, 128), 300) == (128, 0)
    assert f(self, band_mask(96, 128), 96) == (0, 0)
    assert f(self, band_mask(200, 7), 200) == (7, 0)
    bad = band_mask(200, 7)
    bad[100, 100] = True
    with pytest.raises(NotImplementedError):
        f(self, bad, 200)
    m = torch.tensor([[True, False]])
    assert torch.equal(InpaintSpec(m, torch.tensor([[9.0, 9.0]]))(torch.tensor([[1.0, 2.0]])), torch.tensor([[1.0, 9.0]]))


def test_mel_host_tables():
    from mapperatorinator_amd.mel import MelSpectrogram, slaney_filterbank
    from oracle import mel as omel
    fb = slaney_filterbank(16000, 1024, 388, 0.0, 8000.0)
    assert np.array_equal(fb, omel.mel_filterbank(16000, 1024, 388, 0, 8000))
    m = MelSpectrogram()
    dense = np.zeros_like(fb)
    for i in range(388):
        s, ln, of = int(m.fb_start[i]), int(m.fb_len[i]), int(m.fb_off[i])
        dense[i, s:s + ln] = m.fb_w[of:of + ln].numpy()
    assert np.array_equal(dense, fb)
    assert m.n_frames(160000) == 1251
    with pytest.raises(NotImplementedError):
        MelSpectrogram(implementation="librosa")
    with pytest.raises(NotImplementedError):
        MelSpectrogram(pad_mode="replicate")
    # the torchaudio parameterisation of the Whisper-family configs: HTK triangles without area normalisation; the table
    # the kernel receives equals what the torch.stft restatement multiplies by (recovered from an impulse power spectrum)
    from mapperatorinator_amd.mel import htk_filterbank
    t = MelSpectrogram(implementation="torchaudio", log_scale=True, n_mels=128, f_min=20, pad_mode="reflect")
    assert t.reflect and t.log_scale and t.n_mels == 128
    hfb = htk_filterbank(16000, 1024, 128, 20.0, 8000.0)
    assert hfb.shape == (128, 513) and hfb.min() >= 0 and hfb.max() <= 1.0 + 1e-6
    peaks = hfb.argmax(1)
    assert (np.diff(peaks) >= 0).all() and hfb[:, :1].sum() == 0          # rising centre frequencies, nothing below 20 Hz
    dense = np.zeros_like(hfb)
    for i in range(128):
        s_, ln, of = int(t.fb_start[i]), int(t.fb_len[i]), int(t.fb_off[i])
        dense[i, s_:s_ + ln] = t.fb_w[of:of + ln].numpy()
    assert np.array_equal(dense, hfb)
    # both PRODUCT tables against an independent third-party implementation (transformers.audio_utils; nnAudio / torchaudio
    # themselves are absent from the image): every triangle edge on the same bin
    au = pytest.importorskip("transformers.audio_utils")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_sl = au.mel_filter_bank(513, 388, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
        ref_htk = au.mel_filter_bank(513, 128, 20.0, 8000.0, 16000, norm=None, mel_scale="htk").T
    assert np.abs(fb - ref_sl).max() < 5e-8 and np.array_equal(fb > 0, ref_sl.astype(np.float32) > 0)
    assert np.abs(hfb - ref_htk).max() < 5e-7 and np.array_equal(hfb > 0, ref_htk.astype(np.float32) > 0)


def test_diffusion_pipeline_host_glue():
    """points_to_sequence / band mask / window plan of the diffusion stage (diffusion_pipeline.py:145-148,277-278,
    361-387) -- conditioning rows pinned by the reference golden (which asserted equality with the reference's
    `timestep_embedding` when it was generated)."""
    import json

    from conftest import GOLDEN
    from mapperatorinator_amd.diffusion_pipeline import band_mask, points_to_sequence, repeat_type
    from mh_testing import pipeline_windows, synthetic_hit_objects
    g = np.load(f"{GOLDEN}/dit_pipeline.npz")
    c = json.loads(str(g["case"]))
    x, y, times, dist, typ = synthetic_hit_objects(c["T"], c["point_seed"])
    seq_x, seq_o, seq_c = points_to_sequence(x, y, times, dist, typ)
    assert seq_x.shape == (2, c["T"]) and seq_c.shape == (272, c["T"])
    assert np.array_equal(seq_c[:, ::37].numpy(), g["seq_c_slice"])
    assert torch.equal(seq_c[256:].sum(0), torch.ones(c["T"])) and torch.equal(seq_c[256:].argmax(0), torch.as_tensor(typ))
    assert float(seq_x.min()) >= -1 and float(seq_x.max()) <= 1
    # the band mask, column by column as the reference builds it
    T, sl = 50, 8
    ref = torch.full((T, T), True)
    for i in range(T):
        ref[max(0, i - sl): min(T, i + sl), i] = False
    assert torch.equal(band_mask(T, sl), ref)
    assert pipeline_windows(300, 160, 16) == [(0, 160), (128, 288), (256, 300)]
    assert pipeline_windows(100, 1024, 128) == []      # shorter than two buffers: the reference loop does not run
    assert [repeat_type(r) for r in (1, 2, 3, 4, 5, 6, 7)] == [0, 1, 2, 3, 4, 3, 4]
    with pytest.raises(ValueError):
        points_to_sequence(x, y, times, dist, typ + 16)


def _golden_events():
    import json
    g = np.load(os.path.join(ROOT, "tests", "golden", "events_to_sequence.npz"))
    return g, [tuple(c) for c in json.loads(str(g["cases"]))]


def test_events_to_sequence_matches_the_reference_golden():
    """Row a14's host code: event times, hit-object grouping, points, `seq_indices`, the DiffusionSlider list and the
    write-back of positions, against the outputs of the reference's own `update_event_times` / `events_to_sequence` /
    `events_with_pos` (tests/golden/events_to_sequence.npz, oracle/make_golden.py `events`) -- bit for bit: it is integer
    bookkeeping plus the same torch ops in the same order."""
    from mapperatorinator_amd import diffusion_pipeline as dp
    from mh_testing import synthetic_event_stream, synthetic_timing
    g, cases = _golden_events()
    curves = ["Bezier", "PerfectCurve", "Catmull"]
    for seed, n_obj, tf, wp in cases:
        k = f"s{seed}_"
        ev = synthetic_event_stream(n_obj, seed, types_first=bool(tf), with_positions=bool(wp))
        assert dp.event_times(ev, types_first=bool(tf)) == g[k + "times"].tolist()
        seq_x, seq_o, seq_c, n, seq_indices, sliders = dp.events_to_sequence(ev, synthetic_timing(seed), 1.4, types_first=bool(tf))
        assert n == g[k + "seq_x"].shape[1] and len(seq_indices) == len(ev)
        assert np.array_equal(seq_x.numpy(), g[k + "seq_x"]) and np.array_equal(seq_o.numpy(), g[k + "seq_o"])
        if k + "seq_c" in g:
            assert np.array_equal(seq_c.numpy(), g[k + "seq_c"])
        else:
            assert np.array_equal(seq_c[256:].numpy(), g[k + "seq_c_types"]) and np.array_equal(seq_c[:, ::7].numpy(), g[k + "seq_c_cols"])
        assert [seq_indices[i] for i in range(len(ev))] == g[k + "seq_indices"].tolist()
        off = g[k + "slider_off"]
        assert len(sliders) == len(off) - 1 > 3
        for j, sl in enumerate(sliders):
            assert sl.seq_indices.tolist() == g[k + "slider_idx"][off[j]:off[j + 1]].tolist()
            assert (sl.end_index, sl.curve_type, sl.length) == (g[k + "slider_end"][j], curves[g[k + "slider_curve"][j]], g[k + "slider_length"][j])
        pos = torch.from_numpy(np.random.default_rng(seed).uniform(0, 512, (2, n)).astype(np.float32))
        placed = dp.events_with_pos(ev, pos, seq_indices)
        names = g[k + "placed_names"].tolist()
        assert [(e.type.name, e.value) for e in placed] == [(names[t], v) for t, v in zip(g[k + "placed_type"], g[k + "placed_value"])]
        assert all(isinstance(e, Event) for e in placed) and not any(e.type == EventType.DISTANCE for e in placed)
    # the documented corner cases
    empty = dp.events_to_sequence([], None, 1.4)
    assert empty[0].shape == (2, 0) and empty[1].shape == (1, 0) and empty[2].shape == (1, 0) and empty[3:] == (0, {}, [])
    assert dp.events_with_pos([], torch.zeros(2, 0), {}) == []
    with pytest.raises(IndexError):                       # attribute events no type token claims and no record to join
        dp.group_events([Event(EventType.TIME_SHIFT, 5), Event(EventType.DISTANCE, 3)])
    zero_x = dp.events_to_sequence([Event(EventType.TIME_SHIFT, 10), Event(EventType.POS_X, 0), Event(EventType.POS_Y, 100),
                                    Event(EventType.CIRCLE)], None, 1.4)
    assert zero_x[0][:, 0].tolist() == [0.0, 0.0]          # a coordinate of 0 = "no position": the playfield centre
    assert not dp.events_to_sequence(ev, None, 1.4)[5] and not dp.events_to_sequence(ev, synthetic_timing(1), 1.4, has_sv=False)[5]


def test_diffusion_tokenizer_and_class_vector():
    """`DiffusionTokenizer` (the class vocabulary restated from the tokenizer.pkl state) and `get_class_vector`: layout
    [styles | difficulties | mappers | descriptors | circle sizes], unknown = the family's last id, clipping at both ends,
    the state round trip, and a family that is absent from the state."""
    from mapperatorinator_amd.diffusion_pipeline import DiffusionGenerationConfig, DiffusionTokenizer, get_class_vector
    from mh_testing import synthetic_diffusion_tokenizer_state
    st = synthetic_diffusion_tokenizer_state(8)
    tok = DiffusionTokenizer(st)
    n = [st["num_classes"], st["num_diff_classes"], st["num_mapper_classes"], st["num_descriptor_classes"], st["num_cs_classes"]]
    base = np.cumsum([0] + n)
    assert tok.num_tokens == base[5]
    assert [tok.style_unk, tok.diff_unk, tok.mapper_unk, tok.descriptor_unk, tok.cs_unk] == (base[1:] - 1).tolist()
    assert tok.encode_diff(-3.0) == base[1] and tok.encode_diff(1e9) == base[2] - 2
    assert tok.encode_cs(-1.0) == base[4] and tok.encode_cs(99.0) == base[5] - 2
    known = next(iter(st["beatmap_idx"]))
    assert tok.encode_style(known) == st["beatmap_idx"][known] and tok.encode_style(-5) == tok.style_unk
    assert tok.encode_descriptor_name("d1") == base[3] + 1 and tok.encode_descriptor_name("nope") == base[4]
    assert DiffusionTokenizer(tok.state_dict()).state_dict() == tok.state_dict()
    v = get_class_vector(tok, DiffusionGenerationConfig(difficulty=5.3, circle_size=4.2, descriptors=["d1", "nope", "d0"]))
    on = set(torch.nonzero(v)[:, 0].tolist())
    assert on == {tok.style_unk, tok.encode_diff(5.3), tok.mapper_unk, base[3], base[3] + 1, tok.encode_cs(4.2)}
    v = get_class_vector(tok, DiffusionGenerationConfig(descriptors=["nope"]))
    assert set(torch.nonzero(v)[:, 0].tolist()) == {tok.style_unk, tok.diff_unk, tok.mapper_unk, tok.descriptor_unk, tok.cs_unk}
    lean = DiffusionTokenizer(synthetic_diffusion_tokenizer_state(3))      # no difficulty / circle-size family
    assert lean.num_diff_classes == lean.num_cs_classes == 0
    assert get_class_vector(lean, DiffusionGenerationConfig(difficulty=4.0, circle_size=4.0)).sum() == 3


def test_beam_sample_warpers_equal_hf():
    """beam-sample: the top-k / top-p tail of `beam.BeamProcessors` against HF's own TopKLogitsWarper / TopPLogitsWarper
    (transformers is third-party and installed) with `min_tokens_to_keep = #eos + 1`, on log-probabilities that already carry
    -inf entries (the time-shift masks), ties included."""
    from transformers.generation.logits_process import TopKLogitsWarper, TopPLogitsWarper
    from mapperatorinator_amd.beam import BeamProcessors
    from mapperatorinator_amd.server import Sampling
    rng = torch.Generator().manual_seed(3)
    for top_k, top_p, keep in ((0, 0.9, 2), (12, 1.0, 4), (5, 0.6, 9), (40, 0.95, 3), (0, 1.0, 2), (300, 0.2, 1)):
        sp = Sampling()
        sp.do_sample, sp.top_k, sp.top_p, sp.temperature = 1, top_k, top_p, 1.0
        scores = torch.log_softmax(torch.randn(6, 200, generator=rng) * 3, -1)
        scores[:, 17:40] = float("-inf")
        scores[2, 100:110] = scores[2, 100]                                   # ties
        want = scores.clone()
        if top_k:
            want = TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=keep)(None, want)
        if top_p < 1.0:
            want = TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=keep)(None, want)
        got = BeamProcessors(sp, "cpu", min_tokens_to_keep=keep)(torch.zeros(6, 3, dtype=torch.long), scores)
        assert torch.equal(got, want), (top_k, top_p, keep)
        assert (torch.isfinite(got).sum(-1) >= min(keep, 177)).all()
    sp.do_sample = 0
    assert torch.equal(BeamProcessors(sp, "cpu")(torch.zeros(6, 3, dtype=torch.long), scores), scores)


def test_diffusion_generate_host_flow_on_a_stand_in_denoiser():
    """`DiffusionPipelineHIP.generate(events, config, timing)` without a GPU: the denoiser stage replaced by a stand-in that records
    what it is handed.  The stage must receive the tensors / sliders of `events_to_sequence` under the pipeline's stream-format flags,
    the class vector of the config and the null-class vector the reference builds (difficulty + circle size kept, descriptors =
    negative_descriptors, diffusion_pipeline.py:151-155), and its positions must come back as POS_X / POS_Y events; an event stream
    without hit objects is returned untouched and never reaches the stage."""
    import types
    from mapperatorinator_amd import diffusion_pipeline as dp
    from mh_testing import synthetic_diffusion_tokenizer_state, synthetic_event_stream, synthetic_timing
    tok = dp.DiffusionTokenizer(synthetic_diffusion_tokenizer_state(8))
    pipe = dp.DiffusionPipelineHIP(types.SimpleNamespace(device="cpu"), timesteps=[2] + [0] * 9, tokenizer=tok, types_first=True, has_sv=True)
    seen = {}

    def stage(seq_x, seq_o, seq_c, cv, ucv, noise_source=None, sliders=None, **kw):
        seen.update(seq_x=seq_x, seq_o=seq_o, seq_c=seq_c, cv=cv, ucv=ucv, sliders=sliders)
        T = seq_x.shape[1]
        return torch.stack([torch.arange(T, dtype=torch.float32) * 1.5 + 0.5, 300.0 - torch.arange(T, dtype=torch.float32)])[None]

    pipe.generate_positions = stage
    events, timing = synthetic_event_stream(30, 4, types_first=True), synthetic_timing(4)
    cfg = dp.DiffusionGenerationConfig(difficulty=6.1, circle_size=3.5, slider_multiplier=1.9, descriptors=["d0"], negative_descriptors=["d3", "nope"])
    out = pipe.generate(events, cfg, timing)
    want = dp.events_to_sequence(events, timing, 1.9, types_first=True, has_sv=True)
    assert torch.equal(seen["seq_x"], want[0]) and torch.equal(seen["seq_o"], want[1]) and torch.equal(seen["seq_c"], want[2])
    assert [(s.seq_indices.tolist(), s.end_index, s.curve_type, s.length) for s in seen["sliders"]] == \
           [(s.seq_indices.tolist(), s.end_index, s.curve_type, s.length) for s in want[5]] and len(want[5]) > 2
    assert torch.equal(seen["cv"], dp.get_class_vector(tok, cfg))
    null = dp.DiffusionGenerationConfig(difficulty=6.1, circle_size=3.5, descriptors=["d3", "nope"])
    assert torch.equal(seen["ucv"], dp.get_class_vector(tok, null)) and not torch.equal(seen["ucv"], seen["cv"])
    pos = stage(*want[:3], None, None)[0]
    assert [(e.type.name, e.value) for e in out] == [(e.type.name, e.value) for e in dp.events_with_pos(events, pos, want[4])]
    assert sum(e.type == EventType.POS_X for e in out) == sum(e.type == EventType.DISTANCE for e in events) > 20
    seen.clear()
    beats = [Event(EventType.TIME_SHIFT, 100), Event(EventType.BEAT), Event(EventType.TIME_SHIFT, 600), Event(EventType.MEASURE)]
    assert pipe.generate(beats, cfg, timing) is beats and not seen
    with pytest.raises(ValueError, match="tokenizer"):
        dp.DiffusionPipelineHIP(types.SimpleNamespace(device="cpu"), timesteps=[2] + [0] * 9).generate(events, cfg, timing)


def test_import_leaves_the_environment_alone_and_configure_runtime_is_explicit():
    """Importing the package must not touch os.environ (round-5 verdict / advice: a library does not mutate its host's HIP
    runtime); `configure_runtime()` is the explicit opt-in, an existing value wins, and a call after HIP initialised warns."""
    import subprocess
    import sys
    var = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
    env = {k: v for k, v in os.environ.items() if k != var}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")

    def run(code):
        return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip()

    assert run(f"import os, mapperatorinator_amd, mapperatorinator_amd.modeling; print(os.environ.get('{var}'))") == "None"
    assert run(f"import os, mapperatorinator_amd as m; r = m.configure_runtime(); print(os.environ['{var}'], r['applied'], r['effective'])") == "0 True True"
    env[var] = "1"
    assert run(f"import os, mapperatorinator_amd as m; r = m.configure_runtime(); print(os.environ['{var}'], r['applied'])") == "1 False"
    del env[var]
    code = ("import sys, types, warnings, mapperatorinator_amd as m\n"
            "import torch\n"
            "torch.cuda.is_initialized = lambda: True\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always')\n"
            "    r = m.configure_runtime()\n"
            "print(r['effective'], len(w), w[0].category.__name__)")
    assert run(code) == "False 1 RuntimeWarning"


def test_graft_entry_build():
    """The driver's build check: `make` (incremental) + dlopen + ABI / layout verification, no GPU needed."""
    import __graft_entry__ as g
    g.build()


def test_abi_header_is_plain_c():
    """include/mapperhip.h is the contract a maintainer binds: it must compile as C99 and as C++ on its own."""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "mapperhip.h")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.run(["gcc", "-x", "c", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", hdr], check=True)
    subprocess.run(["g++", "-x", "c++", "-std=c++17", "-fsyntax-only", hdr], check=True)


class _StandInEngine:
    """What SequentialWindowScheduler needs of an engine, on the CPU: per-window "cross K/V" = a fingerprint of the
    window's audio, and a decode that is a deterministic function of (fingerprint, prompt without its left padding,
    position) -- batch-invariant like the real engine, honouring EOS / max_length / pad-after-EOS and the guidance row
    layout [negative rows | prompt rows]."""

    def __init__(self, vocab_out, eos_every):
        import contextlib
        import types
        self.device = torch.device("cpu")
        self.packed = types.SimpleNamespace(vocab_out=vocab_out)
        self.calls = []
        self.row_bias_seen = []
        self._ctx = contextlib.nullcontext
        self.eos_every = eos_every

    def _enter(self): pass
    def _leave(self): pass
    def synchronize(self): pass
    def on_stream(self): return self._ctx()
    def mel(self, audio): return audio
    def encode_mel(self, mel, row_bias=None):
        self.row_bias_seen.append(None if row_bias is None else row_bias.clone())
        return mel

    def cross_kv(self, enc):                                   # [layers=1][k|v=2][B][H=1][L=1][64]: the fingerprint
        fp = (enc.abs().sum(-1) * 1000).round().long() % 997
        return fp.view(1, 1, -1, 1, 1, 1).expand(1, 2, -1, 1, 1, 64).contiguous().float()

    def decode(self, kv, prompt, prompt_mask, eos_table, sampling, forced=None, dump_logits=False, poll_every=16, kv_fp8=None):
        B, P = prompt.shape
        cfg = sampling.cfg_scale > 1.0
        G = B // 2 if cfg else B
        assert kv.shape[2] == G
        self.calls.append(dict(B=B, P=P, cfg=cfg, max_length=sampling.max_length, masked=prompt_mask is not None))
        maxlen = sampling.max_length
        tokens = torch.full((B, maxlen), int(sampling.pad_id), dtype=torch.int32)
        tokens[:, :P] = prompt
        n_cols = P
        for r in range(G):
            row = r + (G if cfg else 0)
            own = prompt[row][prompt_mask[row].bool()] if prompt_mask is not None else prompt[row]
            seed = int(kv[0, 0, r, 0, 0, 0]) * 31 + int(own.sum()) * 7 + own.numel()
            if cfg:
                seed += int(prompt[r][prompt_mask[r].bool()].sum() if prompt_mask is not None else prompt[r].sum())
            done = False
            for c in range(P, maxlen):
                if done:
                    break
                t = 3 + (seed + 13 * (c - P)) % (self.packed.vocab_out - 3)
                if eos_table[t] or c + 1 >= maxlen:
                    done = True
                tokens[row, c] = t
                n_cols = max(n_cols, c + 1)
        return tokens, torch.tensor([n_cols], dtype=torch.int32), None


@pytest.mark.parametrize("cfg_scale", [1.0, 2.0])
def test_window_scheduler_waves_equal_the_sequential_loop_on_a_stand_in_engine(cfg_scale):
    """SURVEY 8f rank 1 without a GPU: songs of 3 / 1 / 4 dependent windows through SequentialWindowScheduler (wave w =
    window w of every song that has one, grouped by generate kwargs, ragged prompts left-padded and masked, guidance
    rows doubled and halving the wave) must hand every `on_result` the row a batch-1 call for that window returns --
    prompt included, cut after its own first EOS-set id -- in window order per song."""
    import types
    from mapperatorinator_amd import Tokenizer
    from mapperatorinator_amd.scheduler import SequentialWindowScheduler, SongJob
    from mapperatorinator_amd.server import build_sampling
    tok = Tokenizer.benchmark_vocab(src_seq_len=251)
    tgt = 40
    eng = _StandInEngine(tok.vocab_size_out, eos_every=5)
    model = types.SimpleNamespace(engine=eng, config=types.SimpleNamespace(max_target_positions=tgt))
    n_windows = [3, 1, 4]
    g = torch.Generator().manual_seed(5)
    songs = [torch.randn(n, 64, generator=g) for n in n_windows]

    def kwargs_for(w, n):
        return dict(max_length=tgt, do_sample=False, cfg_scale=cfg_scale, lookback_time=400 if w != 0 else 0,
                    lookahead_time=3000 if w != n - 1 else 0)

    def prompt_from(prev):
        carry = [] if prev is None else [t for t in prev.tolist() if t > 2][-3:]
        return torch.tensor([[tok.sos_id] + carry])

    neg = torch.tensor([[tok.sos_id]])

    def batch1(k, w, prompt):
        gk = dict(kwargs_for(w, n_windows[k]), conditional_temperature_per_row=True)
        sp, eos = build_sampling(tok, gk, tgt)
        table = torch.zeros(tok.vocab_size_out, dtype=torch.uint8)
        table[[e for e in eos if 0 <= e < tok.vocab_size_out]] = 1
        kv = eng.cross_kv(songs[k][w:w + 1])
        p = prompt
        if cfg_scale > 1:
            negp = prompt.clone()
            negp[:, :1] = neg
            p = torch.cat([negp, prompt], 0)
        tokens, n_out, _ = eng.decode(kv, p.int(), torch.ones_like(p, dtype=torch.uint8), table, sp)
        row = tokens[-1, :int(n_out)].long()
        body = row[prompt.shape[1]:]
        hit = torch.isin(body, torch.tensor(sorted(eos))).nonzero()
        return row[:prompt.shape[1] + int(hit[0]) + 1] if hit.numel() else row

    want = []
    for k, n in enumerate(n_windows):
        prev, rows = None, []
        for w in range(n):
            prompt = prompt_from(prev)
            row = batch1(k, w, prompt)
            prev = row[prompt.shape[1]:]
            rows.append(row)
        want.append(rows)
    eng.calls.clear()

    got = [[None] * n for n in n_windows]
    state = [None] * len(n_windows)
    order = []

    def make_job(k, n):
        def prompt_fn(w):
            ask = dict(decoder_input_ids=prompt_from(state[k]), generate_kwargs=kwargs_for(w, n))
            if cfg_scale > 1:
                ask["negative_prompt"] = neg
            return ask

        def on_result(w, row, st):
            p = prompt_from(state[k]).shape[1]
            got[k][w] = row
            state[k] = row[p:]
            order.append((k, w))
            assert st["generated_tokens"] == int((row[p:] != 0).sum())
        return SongJob(frames=songs[k], prompt_fn=prompt_fn, on_result=on_result)

    sched = SequentialWindowScheduler(model, tok, encode_batch=4, decode_batch=4)
    stats = sched.run([make_job(k, n) for k, n in enumerate(n_windows)])
    assert stats["windows"] == sum(n_windows) and stats["encode_calls"] == 2          # 8 windows, 4 per encode batch
    for k, n in enumerate(n_windows):
        assert [w for kk, w in order if kk == k] == list(range(n))                     # per song in window order
        for w in range(n):
            assert torch.equal(got[k][w], want[k][w]), (k, w, got[k][w].tolist(), want[k][w].tolist())
    assert stats["decode_calls"] < sum(n_windows)                                      # songs were interleaved
    for c in eng.calls:
        assert c["B"] <= 4 and c["cfg"] == (cfg_scale > 1) and (c["B"] % 2 == 0 or not c["cfg"])


def test_window_scheduler_hands_per_window_conditioning_to_the_encoder():
    """Models with conditioning embedders: `SongJob.conditioning_fn(window)` -> one row of the encoder's row bias per window,
    in the order the windows are encoded (song_position differs per window: processor.py:341-345)."""
    import types
    from mapperatorinator_amd import Tokenizer
    from mapperatorinator_amd.conditioning import ConditioningEmbedders
    from mapperatorinator_amd.scheduler import SequentialWindowScheduler, SongJob
    from mapperatorinator_amd.t5_engine import T5_PRESETS
    from mh_testing import add_random_conditioning, random_t5_state_dict
    tok = Tokenizer.benchmark_vocab(src_seq_len=251)
    sd = random_t5_state_dict(T5_PRESETS["tiny"], tok.vocab_size_in, tok.vocab_size_out, seed=3)
    add_random_conditioning(sd, 128, 388, 16, 11, seed=1)
    cond = ConditioningEmbedders(sd, 388)
    eng = _StandInEngine(tok.vocab_size_out, eos_every=5)
    model = types.SimpleNamespace(engine=eng, config=types.SimpleNamespace(max_target_positions=24), cond=cond, dtype=torch.float32)
    g = torch.Generator().manual_seed(1)
    n_windows = [2, 3]
    songs = [torch.randn(n, 64, generator=g) for n in n_windows]

    def cond_for(k, w):
        return dict(difficulty=3.0 + k, mapper_idx=-1 if k == 0 else 4, song_position=[w / 3, (w + 1) / 3])
    jobs = [SongJob(frames=songs[k], prompt_fn=lambda w: dict(decoder_input_ids=torch.tensor([[tok.sos_id]])),
                    on_result=lambda *a: None, generate_kwargs=dict(max_length=24, do_sample=False),
                    conditioning_fn=(lambda w, k=k: cond_for(k, w))) for k in range(2)]
    SequentialWindowScheduler(model, tok, encode_batch=4, decode_batch=4).run(jobs)
    seen = torch.cat([r for r in eng.row_bias_seen], 0)
    flat = [cond_for(k, w) for k in range(2) for w in range(n_windows[k])]
    want = cond.row_bias(cond.vectors(5, difficulty=torch.tensor([f["difficulty"] for f in flat]),
                                      mapper_idx=torch.tensor([f["mapper_idx"] for f in flat]),
                                      song_position=torch.tensor([f["song_position"] for f in flat])), torch.float32)
    assert seen.shape == want.shape and torch.allclose(seen, want)
    with pytest.raises(ValueError):
        SequentialWindowScheduler(model, tok).run([SongJob(frames=songs[0], prompt_fn=jobs[0].prompt_fn, on_result=lambda *a: None,
                                                             generate_kwargs=dict(max_length=24, do_sample=False))])


def test_band_mask_with_padding_round_trips_through_the_mask_analysis():
    """BandMask(open_from=) builds exactly the mask the reference pads (diffusion_pipeline.py:146-148 + :190), and the
    analysis of a foreign (T, T) tensor recovers (band, open_from) from it or refuses."""
    from mapperatorinator_amd.dit import BandMask, DiTHIP
    T, real, band = 96, 70, 16
    idx = torch.arange(real)
    inner = ~((idx[:, None] >= idx[None, :] - band) & (idx[:, None] < idx[None, :] + band))     # the reference's band, True = masked
    ref = torch.nn.functional.pad(inner, (0, T - real, 0, T - real), value=False)
    ours = BandMask(T, band, open_from=real)
    assert torch.equal(ours.to_tensor(), ref) and (ours.band, ours.open_from) == (band, real)
    analyse = DiTHIP.band_from_mask
    assert analyse(None, ref, T) == (band, real)
    assert analyse(None, inner, real) == (band, 0)
    assert analyse(None, torch.zeros(T, T, dtype=torch.bool), T) == (0, 0)
    assert analyse(None, BandMask(T, 200), T) == (0, 0)               # a band wider than the window masks nothing
    broken = ref.clone()
    broken[3, 60] = False
    with pytest.raises(NotImplementedError):
        analyse(None, broken, T)


def test_slider_set_packing_filters_windows_and_refuses_what_the_kernel_cannot_do():
    """pack_slider_set: the host side of MhSliderSet (reference filter diffusion_pipeline.py:210-212)."""
    from mapperatorinator_amd.diffusion_pipeline import DiffusionSlider
    from mapperatorinator_amd.dit import MAX_BEZIER_SPAN, pack_slider_set
    S = DiffusionSlider
    inside = S(np.array([12, 13, 13, 14]), 15, "Bezier", 120.0)         # red anchor = the same point twice
    crosses = S(np.array([28, 29, 30]), 31, "PerfectCurve", 50.0)       # end outside [10, 30)
    before = S(np.array([3, 4]), 5, "Catmull", 10.0)
    odd = S(np.array([20, 21]), 22, None, 33.0)                         # unknown curve type -> Bezier (calculate_subpath's else)
    active, chunk_off, types, cp_off, cp_idx, end_idx, length = pack_slider_set([[before, inside, crosses, odd], []], 10, 30)
    assert active == [1, 0] and chunk_off == [0, 2, 2]
    assert types == [3, 3] and cp_off == [0, 4, 6] and cp_idx == [2, 3, 3, 4, 10, 11] and end_idx == [5, 12] and length == [120.0, 33.0]
    # a song whose sliders all fall outside the window still counts as "has sliders" (the pixel round trip, :208)
    assert pack_slider_set([[before]], 10, 30)[:2] == ([1], [0, 0])
    with pytest.raises(NotImplementedError, match="sharing"):
        pack_slider_set([[inside, S(np.array([15, 16]), 17, "Bezier", 5.0)]], 10, 30)      # starts on the other's end point
    with pytest.raises(NotImplementedError, match="curve span"):
        pack_slider_set([[S(np.arange(40, 40 + MAX_BEZIER_SPAN + 1), 90, "Bezier", 5.0)]], 0, 100)
    long_but_split = S(np.concatenate([np.arange(40, 60), [59], np.arange(60, 80)]), 90, "Bezier", 5.0)   # 41 points, spans of 20 + 21
    assert pack_slider_set([[long_but_split]], 0, 100)[2] == [3]


def test_request_batcher_closes_the_requests_of_a_failed_batch():
    """a generate call that raises (a refused kwarg combination, OOM) must not strand its requests -- the reference
    answers every request of the batch with RETRY_SIGNAL (osuT5/osuT5/inference/server.py:418-424).  Here: the failed batch's
    requests are closed with the error (rows still queued are dropped with them), requests that were not in the batch are
    answered normally afterwards, and the error reaches the driver."""
    import types

    import pytest
    import torch

    from mapperatorinator_amd import server as our_server
    calls = []

    def flaky_generate(model, tokenizer, model_kwargs, generate_kwargs):
        n = model_kwargs["decoder_input_ids"].shape[0]
        calls.append(n)
        if len(calls) == 1:
            raise NotImplementedError("refused combination")
        return torch.full((n, 6), 7, dtype=torch.int64), dict(generated_tokens_per_sample=[4] * n, elapsed_seconds=0.25)

    b = our_server.RequestBatcher(None, types.SimpleNamespace(pad_id=0), max_batch_size=4, generate_fn=flaky_generate)
    ids = torch.ones(6, 2, dtype=torch.long)
    big = b.submit(dict(inputs=torch.zeros(6, 10), decoder_input_ids=ids), dict(num_beams=1))      # 4 rows in batch 1, 2 staged
    small = b.submit(dict(inputs=torch.zeros(1, 10), decoder_input_ids=ids[:1]), dict(num_beams=1))
    with pytest.raises(NotImplementedError):
        b.step()
    assert big["done"] and isinstance(big["error"], NotImplementedError) and big["result"] is None
    assert not small["done"] and small["error"] is None
    b.drain()
    assert small["done"] and small["error"] is None and small["result"]["output"].shape == (1, 6)
    assert big["result"] is None and not b.pending


def test_fresh_seed_call_index_is_independent_of_process_history():
    """an explicit (seed, call_index) names a stream by itself; the process-wide count only serves callers that do
    not keep their own, is bounded and locked."""
    from mapperatorinator_amd import server as s
    s.reset_seed_calls()
    a0, a1 = s.fresh_seed(11), s.fresh_seed(11)
    assert a0 != a1
    assert s.fresh_seed(11, call_index=0) == a0 and s.fresh_seed(11, call_index=1) == a1      # no global state touched
    assert s.fresh_seed(11) not in (a0, a1)
    for k in range(s._SEED_CALLS_MAX + 50):
        s.fresh_seed(1000 + k)
    assert len(s._SEED_CALLS) <= s._SEED_CALLS_MAX
    s.reset_seed_calls()
    assert s.fresh_seed(11) == a0


def test_whisper_packer_keeps_conditioning_channels_of_conv1():
    """ADVICE r3 asked that a conv1 with n_mels + conditioning input channels be refused rather than cropped to n_mels; round 6
    builds it (configs/model/whisper_small_v2.yaml: the V30 / V31 wiring): the packer keeps EVERY input channel (tap-major, K
    padded), reports `cond_channels`, and a state dict whose conv1 is wider than n_mels without conditioning embedders is
    refused where the embedders are read."""
    import pytest
    import torch

    from mapperatorinator_amd.conditioning import ConditioningEmbedders
    from mh_testing import add_random_cond_embedders, random_varwhisper_state_dict, random_whisper_family_state_dict
    from mapperatorinator_amd.whisper_engine import VARWHISPER_PRESETS, PackedVarWhisper, fuse_split_projections, whisper_kind
    d = VARWHISPER_PRESETS["test"]
    sd = random_varwhisper_state_dict(d.d_model, d.n_heads, d.n_enc_layers, d.n_dec_layers, d.d_ff, 200, 180, seed=0)
    k = "transformer.model.encoder.conv1.weight"
    n_mels = sd[k].shape[1]
    sd[k] = torch.cat([sd[k], torch.full((sd[k].shape[0], 16, 3), 0.5)], 1)          # + 16 conditioning channels
    p = PackedVarWhisper(sd, d, 200, 180, n_mels, 64, 32, torch.float32, "cpu")
    assert p.kind == "var" and p.cond_channels == 16 and p.n_mels_pad == 160 and p.cfg.n_mels == n_mels + 16 and p.cfg.arch == 1
    w1 = [t for t in p._keep if t.shape == (d.d_model, 3 * 160)][0]
    assert torch.equal(w1[:, n_mels:n_mels + 16], torch.full((d.d_model, 16), 0.5)) and float(w1[:, n_mels + 16:160].abs().sum()) == 0
    with pytest.raises(ValueError, match="conditioning"):
        ConditioningEmbedders(sd, n_mels)
    add_random_cond_embedders(sd, cond_dim=5)                # 3 x 5 = 15 != 16: the widths must agree when vectors are built
    ce = ConditioningEmbedders(sd, n_mels)
    assert ce.as_channels and ce.cond_size == 16
    with pytest.raises(ValueError, match="columns"):
        ce.vectors(2, difficulty=[1.0, 2.0], mapper_idx=[0, 1], song_position=[[0, 0.1], [0.5, 0.6]])
    # the split-projection families: names fused, k bias zero, kinds told apart, arch 2 config for stock Whisper
    for kind, n_m in (("rope", 80), ("hf", 388)):
        sd2 = random_whisper_family_state_dict(kind, d.d_model, d.n_heads, d.n_enc_layers, d.n_dec_layers, d.d_ff, 200, 180, n_m,
                                               src_positions=32, tgt_positions=32, cond_size=0, seed=1)
        assert whisper_kind(sd2) == kind
        f = fuse_split_projections(sd2)
        b0 = "transformer.model.decoder.layers.0."
        assert f[b0 + "self_attn.Wqkv.weight"].shape == (3 * d.d_model, d.d_model) and float(f[b0 + "self_attn.Wqkv.bias"][d.d_model:2 * d.d_model].abs().sum()) == 0
        assert torch.equal(f[b0 + "cross_attn.Wkv.weight"][:d.d_model], sd2[b0 + "encoder_attn.k_proj.weight"])
        assert b0 + "cross_attn_layer_norm.weight" in f and not [x for x in f if "q_proj" in x or "encoder_attn" in x]
        p2 = PackedVarWhisper(sd2, d, 200, 180, n_m, 64, 32, torch.float32, "cpu")
        assert p2.kind == kind and p2.cfg.arch == (2 if kind == "hf" else 1) and p2.cond_channels == 0
        assert bool(p2.w.dec_pos) == (kind == "hf") and bool(p2.w.dec_ln1_b[0]) == (kind == "hf")


def test_kernels_with_asm_issued_loads_have_no_scratch_and_no_spills():
    """gemm_glds3 / gemm_s3g (and every later kernel on the same discipline) read LDS through inline-asm
    `ds_read`s behind hand-counted `s_waitcnt`s; hipcc treats an asm output as valid at once, so a spill of such a register
    stores stale bytes.  The built code objects must report zero scratch and zero spilled registers for them."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ckr", os.path.join(root, "tools", "check_kernel_resources.py"))
    ckr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ckr)
    lib = os.path.join(root, "mapperatorinator_amd", "lib", "libmapperhip.so")
    rows = [(k, elf) for elf in ckr.code_objects(lib) for k in ckr.kernels_of(elf)]
    watched = [(k, elf) for k, elf in rows if any(w in k[".symbol"] for w in ckr.ASM_LOAD_KERNELS)]
    assert len(rows) > 100 and len(watched) >= 20
    for k, elf in watched:
        dirty = (int(k.get(".private_segment_fixed_size", 0)) or int(k.get(".vgpr_spill_count", 0)) or int(k.get(".sgpr_spill_count", 0)))
        if dirty:   # allowed only where the disassembly shows every scratch access behind the kernel's last MFMA (an epilogue
            # spill: nothing asm-issued is in flight there) -- the 256 x 256 bf16 tile with its 128 AGPR accumulators
            assert "gemm_glds4_kernel" in k[".symbol"], k[".symbol"]
            assert ckr.scratch_only_behind_last_mfma(elf, k[".symbol"].removesuffix(".kd")), k[".symbol"]
    assert ckr.main([lib]) == 0


def test_mx8_host_packer_oracle_and_torch_restatement_agree():
    """three statements of the MX-fp8 quantisation rule (numpy oracle, its torch form used inside the model oracles, the
    product's weight packer) produce the same bytes / values, including zero blocks, outliers and the 448 boundary"""
    import numpy as np
    import torch

    from mapperatorinator_amd import mx8 as host
    from oracle import mx8 as omx
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((41, 640)) * np.exp(rng.standard_normal((41, 640)) * 2)).astype(np.float32)
    x[3, :96] = 0.0
    x[7, 33] = 1.0e6
    x[9, :32] = np.linspace(449.0, 511.0, 32, dtype=np.float32) * 2.0 ** -3          # amax * 2^-e in (448, 512): the exponent is raised
    q, s = omx.quantize_mx8(x)
    qh, sh = host.quantize_mx8(torch.from_numpy(x))
    assert np.array_equal(q, qh.numpy()) and np.array_equal(s, sh.numpy())
    assert s.shape[1] == host.scale_row_bytes(640) == 32
    d = omx.dequantize_mx8(q, s)
    assert np.array_equal(d, host.dequantize_mx8(qh, sh).numpy().astype(np.float64))
    assert np.array_equal(d, omx.fake_quant_torch(torch.from_numpy(x)).numpy().astype(np.float64))
    assert np.all(np.abs(d[9, :32] - x[9, :32]) <= np.abs(x[9, :32]).max() * 2.0 ** -4), "nothing may be clipped: round-to-nearest error only"
    # e4m3 grid facts the rule relies on
    t = omx.e4m3_decode_table()
    assert t[0x7e] == 448.0 and np.isnan(t[0x7f]) and t[0x01] == 2.0 ** -9 and t[0x08] == 2.0 ** -6
