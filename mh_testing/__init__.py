"""TEST SUPPORT (not part of the product package; round 6 moved it out of mapperatorinator_amd/): deterministic synthetic inputs /
random-init weights shared by tests, bench.py, tools/ and the golden generator (oracle/make_golden.py).  numpy's PCG64 stream is platform-stable, so the GPU box
regenerates bit-identical weights from a seed instead of shipping 100s of MB of fixtures.

Weight scales follow HF `T5PreTrainedModel._init_weights` (factor 1.0) and the reference wrapper's
own initialisers (modeling_mapperatorinator.py:123-128: nn.Linear default, Embedding std=0.02);
`lm_head_gain` optionally widens the logit gaps so that greedy ties are well above fp noise
(SURVEY.md 7 "hard parts").
"""
from __future__ import annotations

import dataclasses
import datetime
from typing import Optional

import numpy as np
import torch

from mapperatorinator_amd.t5_engine import T5Dims


def _normal(rng, shape, std):
    return torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32))


def random_t5_state_dict(dims: T5Dims, vocab_in: int, vocab_out: int, n_mels: int = 388, seed: int = 0,
                         lm_head_gain: float = 1.0, ln_jitter: float = 0.1, gains: dict | None = None) -> dict:
    """`gains`: {substring of a parameter name: factor} applied after the draw (the draw itself is unchanged, so
    fixtures made without gains stay bit-identical).  DIVERSE_GAINS makes greedy decoding of random weights depend
    on the audio and the history instead of falling into a short cycle."""
    rng = np.random.default_rng(seed)
    d, dff, inner, H = dims.d_model, dims.d_ff, dims.inner, dims.n_heads
    sd = {}
    sd["encoder_embedder.weight"] = _normal(rng, (d, n_mels), 1.0 / np.sqrt(n_mels))
    sd["encoder_embedder.bias"] = _normal(rng, (d,), 0.02)
    sd["decoder_embedder.weight"] = _normal(rng, (vocab_in, d), 1.0)
    sd["transformer.lm_head.weight"] = _normal(rng, (vocab_out, d), lm_head_gain * d ** -0.5)

    def attn(prefix, with_bias_table):
        sd[prefix + "q.weight"] = _normal(rng, (inner, d), (d * dims.d_kv) ** -0.5)
        sd[prefix + "k.weight"] = _normal(rng, (inner, d), d ** -0.5)
        sd[prefix + "v.weight"] = _normal(rng, (inner, d), d ** -0.5)
        sd[prefix + "o.weight"] = _normal(rng, (d, inner), inner ** -0.5)
        if with_bias_table:
            sd[prefix + "relative_attention_bias.weight"] = _normal(rng, (dims.n_buckets, H), 0.5)

    def ffn(prefix):
        sd[prefix + "wi_0.weight"] = _normal(rng, (dff, d), d ** -0.5)
        sd[prefix + "wi_1.weight"] = _normal(rng, (dff, d), d ** -0.5)
        sd[prefix + "wo.weight"] = _normal(rng, (d, dff), dff ** -0.5)

    def ln(name):
        sd[name] = 1.0 + _normal(rng, (d,), ln_jitter)

    for l in range(dims.n_enc_layers):
        b = f"transformer.encoder.block.{l}."
        attn(b + "layer.0.SelfAttention.", l == 0)
        ln(b + "layer.0.layer_norm.weight")
        ffn(b + "layer.1.DenseReluDense.")
        ln(b + "layer.1.layer_norm.weight")
    ln("transformer.encoder.final_layer_norm.weight")
    for l in range(dims.n_dec_layers):
        b = f"transformer.decoder.block.{l}."
        attn(b + "layer.0.SelfAttention.", l == 0)
        ln(b + "layer.0.layer_norm.weight")
        attn(b + "layer.1.EncDecAttention.", False)
        ln(b + "layer.1.layer_norm.weight")
        ffn(b + "layer.2.DenseReluDense.")
        ln(b + "layer.2.layer_norm.weight")
    ln("transformer.decoder.final_layer_norm.weight")
    for pat, g in (gains or {}).items():
        for k in sd:
            if pat in k:
                sd[k] = sd[k] * g
    return sd


def random_varwhisper_state_dict(d_model: int, n_heads: int, n_enc: int, n_dec: int, ffn: int, vocab_in: int, vocab_out: int,
                                 n_mels: int = 128, seed: int = 0, head_gain: float = 1.0, attention_bias: bool = True,
                                 ln_jitter: float = 0.1, gains: dict | None = None) -> dict:
    """Reference-named parameters of `Mapperatorinator` over the VarWhisper backbone
    (custom_transformers/modeling_varwhisper.py; configs/model/varwhisper_*_v3.yaml: input_features, no encoder
    projection, wrapper-side decoder embedding, untied head).  Scales keep the residual stream O(1); `attention_bias`
    adds the Wqkv / Wq / Wkv / Wo biases of `config.attention_bias`."""
    rng = np.random.default_rng(7000 + seed)
    d = d_model
    sd = {}
    sd["decoder_embedder.weight"] = _normal(rng, (vocab_in, d), 1.0)
    sd["transformer.model.decoder.embed_tokens.weight"] = _normal(rng, (vocab_out, d), 0.02)      # (unused: embed_decoder_input)
    sd["transformer.proj_out.weight"] = _normal(rng, (vocab_out, d), head_gain * d ** -0.5)
    e = "transformer.model.encoder."
    sd[e + "conv1.weight"] = _normal(rng, (d, n_mels, 3), (3 * n_mels) ** -0.5)
    sd[e + "conv1.bias"] = _normal(rng, (d,), 0.05)
    sd[e + "conv2.weight"] = _normal(rng, (d, d, 3), 1.7 * (3 * d) ** -0.5)
    sd[e + "conv2.bias"] = _normal(rng, (d,), 0.05)

    def lin(name, n_out, n_in, bias, std=None):
        sd[name + ".weight"] = _normal(rng, (n_out, n_in), std if std is not None else n_in ** -0.5)
        if bias:
            sd[name + ".bias"] = _normal(rng, (n_out,), 0.05)

    def ln(name):
        sd[name] = 1.0 + _normal(rng, (d,), ln_jitter)

    def mlp(b):
        lin(b + "fc1", ffn, d, True)
        lin(b + "fc2", d, ffn, True)

    for l in range(n_enc):
        b = e + f"layers.{l}."
        lin(b + "self_attn.Wqkv", 3 * d, d, attention_bias, 1.5 * d ** -0.5)
        lin(b + "self_attn.Wo", d, d, attention_bias)
        ln(b + "self_attn_layer_norm.weight")
        mlp(b)
        ln(b + "final_layer_norm.weight")
    ln(e + "layer_norm.weight")
    dd = "transformer.model.decoder."
    for l in range(n_dec):
        b = dd + f"layers.{l}."
        lin(b + "self_attn.Wqkv", 3 * d, d, attention_bias, 1.5 * d ** -0.5)
        lin(b + "self_attn.Wo", d, d, attention_bias)
        ln(b + "self_attn_layer_norm.weight")
        lin(b + "cross_attn.Wq", d, d, attention_bias, 2.0 * d ** -0.5)
        lin(b + "cross_attn.Wkv", 2 * d, d, attention_bias, 1.5 * d ** -0.5)
        lin(b + "cross_attn.Wo", d, d, attention_bias, 2.0 * d ** -0.5)
        ln(b + "cross_attn_layer_norm.weight")
        mlp(b)
        ln(b + "final_layer_norm.weight")
    ln(dd + "layer_norm.weight")
    for pat, g in (gains or {}).items():
        for k in sd:
            if pat in k:
                sd[k] = sd[k] * g
    return sd


def random_whisper_family_state_dict(kind: str, d_model: int, n_heads: int, n_enc: int, n_dec: int, ffn: int, vocab_in: int,
                                      vocab_out: int, n_mels: int, src_positions: int = 0, tgt_positions: int = 0, cond_size: int = 0,
                                      seed: int = 0, head_gain: float = 1.0, ln_jitter: float = 0.1, gains: dict | None = None) -> dict:
    """Reference-named parameters of `Mapperatorinator` over
      kind "rope": RoPEWhisperForConditionalGeneration (custom_transformers/modeling_ropewhisper.py; configs/model/
                   whisper_small_v2.yaml: input_features, no encoder projection -> conv1 takes n_mels + cond_size channels),
      kind "hf":   transformers' WhisperForConditionalGeneration (configs/model/whisper_{base,small}.yaml: the wrapper's
                   encoder_embedder (n_mels + cond_size -> d) in front of conv1 (d channels), affine LayerNorms, the fixed
                   sinusoid encoder positions (`src_positions` rows) and LEARNED decoder positions (`tgt_positions` rows)).
    Both: separate q_proj (bias) / k_proj (no bias) / v_proj (bias) / out_proj (bias), fc1 / fc2 with bias, untied proj_out."""
    assert kind in ("rope", "hf")
    rng = np.random.default_rng(9000 + seed)
    d = d_model
    sd = {}
    sd["decoder_embedder.weight"] = _normal(rng, (vocab_in, d), 1.0)
    sd["transformer.model.decoder.embed_tokens.weight"] = _normal(rng, (vocab_out, d), 0.02)      # (unused: embed_decoder_input)
    sd["transformer.proj_out.weight"] = _normal(rng, (vocab_out, d), head_gain * d ** -0.5)
    e, dd = "transformer.model.encoder.", "transformer.model.decoder."
    c_in = n_mels + cond_size
    if kind == "hf":
        sd["encoder_embedder.weight"] = _normal(rng, (d, c_in), 1.0 / np.sqrt(n_mels))
        sd["encoder_embedder.bias"] = _normal(rng, (d,), 0.02)
        c_in = d
        # sinusoids(length, channels) of modeling_whisper.py (what HF's _init_weights copies into embed_positions)
        inc = np.log(10000.0) / (d // 2 - 1)
        t = np.arange(src_positions)[:, None] * np.exp(-inc * np.arange(d // 2))[None, :]
        sd[e + "embed_positions.weight"] = torch.from_numpy(np.concatenate([np.sin(t), np.cos(t)], 1).astype(np.float32))
        sd[dd + "embed_positions.weight"] = _normal(rng, (tgt_positions, d), 0.3)
    sd[e + "conv1.weight"] = _normal(rng, (d, c_in, 3), (3 * c_in) ** -0.5)
    sd[e + "conv1.bias"] = _normal(rng, (d,), 0.05)
    sd[e + "conv2.weight"] = _normal(rng, (d, d, 3), 1.7 * (3 * d) ** -0.5)
    sd[e + "conv2.bias"] = _normal(rng, (d,), 0.05)

    def lin(name, n_out, n_in, bias, std=None):
        sd[name + ".weight"] = _normal(rng, (n_out, n_in), std if std is not None else n_in ** -0.5)
        if bias:
            sd[name + ".bias"] = _normal(rng, (n_out,), 0.05)

    def ln(name):
        sd[name + ".weight"] = 1.0 + _normal(rng, (d,), ln_jitter)
        if kind == "hf":
            sd[name + ".bias"] = _normal(rng, (d,), 0.1)

    def attn(b, qstd, ostd=None):
        lin(b + "q_proj", d, d, True, qstd)
        lin(b + "k_proj", d, d, False, 1.5 * d ** -0.5)
        lin(b + "v_proj", d, d, True)
        lin(b + "out_proj", d, d, True, ostd)

    for l in range(n_enc):
        b = e + f"layers.{l}."
        attn(b + "self_attn.", 1.5 * d ** -0.5)
        ln(b + "self_attn_layer_norm")
        lin(b + "fc1", ffn, d, True)
        lin(b + "fc2", d, ffn, True)
        ln(b + "final_layer_norm")
    ln(e + "layer_norm")
    for l in range(n_dec):
        b = dd + f"layers.{l}."
        attn(b + "self_attn.", 1.5 * d ** -0.5)
        ln(b + "self_attn_layer_norm")
        attn(b + "encoder_attn.", 2.0 * d ** -0.5, 2.0 * d ** -0.5)
        ln(b + "encoder_attn_layer_norm")
        lin(b + "fc1", ffn, d, True)
        lin(b + "fc2", d, ffn, True)
        ln(b + "final_layer_norm")
    ln(dd + "layer_norm")
    for pat, g in (gains or {}).items():
        for k in sd:
            if pat in k:
                sd[k] = sd[k] * g
    return sd


def add_random_cond_embedders(sd: dict, cond_dim: int = 16, num_mappers: int = 11, seed: int = 0) -> dict:
    """The difficulty / mapper / song-position embedders of `add_random_conditioning` alone (in place): for models whose
    conditioning enters as conv1 CHANNELS (project_encoder_input = false) nothing else changes."""
    tmp = {"encoder_embedder.weight": torch.zeros(1, 0)}
    add_random_conditioning(tmp, 1, 0, cond_dim, num_mappers, seed)
    del tmp["encoder_embedder.weight"]
    sd.update(tmp)
    return sd


# weaker token embedding + sharper / stronger cross-attention: the next token depends on WHICH encoder frames the
# query selects, not only on the previous token (random-init greedy decoding otherwise repeats a handful of ids).
# Kept mild on purpose: at (0.3, 6, 3) the REFERENCE itself in bfloat16 agrees with its fp32 self on only 73 % of
# teacher-forced steps (attention scores of magnitude ~50 rounded to 8 bits); at these values it is 96 %.
DIVERSE_GAINS = {"decoder_embedder": 0.5, "EncDecAttention.q.weight": 2.0, "EncDecAttention.o.weight": 2.0}


def add_random_conditioning(sd: dict, d_model: int, n_mels: int = 388, cond_dim: int = 16, num_mappers: int = 11, seed: int = 0) -> dict:
    """Adds the parameters of the reference's difficulty / mapper / song-position embedders
    (modeling_mapperatorinator.py:110-128,462-660) and widens `encoder_embedder.weight` by the 3 * cond_dim conditioning
    columns, in place.  Scales are chosen so that the conditioning moves the encoder input by about as much as the mel
    does (the reference's own initialisers -- xavier gain 0.1, zero bias -- would leave it invisible to a parity test)."""
    rng = np.random.default_rng(5000 + seed)
    C = cond_dim
    sd["encoder_embedder.weight"] = torch.cat([sd["encoder_embedder.weight"], _normal(rng, (d_model, 3 * C), 0.5 / np.sqrt(3 * C))], 1)

    def ln(prefix, n):
        sd[prefix + ".weight"] = 1.0 + _normal(rng, (n,), 0.1)
        sd[prefix + ".bias"] = _normal(rng, (n,), 0.1)

    def lin(prefix, n_out, n_in):
        sd[prefix + ".weight"] = _normal(rng, (n_out, n_in), n_in ** -0.5)
        sd[prefix + ".bias"] = _normal(rng, (n_out,), 0.1)

    sd["difficulty_embedder.basis_centers"] = torch.linspace(0, 1, 8) + _normal(rng, (8,), 0.02)
    sd["difficulty_embedder.basis_widths"] = 0.1 + _normal(rng, (8,), 0.01).abs()
    lin("difficulty_embedder.difficulty_proj.0", C, 8)
    ln("difficulty_embedder.difficulty_proj.1", C)
    lin("difficulty_embedder.difficulty_proj.4", C, C)
    ln("difficulty_embedder.difficulty_proj.5", C)
    sd["mapper_embedder.embedding.weight"] = _normal(rng, (num_mappers + 1, C), 0.5)
    ln("mapper_embedder.layer_norm", C)
    sd["song_pos_embedder.basis_centers"] = torch.linspace(0, 1, 10) + _normal(rng, (10,), 0.02)
    sd["song_pos_embedder.basis_widths"] = 0.1 + _normal(rng, (10,), 0.01).abs()
    lin("song_pos_embedder.position_proj.0", 2 * C, 20)
    ln("song_pos_embedder.position_proj.1", 2 * C)
    lin("song_pos_embedder.position_proj.4", C, 2 * C)
    ln("song_pos_embedder.position_proj.5", C)
    return sd


def synthetic_audio(batch: int, n_samples: int = 160000, seed: int = 0) -> torch.Tensor:
    """N(0,1) noise plus two tones, peak-normalised to [-1, 1] per row (mirrors
    `normalize_audio_samples`, osuT5/osuT5/dataset/data_utils.py:132-137)."""
    rng = np.random.default_rng(1000 + seed)
    x = rng.standard_normal((batch, n_samples)).astype(np.float32)
    t = np.arange(n_samples, dtype=np.float32) / 16000.0
    for b in range(batch):
        x[b] += 2.0 * np.sin(2 * np.pi * (220.0 * (1 + b % 5)) * t) + 1.0 * np.sin(2 * np.pi * 3520.0 * t + b)
    x /= np.abs(x).max(axis=1, keepdims=True)
    return torch.from_numpy(x)


def synthetic_audio_varied(batch: int, n_samples: int = 160000, seed: int = 0) -> torch.Tensor:
    """Non-stationary test audio: back-to-back 50-250 ms notes (a tone + one partial, random pitch and level) over a
    weak noise floor, peak-normalised per row.  Stationary noise makes every encoder frame look alike (frame-to-frame
    difference of the encoder states ~3 %), which leaves the decoder's cross-attention nothing to select; here the
    frames differ by ~25 % and greedy ids become diverse."""
    rng = np.random.default_rng(2000 + seed)
    x = 0.05 * rng.standard_normal((batch, n_samples)).astype(np.float32)
    t = np.arange(n_samples, dtype=np.float32) / 16000.0
    for b in range(batch):
        pos = 0
        while pos < n_samples:
            ln = int(rng.integers(800, 4000))
            f = float(rng.uniform(80, 7000))
            a = float(rng.uniform(0.2, 1.0))
            seg = slice(pos, min(n_samples, pos + ln))
            x[b, seg] += a * np.sin(2 * np.pi * f * t[seg]) + 0.5 * a * np.sin(2 * np.pi * 2.31 * f * t[seg])
            pos += ln
    x /= np.abs(x).max(axis=1, keepdims=True)
    return torch.from_numpy(x)


# ---- DiT ------------------------------------------------------------------------------------------
from mapperatorinator_amd.dit import DIT_PRESETS  # noqa: E402,F401  (depth, hidden, heads) per preset name


def random_dit_state_dict(depth: int, hidden: int, context_size: int = 272, class_size: int = 300, seed: int = 0,
                          std: float = 0.02) -> dict:
    """Keys of `DiT.state_dict()` (osu_diffusion/utils/models.py:213-279).  adaLN / final layers are
    zero-init in the reference (:270-279) => output identically 0; for parity tests they get N(0, std)."""
    rng = np.random.default_rng(7000 + seed)
    D = hidden
    sd = {}

    def lin(name, n_out, n_in, wstd=None, bstd=std):
        sd[name + ".weight"] = _normal(rng, (n_out, n_in), wstd if wstd is not None else (2.0 / (n_in + n_out)) ** 0.5)
        sd[name + ".bias"] = _normal(rng, (n_out,), bstd)

    lin("context_embedder.mlp.0", D, 2 * 128 + context_size, std * 2)
    lin("t_embedder.mlp.0", D, 256, std * 3)
    lin("t_embedder.mlp.2", D, D, std * 3)
    lin("y_embedder.class_embedding.0", D, class_size, std * 3)
    lin("y_embedder.class_embedding.2", D, D, std * 3)
    for l in range(depth):
        b = f"blocks.{l}."
        sd[b + "attn.in_proj_weight"] = _normal(rng, (3 * D, D), (2.0 / (4 * D)) ** 0.5)
        sd[b + "attn.in_proj_bias"] = _normal(rng, (3 * D,), std)
        lin(b + "attn.out_proj", D, D)
        lin(b + "mlp.fc1", 4 * D, D)
        lin(b + "mlp.fc2", D, 4 * D)
        lin(b + "adaLN_modulation.1", 6 * D, D, std * 2)
    lin("final_layer.adaLN_modulation.1", 2 * D, D, std * 2)
    lin("final_layer.linear", 4, D, std * 4)
    return sd


def synthetic_dit_inputs(T: int = 128, context_size: int = 272, class_size: int = 300, seed: int = 0):
    """One chunk's DiT inputs with CFG batch 2: z (2,2,T) U(-1,1) duplicated halves,
    c (2,272,T), y (2,C) multi-hot (cond / null)."""
    rng = np.random.default_rng(9000 + seed)
    z1 = rng.uniform(-1, 1, (1, 2, T)).astype(np.float32)
    c1 = rng.standard_normal((1, context_size, T)).astype(np.float32) * 0.5
    y = np.zeros((2, class_size), np.float32)
    y[0, rng.integers(0, class_size, 4)] = 1.0
    y[1, class_size - 1] = 1.0
    z = np.concatenate([z1, z1], 0)
    c = np.concatenate([c1, c1], 0)
    return torch.from_numpy(z), torch.from_numpy(c), torch.from_numpy(y)


def boost_timed_rows(sd: dict, tok, gain: float) -> dict:
    """Random weights almost never emit a timed event (CIRCLE, BEAT, HOLD_NOTE ...); scale their lm_head rows so that
    the types_first processors, which key on them, fire in the parity cases.  In place; used identically by
    oracle/make_golden.py and the tests."""
    from mapperatorinator_amd.server import TIMED_EVENT_NAMES, _ev, _has
    w = sd["transformer.lm_head.weight"]
    for name in TIMED_EVENT_NAMES:
        if _has(tok.event_start, name):
            w[_ev(tok.event_start, name):_ev(tok.event_end, name)] *= gain
    return sd


def synthetic_hit_objects(T: int, seed: int, span_ms: float = 60000.0):
    """Hit-object points for the diffusion pipeline cases: sorted times, playfield positions, distances, type rows."""
    rng = np.random.default_rng(seed)
    times = np.sort(rng.uniform(0, span_ms, T)).astype(np.float32)
    x = rng.uniform(0, 512, T).astype(np.float32)
    y = rng.uniform(0, 384, T).astype(np.float32)
    dist = rng.uniform(0, 200, T).astype(np.float32)
    typ = rng.integers(0, 16, T)
    return x, y, times, dist, typ


def synthetic_sliders(T: int, seed: int, every: int = 9):
    """DiffusionSlider lists over the points of `synthetic_hit_objects`: head, 1-6 anchors (a red anchor = the same point
    twice, as `events_to_sequence` emits it), last anchor, and the slider end as the next point; curve types cycle through
    Bezier / PerfectCurve / Catmull.  Consecutive sliders use disjoint points, like a real event stream."""
    from mapperatorinator_amd.diffusion_pipeline import DiffusionSlider
    rng = np.random.default_rng(seed)
    out, i, k = [], 2, 0
    while i + 9 < T:
        curve = ("Bezier", "PerfectCurve", "Catmull", "Bezier")[k % 4]
        n_anchor = 1 if (curve == "PerfectCurve" and k % 8 == 1) else int(rng.integers(1, 6))
        idx = [i]
        for a in range(n_anchor):
            idx.append(i + 1 + a)
            if curve == "Bezier" and k % 4 == 3 and a == n_anchor // 2:
                idx.append(i + 1 + a)                      # red anchor
        idx.append(i + 1 + n_anchor)                       # last anchor
        end = i + 2 + n_anchor
        out.append(DiffusionSlider(np.array(idx), end, curve, float(rng.uniform(20, 500))))
        i = end + 1 + int(rng.integers(0, every))
        k += 1
    return out


def pipeline_windows(seq_len: int, max_seq_len: int, overlap_buffer: int):
    """(start, end) of every diffusion window (reference diffusion_pipeline.py:277-278)."""
    return [(i, min(i + max_seq_len, seq_len))
            for i in range(0, seq_len - overlap_buffer * 2, max_seq_len - overlap_buffer * 2)]


@dataclasses.dataclass
class TimingPointLike:
    """The attributes of `slider.TimingPoint` that the diffusion stage reads (diffusion_pipeline.py:423-424, 440-445)."""
    offset: datetime.timedelta
    ms_per_beat: float
    parent: Optional["TimingPointLike"] = None


def synthetic_timing(seed: int, span_ms: float = 60000.0):
    """Three uninherited timing points, each followed by an inherited one (`parent` set) a little later."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(3):
        red = TimingPointLike(datetime.timedelta(milliseconds=int(span_ms * k / 3) + (137 if k else 0)), float(rng.uniform(280, 520)))
        out.append(red)
        out.append(TimingPointLike(red.offset + datetime.timedelta(milliseconds=int(rng.integers(500, 4000))), -100.0 / float(rng.uniform(0.5, 2.0)), red))
    return out


def synthetic_event_stream(n_objects: int, seed: int, *, types_first: bool = False, with_positions: bool = False,
                           oddities: bool = True):
    """A hit-object event stream in the format the T5 stage emits (reference osuT5/osuT5/tokenizer + the group orders
    drawn in data_utils.py:757-766): per object `t, (dist | pos_x pos_y), [new_combo], [scroll_speed], TYPE` -- the type
    token first with `types_first` -- anchors without a time of their own, beats / measures / timing points / kiai in
    between.  `oddities` adds what a sampled stream can contain and the reference's code tolerates: a coordinate of 0,
    a distance of 0, a slider head without scroll speed, anchors and a slider end without a head, a slider end straight
    after the head, attribute events no type token claims at the very end."""
    from mapperatorinator_amd.event import Event, EventType as ET
    rng = np.random.default_rng(seed)
    ev, t = [], 0

    def emit(type_name, *, time=None, x=None, y=None, dist=None, nc=False, sv=None, value=0):
        attrs = []
        if time is not None:
            attrs.append(Event(ET.TIME_SHIFT, int(time)))
            if rng.random() < 0.5:
                attrs.append(Event(ET.SNAPPING, int(rng.integers(0, 16))))
        if with_positions and x is not None:
            attrs += [Event(ET.POS_X, int(x)), Event(ET.POS_Y, int(y))]
        elif dist is not None:
            attrs.append(Event(ET.DISTANCE, int(dist)))
        if nc:
            attrs.append(Event(ET.NEW_COMBO, 0))
        if sv is not None:
            attrs.append(Event(ET.SCROLL_SPEED, int(sv)))
        head = [Event(ET[type_name], value)]
        ev.extend(head + attrs if types_first else attrs + head)
        if type_name in ("CIRCLE", "SLIDER_HEAD", "SLIDER_END") and rng.random() < 0.4:
            ev.append(Event(ET.HITSOUND, int(rng.integers(0, 72))))
            ev.append(Event(ET.VOLUME, int(rng.integers(0, 100))))

    def xy():
        return int(rng.integers(1, 512)), int(rng.integers(1, 384))

    for k in range(n_objects):
        t += int(rng.integers(60, 900))
        if k % 5 == 0:
            emit("BEAT" if k % 10 else "MEASURE", time=t - 30)
        if k % 37 == 11:
            emit("TIMING_POINT", time=t - 20)
        if k % 41 == 13:
            emit("KIAI", time=t - 10, value=1)
        kind = rng.random()
        x, y = xy()
        if kind < 0.45:
            zero = oddities and k % 23 == 7
            emit("CIRCLE", time=t, x=0 if zero else x, y=y, dist=0 if zero else rng.integers(1, 400), nc=rng.random() < 0.3)
        elif kind < 0.55:
            emit("SPINNER", time=t, x=256, y=192, dist=rng.integers(0, 300))
            t += int(rng.integers(400, 3000))
            emit("SPINNER_END", time=t, x=256, y=192, dist=rng.integers(0, 300))
        else:
            no_sv = oddities and k % 17 == 5
            emit("SLIDER_HEAD", time=t, x=x, y=y, dist=rng.integers(1, 400), nc=rng.random() < 0.3,
                 sv=None if no_sv else rng.integers(30, 400))
            head_t = t
            if oddities and k % 29 == 3:                                  # slider end straight after the head
                t += int(rng.integers(100, 600))
                emit("SLIDER_END", time=t, x=x, y=y, dist=rng.integers(1, 400))
                continue
            curve = ("BEZIER_ANCHOR", "PERFECT_ANCHOR", "CATMULL_ANCHOR")[int(rng.integers(0, 3))]
            for a in range(1 if curve == "PERFECT_ANCHOR" else int(rng.integers(0, 6))):
                red = curve == "BEZIER_ANCHOR" and rng.random() < 0.25
                emit("RED_ANCHOR" if red else curve, x=xy()[0], y=xy()[1], dist=rng.integers(1, 200))
            span = int(rng.integers(80, 1200))
            t += span
            emit("LAST_ANCHOR", time=t, x=xy()[0], y=xy()[1], dist=rng.integers(1, 200))
            repeats = int(rng.integers(1, 7))
            t = head_t + span * repeats + int(rng.integers(-3, 4))
            emit("SLIDER_END", time=max(t, head_t + 1), x=xy()[0], y=xy()[1], dist=rng.integers(1, 400))
        if oddities and k % 31 == 19:                                     # anchors and an end nobody opened
            emit("BEZIER_ANCHOR", x=xy()[0], y=xy()[1], dist=rng.integers(1, 200))
            t += 50
            emit("SLIDER_END", time=t, x=xy()[0], y=xy()[1], dist=rng.integers(1, 400))
    if oddities:
        t += 100
        ev.append(Event(ET.TIME_SHIFT, t))
        ev.append(Event(ET.DISTANCE, 17))
    return ev


def synthetic_diffusion_tokenizer_state(seed: int) -> dict:
    """A state dict of the diffusion class tokenizer (osu_diffusion/utils/tokenizer.py:216-230): some beatmaps with a
    style class, mapper and descriptors each; every fourth seed drops a family (its count stays 0)."""
    rng = np.random.default_rng(seed)
    n_maps, n_mappers, n_desc = int(rng.integers(3, 40)), int(rng.integers(2, 9)), int(rng.integers(2, 10))
    ids = [int(i) for i in rng.choice(50, n_maps, replace=False)]
    st = dict(beatmap_idx={b: k for k, b in enumerate(ids)}, num_classes=n_maps + 1,
              num_diff_classes=int(rng.integers(3, 12)), max_difficulty=float(rng.uniform(6, 10)),
              beatmap_mapper={b: int(rng.integers(0, n_mappers + 2)) for b in ids},
              mapper_idx={u: u for u in range(n_mappers)}, num_mapper_classes=n_mappers + 1,
              beatmap_descriptors={b: [int(i) for i in rng.choice(n_desc, int(rng.integers(1, 3)), replace=False)] for b in ids},
              descriptor_idx={f"d{i}": i for i in range(n_desc)}, num_descriptor_classes=n_desc + 1,
              num_cs_classes=int(rng.integers(3, 14)))
    drop = seed % 4
    if drop == 1:
        st.update(beatmap_idx={}, num_classes=0)
    elif drop == 2:
        st.update(mapper_idx={}, beatmap_mapper={}, num_mapper_classes=0)
    elif drop == 3:
        st.update(num_cs_classes=0, num_diff_classes=0)
    return st


class SeededMultinomial:
    """A deterministic, device-independent stand-in for `torch.multinomial(probs, k)` WITHOUT replacement, for the
    beam-sample parity cases: the reference (CPU) and the HIP path (GPU) cannot share a torch generator, so both are handed
    this sampler (oracle/make_golden.py patches it into HF's `_get_top_k_continuations`; beam.py takes it as `sample_fn`).
    Per row, k times: inverse CDF of the remaining weights in float64 against one uniform of a numpy stream (once the
    weighted categories are used up, the lowest unused index -- torch also keeps drawing zero-weight categories then).  Weights
    that differ by fp32 rounding between the two sides select the same index unless a uniform lands within ~1e-7 of a
    CDF step."""

    def __init__(self, seed: int):
        self.rng = np.random.default_rng(seed)
        self.calls = 0

    def __call__(self, probs: torch.Tensor, num_samples: int, replacement: bool = False, **_):
        if replacement:
            raise NotImplementedError("beam-sample draws without replacement")
        self.calls += 1
        w_all = probs.detach().to(torch.float64).cpu().numpy()
        out = np.empty((w_all.shape[0], num_samples), np.int64)
        for r, w in enumerate(w_all):
            w, taken = w.copy(), set()
            for j in range(num_samples):
                c = np.cumsum(w)
                if c[-1] > 0:
                    i = min(int(np.searchsorted(c, self.rng.random() * c[-1], side="right")), len(w) - 1)
                else:      # every weighted category is drawn: torch goes on with zero-weight ones (its exponential-race top-k does
                    i = next(n for n in range(len(w)) if n not in taken)      # not stop); here the lowest unused index
                out[r, j] = i
                taken.add(i)
                w[i] = 0.0
        return torch.from_numpy(out).to(probs.device)
