"""TEST INFRASTRUCTURE ONLY -- makes the *unmodified* reference importable in this container.

Used by `oracle/make_golden.py` and by the `-m "not gpu"` pinning tests (which skip when
`/root/reference` is absent, e.g. on the GPU box).  Nothing under `mapperatorinator_amd/` may
import this module.

The reference (`/root/reference`, read-only) depends on packages that are not installable
offline (hydra, omegaconf, slider, pydub, nnAudio, wandb, peft, ...).  None of them takes part
in the audio->event hot path arithmetic, except nnAudio, whose MelSpectrogram is replaced by
the CPU restatement in `oracle/mel.py` ("parity unpinned" for that one
piece -- the reference repo has no mel vectors and the nnAudio wheel is absent).

Recipe follows SURVEY.md Appendix C.
"""
from __future__ import annotations

import contextlib
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    import os
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "osuT5"))


class _Bag(types.ModuleType):
    """Module whose unknown attributes resolve to inert placeholder classes."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _stub(name: str) -> _Bag:
    if name in sys.modules:
        return sys.modules[name]
    mod = _Bag(name)
    mod.__path__ = []  # behave as a package so that submodule imports resolve
    sys.modules[name] = mod
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(_stub(parent), child, mod)
    return mod


_T5_DIMS = {
    # name suffix -> (d_model, d_kv, d_ff, num_layers, num_decoder_layers, num_heads)
    "small": (512, 64, 1024, 8, 8, 6),
    "base": (768, 64, 2048, 12, 12, 12),
    "large": (1024, 64, 2816, 24, 24, 16),
}


def local_t5_config(name: str):
    """google/t5-v1_1-{small,base,large} dims without touching the hub
    (replaces T5Config.from_pretrained at configuration_mapperatorinator.py:67-68)."""
    from transformers import T5Config
    for suffix, (d, dkv, dff, nl, ndl, nh) in _T5_DIMS.items():
        if name.endswith(suffix):
            return T5Config(
                vocab_size=32128, d_model=d, d_kv=dkv, d_ff=dff, num_layers=nl,
                num_decoder_layers=ndl, num_heads=nh, relative_attention_num_buckets=32,
                relative_attention_max_distance=128, dropout_rate=0.1,
                layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu",
                tie_word_embeddings=False, is_encoder_decoder=True, use_cache=True,
                pad_token_id=0, eos_token_id=1, decoder_start_token_id=0,
            )
    raise KeyError(name)


_installed = False


def install():
    """Idempotently install import stubs + offline config patches, put the reference on sys.path."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference checkout not present at " + REFERENCE_ROOT)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import transformers  # noqa: F401  (must be imported before any stub exists: it probes packages)
    from transformers import T5Config, WhisperConfig

    for name in ("slider", "slider.beatmap", "slider.curve", "slider.position", "slider.mod",
                 "pydub", "wandb", "peft", "hydra", "hydra.core", "hydra.core.config_store",
                 "hydra.utils", "omegaconf", "nnAudio", "nnAudio.features", "rosu_pp_py"):
        _stub(name)

    om = sys.modules["omegaconf"]
    om.MISSING = "???"

    class _OmegaConf:
        @staticmethod
        def register_new_resolver(*a, **k):
            return None

        @staticmethod
        def to_object(x):
            return x

        @staticmethod
        def to_container(x, **k):
            return x

    om.OmegaConf = _OmegaConf
    om.DictConfig = dict
    om.open_dict = lambda cfg: contextlib.nullcontext(cfg)

    class _ConfigStore:
        _inst = None

        @classmethod
        def instance(cls):
            if cls._inst is None:
                cls._inst = cls()
            return cls._inst

        def store(self, *a, **k):
            return None

    sys.modules["hydra.core.config_store"].ConfigStore = _ConfigStore
    sys.modules["hydra"].main = lambda *a, **k: (lambda f: f)

    # nnAudio.features.MelSpectrogram -> CPU restatement (oracle/mel.py)
    from oracle.mel import NnAudioMelSpectrogram
    sys.modules["nnAudio.features"].MelSpectrogram = NnAudioMelSpectrogram
    sys.modules["nnAudio"].features = sys.modules["nnAudio.features"]

    # offline backbone configs
    def _t5_from_pretrained(cls, name, *a, **k):
        return local_t5_config(str(name))

    def _whisper_from_pretrained(cls, name, *a, **k):
        # openai/whisper-{tiny,base,small} dims without the hub; `cls` is WhisperConfig or one of the reference's forks
        # (VarWhisperConfig.from_pretrained("openai/whisper" + suffix), configuration_mapperatorinator.py:73-78)
        for suffix, (d, layers, heads) in _WHISPER_DIMS.items():
            if str(name).endswith(suffix):
                return cls(d_model=d, encoder_layers=layers, decoder_layers=layers, encoder_attention_heads=heads,
                           decoder_attention_heads=heads, encoder_ffn_dim=4 * d, decoder_ffn_dim=4 * d)
        return cls()

    T5Config.from_pretrained = classmethod(_t5_from_pretrained)
    WhisperConfig.from_pretrained = classmethod(_whisper_from_pretrained)

    # ---- third-party API drift between the reference's pin (transformers 4.57) and the installed 5.x, Whisper forks only ----
    # (1) ROPE_INIT_FUNCTIONS["default"] (modeling_varwhisper.py:207) left the table: the published default RoPE
    #     initialiser restated -- inv_freq[i] = theta^(-2i / head_dim), attention factor 1
    import torch
    from transformers import modeling_rope_utils as mru

    def _default_rope(config, device=None, seq_len=None, **kw):
        head_dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        dim = int(head_dim * getattr(config, "partial_rotary_factor", 1.0))
        inv_freq = 1.0 / (config.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(device=device, dtype=torch.float) / dim))
        return inv_freq, 1.0

    mru.ROPE_INIT_FUNCTIONS.setdefault("default", _default_rope)
    # (2) torchaudio is not installed: spectrogram.py:38-49 builds torchaudio.transforms.MelSpectrogram -> the torch.stft
    #     restatement of oracle/mel.py (parity unpinned for that one stage, like nnAudio)
    from oracle import mel as omel

    class _TorchaudioMel(torch.nn.Module):
        def __init__(self, sample_rate=16000, n_fft=1024, n_mels=128, hop_length=128, center=True, f_min=0.0, f_max=None,
                     pad_mode="reflect", **kw):
            super().__init__()
            self.kw = dict(n_fft=n_fft, hop=hop_length, n_mels=n_mels, sr=sample_rate, f_min=float(f_min), f_max=float(f_max),
                           pad_mode=pad_mode)

        def forward(self, x):   # torchaudio returns (B, n_mels, frames), power spectrogram (the wrapper applies log1p itself)
            return omel.mel_spectrogram_torchaudio(x, log_scale=False, **self.kw).permute(0, 2, 1)

    ta, tat = types.ModuleType("torchaudio"), types.ModuleType("torchaudio.transforms")
    tat.MelSpectrogram = _TorchaudioMel
    ta.transforms = tat
    sys.modules.setdefault("torchaudio", ta)
    sys.modules.setdefault("torchaudio.transforms", tat)
    _installed = True


_WHISPER_DIMS = {"tiny": (384, 4, 6), "base": (512, 6, 8), "small": (768, 12, 12)}


def varwhisper_module():
    """The reference's VarWhisper fork, importable under transformers 5.x: `_tied_weights_keys` is a list there and a dict
    here (the fork unties its head anyway: configs/model/varwhisper_*_v3.yaml `tie_word_embeddings: false`)."""
    install()
    from osuT5.osuT5.model.custom_transformers import modeling_varwhisper as mv
    cls = mv.VarWhisperForConditionalGeneration
    cls._tied_weights_keys = {}
    if not getattr(cls, "_mh_kwargs_filtered", False):
        # transformers 5.x `generate` hands its own bookkeeping kwargs (`next_sequence_length`, ...) to
        # prepare_inputs_for_generation, and the wrapper forwards every kwarg it does not know
        # (modeling_mapperatorinator.py:265-268) into a forward with a closed signature: drop what that signature lacks
        import inspect
        orig = cls.forward
        known = set(inspect.signature(orig).parameters)

        def forward(self, *a, **k):
            return orig(self, *a, **{kk: v for kk, v in k.items() if kk in known})

        cls.forward = forward
        cls._mh_kwargs_filtered = True
        # `cache[layer_idx] -> (keys, values)` (modeling_varwhisper.py:538, the cross-attention cache after the first step)
        # existed on the pinned 4.57 Cache classes; 5.x keeps the tensors on per-layer objects
        from transformers.cache_utils import Cache
        if not hasattr(Cache, "__getitem__"):
            Cache.__getitem__ = lambda self, i: (self.layers[i].keys, self.layers[i].values)
    return mv


def _filter_forward_kwargs(cls):
    """transformers 5.x `generate` hands its own bookkeeping kwargs (`next_sequence_length`, ...) to
    prepare_inputs_for_generation, and the wrapper forwards every kwarg it does not know (modeling_mapperatorinator.py:265-268)
    into a forward with a closed signature: drop what that signature lacks (as varwhisper_module does for its fork)."""
    if getattr(cls, "_mh_kwargs_filtered", False):
        return
    import inspect
    orig = cls.forward
    params = inspect.signature(orig).parameters
    if any(p.kind == inspect.Parameter.VAR_KEYWORD for p in params.values()):
        cls._mh_kwargs_filtered = True
        return
    known = set(params)

    def forward(self, *a, **k):
        return orig(self, *a, **{kk: v for kk, v in k.items() if kk in known})

    cls.forward = forward
    cls._mh_kwargs_filtered = True


def ropewhisper_module():
    """The reference's RoPEWhisper fork (custom_transformers/modeling_ropewhisper.py), runnable under transformers 5.x:
    (1) the closed forward signature (see _filter_forward_kwargs); (2) `cache.key_cache[i]` / `cache.value_cache[i]`
    (modeling_ropewhisper.py:438-439, the cross-attention cache after the first step) were attributes of the pinned 4.57 Cache
    classes; 5.x keeps the tensors on per-layer objects -> read-only views with the same indexing."""
    install()
    from osuT5.osuT5.model.custom_transformers import modeling_ropewhisper as mr
    cls = mr.RoPEWhisperForConditionalGeneration
    if isinstance(getattr(cls, "_tied_weights_keys", None), list):
        cls._tied_weights_keys = {}
    _filter_forward_kwargs(cls)
    from transformers.cache_utils import Cache
    if not hasattr(Cache, "key_cache"):
        class _LayerView:
            def __init__(self, cache, attr):
                self.cache, self.attr = cache, attr

            def __getitem__(self, i):
                return getattr(self.cache.layers[i], self.attr)

            def __len__(self):
                return len(self.cache.layers)

        Cache.key_cache = property(lambda self: _LayerView(self, "keys"))
        Cache.value_cache = property(lambda self: _LayerView(self, "values"))
    return mr


def hf_whisper_module():
    """Stock transformers WhisperForConditionalGeneration (the backbone of 'openai/whisper-*' configs: third-party code, not a
    fork): only the kwargs filter, where its forward has a closed signature."""
    install()
    from transformers.models.whisper import modeling_whisper as mw
    _filter_forward_kwargs(mw.WhisperForConditionalGeneration)
    return mw
