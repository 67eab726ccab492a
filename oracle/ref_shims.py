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
        return WhisperConfig()

    T5Config.from_pretrained = classmethod(_t5_from_pretrained)
    WhisperConfig.from_pretrained = classmethod(_whisper_from_pretrained)
    _installed = True
