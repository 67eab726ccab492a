"""mapperatorinator_amd -- MI355X-native (gfx950) audio->event hot path of Mapperatorinator.

  mel frontend (K1) -> osuT5 encoder/decoder with KV-cached AR decode (K3-K6) -> osu_diffusion DiT +
  DDPM refinement (K7-K9), hand-written HIP behind the C ABI of include/mapperhip.h, with the host
  side mirroring the reference's Python seams: `model_generate`, `Mapperatorinator`,
  `DiT.forward_with_cfg`, `diffusion.p_sample_loop`, `Tokenizer` / `Event`.
"""
import os as _os


def configure_runtime(graph_packet_capture: bool = False) -> dict:
    """EXPLICIT opt-in to a HIP runtime setting, for hosts that own their process (bench.py, a server's entry point:
    INTEGRATION.md 3) -- importing the package never touches `os.environ`.

    `graph_packet_capture=False` asks ROCm 7.2's CLR to replay hipGraphs through its classic per-node submission instead of
    its captured-AQL-packet path (`DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`): the decode loop's step graphs replay 1.5-1.9 % faster
    that way on the headline workload (profiles/r05_graph_packet_capture.txt; measured on ROCm 7.2.0 only).  It is a debug
    knob of that runtime version, it affects EVERY hipGraph user of the process, and the runtime reads it once, when it
    initialises: call this before the first HIP call.  An explicit value already in the environment wins.  Returns
    {"variable", "value", "applied", "effective"}: `effective` is False (and a RuntimeWarning is raised) when HIP was
    initialised before the call, i.e. when the setting cannot take effect any more."""
    import sys
    import warnings
    var = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
    want = "1" if graph_packet_capture else "0"
    applied = var not in _os.environ
    if applied:
        _os.environ[var] = want
    torch = sys.modules.get("torch")
    late = bool(torch is not None and torch.cuda.is_initialized())
    if late:
        warnings.warn(f"mapperatorinator_amd.configure_runtime: HIP is already initialised, {var} cannot take effect in this "
                      "process (call it before the first torch.cuda / HIP call)", RuntimeWarning, stacklevel=2)
    return {"variable": var, "value": _os.environ[var], "applied": applied, "effective": not late}


from .event import ContextType, Event, EventRange, EventType  # noqa: F401,E402
from .tokenizer import Tokenizer  # noqa: F401,E402

__all__ = ["ContextType", "Event", "EventRange", "EventType", "Tokenizer", "MapperatorinatorHIP",
           "model_generate", "DiTHIP", "create_diffusion", "MelSpectrogram", "configure_runtime"]


def __getattr__(name):  # torch-dependent pieces are imported lazily
    if name == "MapperatorinatorHIP":
        from .modeling import MapperatorinatorHIP
        return MapperatorinatorHIP
    if name in ("model_generate", "get_eos_token_id"):
        from . import server
        return getattr(server, name)
    if name in ("DiTHIP", "create_diffusion", "InpaintSpec", "SpacedDiffusionHIP"):
        from . import dit
        return getattr(dit, name)
    if name in ("encode_events", "decode_tokens"):
        from . import tokenizer
        return getattr(tokenizer, name)
    if name in ("DiffusionPipelineHIP", "points_to_sequence", "events_to_sequence", "events_with_pos", "DiffusionTokenizer",
                "DiffusionGenerationConfig", "get_class_vector"):
        from . import diffusion_pipeline
        return getattr(diffusion_pipeline, name)
    if name in ("SequentialWindowScheduler", "SongJob"):
        from . import scheduler
        return getattr(scheduler, name)
    if name == "MelSpectrogram":
        from .mel import MelSpectrogram
        return MelSpectrogram
    raise AttributeError(name)
