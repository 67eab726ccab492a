"""mapperatorinator_amd -- MI355X-native (gfx950) audio->event hot path of Mapperatorinator.

  mel frontend (K1) -> osuT5 encoder/decoder with KV-cached AR decode (K3-K6) -> osu_diffusion DiT +
  DDPM refinement (K7-K9), hand-written HIP behind the C ABI of include/mapperhip.h, with the host
  side mirroring the reference's Python seams: `model_generate`, `Mapperatorinator`,
  `DiT.forward_with_cfg`, `diffusion.p_sample_loop`, `Tokenizer` / `Event`.
"""
import os as _os

# HIP runtime knob (read once, when the runtime initialises -- import this package before the first torch.cuda call): the step
# graphs of the decode loop replay faster through the runtime's classic per-node submission than through its captured-AQL-packet
# path: 272-274 vs 277-278 ms per batch of the headline workload, 4 chains 335 vs 416 ms (profiles/r05_graph_packet_capture.txt).
# An explicit value in the environment wins; C hosts of libmapperhip.so set it themselves.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from .event import ContextType, Event, EventRange, EventType  # noqa: F401,E402
from .tokenizer import Tokenizer  # noqa: F401,E402

__all__ = ["ContextType", "Event", "EventRange", "EventType", "Tokenizer", "MapperatorinatorHIP",
           "model_generate", "DiTHIP", "create_diffusion", "MelSpectrogram"]


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
