"""Host glue of the osu_diffusion stage: what `DiffisionPipeline.generate` does between
`events_to_sequence` and `events_with_pos` (reference diffusion_pipeline.py:111-287), on top of the HIP DiT + DDPM
loop (K7-K9).  Integer / indexing logic on the host, every float of the hot loop on the device:

  points_to_sequence   the tensor assembly at the end of `events_to_sequence` (:361-387): normalised positions,
                       times, and the 272-row conditioning `timestep_embedding(0.1 t, 128) | timestep_embedding(d, 128)
                       | 16 one-hot types`
  DiffusionPipelineHIP.generate_positions
                       banded mask (:145-148), CFG doubling (:158-166), `random_init` (:168-169), the overlapping
                       window loop (:276-284), per-window in-paint mask incl. start_time / end_time (:223-234),
                       `p_sample_loop` (:243-252), the refine iterations (:254-267) and `to_positions` (:171-176)

Round 5: the integer host code either side of that seam is here too, so the stage is callable with the reference's own
signature `generate(events, generation_config, timing) -> events`:

  event_times          `update_event_times` (osuT5/osuT5/dataset/data_utils.py:724-804): the time of every event,
                       anchor events interpolated between the timed events around them
  group_events         `get_groups` (data_utils.py:907-979): events -> hit-object records + the event indices of each
  events_to_sequence   (diffusion_pipeline.py:289-438) records -> points, `seq_indices`, the DiffusionSlider list
  events_with_pos      (:447-469) positions written back as POS_X / POS_Y events
  get_class_vector     (:65-109) the multi-hot class vector of a GenerationConfig
All of it pinned against the reference's functions run here (`tests/golden/events_to_sequence.npz`, made by
`oracle/make_golden.py events`).  The slider end re-projection of `denoised_fn` (:208-220) runs on the device inside the DDPM
graph (csrc/slider.hip).  `pad_sequence=True` (:186-193) is reproduced with
the reference's own quirk: the pad positions stay attendable (the attention kernels' `open_from`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .dit import BandMask, DiTHIP, InpaintSpec, SliderInpaintSpec, create_diffusion


@dataclass
class DiffusionSlider:
    """(diffusion_pipeline.py:30-35) sequence points of the head and anchors, the point of the slider end, the curve
    type of the first anchor ('Bezier' | 'PerfectCurve' | 'Catmull') and the length in playfield pixels"""
    seq_indices: np.ndarray
    end_index: int
    curve_type: Optional[str]
    length: Optional[float]


# one-hot row of each hit-object type inside the 16 type rows (diffusion_pipeline.py:304-315); +1 for a new combo on
# CIRCLE / SLIDER_HEAD (:339-340), + repeat_type(repeats) on SLIDER_END (:343-347)
EVENT_INDEX = {"CIRCLE": 0, "SPINNER": 2, "SPINNER_END": 3, "SLIDER_HEAD": 4, "BEZIER_ANCHOR": 6, "PERFECT_ANCHOR": 7,
               "CATMULL_ANCHOR": 8, "RED_ANCHOR": 9, "LAST_ANCHOR": 10, "SLIDER_END": 11}
PLAYFIELD = (512.0, 384.0)


def repeat_type(repeat: int) -> int:
    """(osu_diffusion/utils/data_loading.py:43-49)"""
    if repeat < 4:
        return repeat - 1
    return 3 if repeat % 2 == 0 else 4


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """(osu_diffusion/utils/positional_embedding.py:28-49) -- same torch ops in the same order, on t's device."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def points_to_sequence(x, y, time, distance, type_index):
    """Hit-object points -> (seq_x (2,T), seq_o (T,), seq_c (272,T)), the tensors `events_to_sequence` returns
    (diffusion_pipeline.py:361-387).  `type_index` in [0, 16) is the final one-hot row (EVENT_INDEX + offsets)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    T = x.shape[0]
    seq = torch.zeros(20, T)
    seq[0], seq[1] = x, torch.as_tensor(y, dtype=torch.float32)
    seq[2], seq[3] = torch.as_tensor(time, dtype=torch.float32), torch.as_tensor(distance, dtype=torch.float32)
    idx = torch.as_tensor(type_index, dtype=torch.long)
    if T and (int(idx.min()) < 0 or int(idx.max()) >= 16):
        raise ValueError("type_index out of [0, 16)")
    seq[idx + 4, torch.arange(T)] = 1
    seq_x = seq[:2, :] / torch.tensor(PLAYFIELD).unsqueeze(1) * 2 - 1
    seq_o, seq_d = seq[2, :], seq[3, :]
    seq_c = torch.concatenate([timestep_embedding(seq_o * 0.1, 128).T, timestep_embedding(seq_d, 128).T, seq[4:, :]], 0)
    return seq_x, seq_o, seq_c


def band_mask(T: int, seq_len: int, device="cpu") -> torch.Tensor:
    """(diffusion_pipeline.py:145-148) True = masked; column i is visible from rows [i - seq_len, i + seq_len)."""
    q = torch.arange(T, device=device)[:, None]
    k = torch.arange(T, device=device)[None, :]
    return ~((q >= k - seq_len) & (q < k + seq_len))


# ---- events <-> points: the integer host code either side of the diffusion stage -------------------------------------
# Event classes by NAME, so the reference's own Event / EventType objects and ours (mapperatorinator_amd.event) both work.
_ANCHORS = frozenset(("BEZIER_ANCHOR", "PERFECT_ANCHOR", "CATMULL_ANCHOR", "RED_ANCHOR"))            # data_utils.py:53-58
_TIMED = frozenset(("CIRCLE", "SPINNER", "SPINNER_END", "SLIDER_HEAD", "LAST_ANCHOR", "SLIDER_END", "BEAT", "MEASURE",
                    "TIMING_POINT", "KIAI", "HOLD_NOTE", "HOLD_NOTE_END", "DRUMROLL", "DRUMROLL_END", "DENDEN",
                    "DENDEN_END", "SCROLL_SPEED_CHANGE"))                                             # data_utils.py:60-78
_OBJECT_TYPES = _TIMED | _ANCHORS                                                                     # TYPE_EVENTS, :29-51
_CURVE_OF_ANCHOR = {"BEZIER_ANCHOR": "Bezier", "PERFECT_ANCHOR": "PerfectCurve", "CATMULL_ANCHOR": "Catmull",
                    "RED_ANCHOR": "Bezier", "LAST_ANCHOR": "Bezier"}


def event_times(events: Sequence, types_first: bool = False, end_time: Optional[float] = None) -> list:
    """Time of every event of a whole stream = `update_event_times(events, [], end_time, types_first)`
    (data_utils.py:724-804).  Two sweeps.  (1) carry: an event takes the value of the TIME_SHIFT that governs its group --
    the latest one at or before it, or with `types_first` the one right behind the type token.  (2) anchors have no
    TIME_SHIFT of their own: walking AGAINST the group order (backwards; forwards with `types_first`) from the timed
    event that follows the anchors in time, the k anchors still ahead of the walk split the remaining interval evenly,
    truncated to int at every anchor (so the truncations compound exactly like the reference's); the attribute events
    of an anchor group take the anchor's time."""
    n = len(events)
    name = [e.type.name for e in events]
    times, now = [], 0
    for i in range(n):
        if types_first:
            if i + 1 < n and name[i + 1] == "TIME_SHIFT":
                now = events[i + 1].value
        elif name[i] == "TIME_SHIFT":
            now = events[i].value
        times.append(now)
    if n == 0:
        return times
    # for every position: where the nearest TIME_SHIFT lies on the far side of the walk, and how many anchors sit between
    walk = range(n) if types_first else range(n - 1, -1, -1)
    far_shift, anchors_to_shift = [None] * n, [0] * n
    shift_at, run = None, 0
    for i in reversed(walk):                       # from the far side towards the walk's start
        if name[i] == "TIME_SHIFT":
            shift_at, run = i, 0
        elif name[i] in _ANCHORS:
            run += 1
        far_shift[i], anchors_to_shift[i] = shift_at, run
    edge_time = (end_time if end_time is not None else times[-1]) if types_first else 0
    now = times[0] if types_first else (end_time if end_time is not None else times[-1])
    inside_anchor_run = False
    for i in walk:
        if name[i] in _TIMED:
            inside_anchor_run = False
        elif name[i] in _ANCHORS:
            inside_anchor_run = True
        if not inside_anchor_run:
            now = times[i]
        elif name[i] in _ANCHORS:
            k = anchors_to_shift[i]
            far = times[far_shift[i]] if far_shift[i] is not None else edge_time
            now = int((now - far) / (k + 1) * k + far)
            times[i] = now
        else:
            times[i] = now
    return times


@dataclass
class HitObjectRecord:
    """The fields of the reference's `Group` (data_utils.py:907-919) that the diffusion stage reads."""
    type_name: Optional[str] = None
    time: float = 0
    distance: Optional[float] = None
    x: Optional[float] = None
    y: Optional[float] = None
    new_combo: bool = False
    scroll_speed: Optional[float] = None


def group_events(events: Sequence, times: Optional[Sequence] = None, types_first: bool = False):
    """`get_groups` (data_utils.py:922-979): one record per object-type event, with the attribute events that belong to it
    (before the type token; behind it with `types_first`), and the event indices of every record.  Attribute events
    that no type token claims join the last record (IndexError when there is none, like the reference)."""
    records, members = [], []
    cur, idx = HitObjectRecord(), []
    for i, e in enumerate(events):
        nm = e.type.name
        idx.append(i)
        if nm in _OBJECT_TYPES:
            if types_first and cur.type_name is not None:
                records.append(cur)
                members.append(idx[:-1])
                cur, idx = HitObjectRecord(), [i]
            cur.type_name = nm
            if times is not None:
                cur.time = times[i]
            if not types_first:
                records.append(cur)
                members.append(idx)
                cur, idx = HitObjectRecord(), []
        elif nm == "TIME_SHIFT":
            cur.time = e.value
        elif nm == "DISTANCE":
            cur.distance = e.value
        elif nm == "POS_X":
            cur.x = e.value
        elif nm == "POS_Y":
            cur.y = e.value
        elif nm == "NEW_COMBO":
            cur.new_combo = True
        elif nm == "SCROLL_SPEED":
            cur.scroll_speed = e.value / 100
    if cur.type_name is not None:
        records.append(cur)
        members.append(idx)
    elif idx:
        members[-1].extend(idx)
    return records, members


def timing_point_at(time, timing_points: Sequence):
    """(diffusion_pipeline.py:440-445) the last timing point at or before `time` (a timedelta), else the first."""
    hit = [tp for tp in timing_points if tp.offset <= time]
    return hit[-1] if hit else timing_points[0]


def events_to_sequence(events: Sequence, timing: Optional[Sequence], slider_multiplier: float, *,
                       types_first: bool = False, has_sv: bool = True):
    """(diffusion_pipeline.py:289-438) -> (seq_x (2,T), seq_o (T,), seq_c (272,T), T, seq_indices {event index -> point},
    [DiffusionSlider]).  Every record whose type has a row in EVENT_INDEX becomes a point; events of other records
    (beats, timing points ...) map to the NEXT point, trailing ones to the last.  Kept quirks: a coordinate of 0 counts
    as "no position" and moves the point to the playfield centre (`not group.x`), a distance of 0 is re-derived from
    the previous point; an empty stream returns `(zeros(2,0), zeros(1,0), zeros(1,0), 0, {}, [])`.
    `timing`: objects with `.offset` (timedelta), `.ms_per_beat`, `.parent` (slider.TimingPoint); `types_first` /
    `has_sv` are `args.train.data.types_first` / `.add_sv`."""
    from datetime import timedelta
    times = event_times(events, types_first=types_first)
    records, members = group_events(events, times, types_first=types_first)
    seq_indices, waiting = {}, []
    px, py, pt, pd, prow = [], [], [], [], []
    head_time = last_anchor_time = 0
    prev = (256, 192)
    for rec, idx in zip(records, members):
        waiting.extend(idx)
        row = EVENT_INDEX.get(rec.type_name)
        if row is None:
            continue
        if rec.new_combo and rec.type_name in ("CIRCLE", "SLIDER_HEAD"):
            row += 1
        if rec.type_name == "SLIDER_END":
            span, total = last_anchor_time - head_time, rec.time - head_time
            row += repeat_type(max(int(round(total / span)), 1) if span > 0 else 1)
        elif rec.type_name == "SLIDER_HEAD":
            head_time = rec.time
        elif rec.type_name == "LAST_ANCHOR":
            last_anchor_time = rec.time
        if not rec.x or not rec.y:
            rec.x, rec.y = 256, 192
        if not rec.distance:
            rec.distance = ((rec.x - prev[0]) ** 2 + (rec.y - prev[1]) ** 2) ** 0.5
        px.append(rec.x), py.append(rec.y), pt.append(rec.time), pd.append(rec.distance), prow.append(row)
        for j in waiting:
            seq_indices[j] = len(px) - 1
        waiting = []
        prev = (rec.x, rec.y)
    for j in waiting:
        seq_indices[j] = len(px) - 1
    if not px:
        return torch.zeros(2, 0), torch.zeros(1, 0), torch.zeros(1, 0), 0, {}, []
    seq_x, seq_o, seq_c = points_to_sequence(px, py, pt, pd, prow)

    sliders = []
    if has_sv and timing is not None:
        head = last = None
        path = []                                   # (curve name, point) of head, anchors (a red anchor twice), last anchor
        for rec, idx in zip(records, members):
            point = None if rec.type_name not in _CURVE_OF_ANCHOR and rec.type_name != "SLIDER_HEAD" else seq_indices[idx[0]]
            if rec.type_name == "SLIDER_HEAD":
                head, last, path = rec, None, [("Bezier", point)]
            elif rec.type_name in _CURVE_OF_ANCHOR:
                path += [(_CURVE_OF_ANCHOR[rec.type_name], point)] * (2 if rec.type_name == "RED_ANCHOR" else 1)
                if rec.type_name == "LAST_ANCHOR":
                    last = rec
            elif rec.type_name == "SLIDER_END" and head is not None and last is not None:
                tp = timing_point_at(timedelta(milliseconds=int(round(head.time))), timing)
                redline = tp if tp.parent is None else tp.parent
                if head.scroll_speed is not None:
                    length = head.scroll_speed * (last.time - head.time) * 100 / redline.ms_per_beat * slider_multiplier
                    sliders.append(DiffusionSlider(np.array([pt_ for _, pt_ in path]), seq_indices[idx[0]], path[1][0], length))
                head, last, path = None, None, []
    return seq_x, seq_o, seq_c, len(px), seq_indices, sliders


def events_with_pos(events: Sequence, positions: torch.Tensor, seq_indices: dict) -> list:
    """(diffusion_pipeline.py:447-469) positions (2, T) in playfield pixels -> the event stream with every DISTANCE
    event replaced by POS_X, POS_Y and every POS_X / POS_Y refreshed, rounded half-to-even like `int(round(.))`.  The new
    events are of the class of the events handed in (the reference's or ours)."""
    if not events:
        return []
    Ev, ET = type(events[0]), type(events[0].type)
    xy = positions.detach().cpu().tolist()
    out = []
    for i, e in enumerate(events):
        nm = e.type.name
        if nm == "DISTANCE":
            out += [Ev(ET["POS_X"], int(round(xy[0][seq_indices[i]]))), Ev(ET["POS_Y"], int(round(xy[1][seq_indices[i]])))]
        elif nm in ("POS_X", "POS_Y"):
            out.append(Ev(ET[nm], int(round(xy[0 if nm == "POS_X" else 1][seq_indices[i]]))))
        else:
            out.append(e)
    return out


class DiffusionTokenizer:
    """The class vocabulary of the diffusion model: `osu_diffusion.utils.tokenizer.Tokenizer` (tokenizer.py:11-118, 216-250)
    restated from its saved state (`tokenizer.pkl` beside the checkpoint is that state dict, inference.py:626-635), so
    the stage runs without the reference package.  Token layout: [styles | difficulties | mappers | descriptors | circle
    sizes], the last id of every family = "unknown".  Kept as they are: `encode_mapper` takes a BEATMAP id and looks its
    mapper up (the pipeline hands it `mapper_id`, diffusion_pipeline.py:84); an unknown descriptor name encodes one PAST the
    family's unknown id."""
    _FAMILIES = ("num_classes", "num_diff_classes", "num_mapper_classes", "num_descriptor_classes", "num_cs_classes")
    _TABLES = ("beatmap_idx", "beatmap_mapper", "mapper_idx", "beatmap_descriptors", "descriptor_idx")

    def __init__(self, state_dict: Optional[dict] = None):
        for k in self._FAMILIES:
            setattr(self, k, 0)
        for k in self._TABLES:
            setattr(self, k, {})
        self.max_difficulty = 0
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict: dict) -> None:
        for k in ("beatmap_idx", "num_classes", "num_diff_classes", "max_difficulty"):       # required (tokenizer.py:232-235)
            setattr(self, k, state_dict[k])
        for k in self._FAMILIES[2:] + self._TABLES[1:]:
            if k in state_dict:
                setattr(self, k, state_dict[k])

    def state_dict(self) -> dict:
        return {k: getattr(self, k) for k in self._FAMILIES + self._TABLES + ("max_difficulty",)}

    def _base(self, family: int) -> int:
        return sum(getattr(self, k) for k in self._FAMILIES[:family])

    def _unk(self, family: int) -> int:
        return self._base(family + 1) - 1

    @staticmethod
    def _bucket(value: float, n: int, full_scale: float) -> int:
        return min(max(int(value * (n - 2) / full_scale), 0), n - 2)

    num_tokens = property(lambda self: self._base(5))
    style_unk = property(lambda self: self._unk(0))
    diff_unk = property(lambda self: self._unk(1))
    mapper_unk = property(lambda self: self._unk(2))
    descriptor_unk = property(lambda self: self._unk(3))
    cs_unk = property(lambda self: self._unk(4))

    def encode_style(self, beatmap_id: int) -> int:
        return self.beatmap_idx.get(beatmap_id, self.style_unk)

    def encode_diff(self, diff: float) -> int:
        return self._base(1) + self._bucket(diff, self.num_diff_classes, self.max_difficulty)

    def encode_mapper_id(self, user_id: int) -> int:
        return self._base(2) + self.mapper_idx.get(user_id, self.num_mapper_classes - 1)

    def encode_mapper(self, beatmap_id: int) -> int:
        return self.encode_mapper_id(self.beatmap_mapper.get(beatmap_id, -1))

    def encode_descriptor_idx(self, descriptor_idx: int) -> int:
        return self._base(3) + descriptor_idx

    def encode_descriptor_name(self, descriptor: str) -> int:
        return self.encode_descriptor_idx(self.descriptor_idx.get(descriptor, self.num_descriptor_classes))

    def encode_descriptor(self, beatmap_id: int) -> list:
        return [self.encode_descriptor_idx(i)
                for i in self.beatmap_descriptors.get(beatmap_id, [self.num_descriptor_classes - 1])]

    def encode_cs(self, cs: float) -> int:
        return self._base(4) + self._bucket(cs, self.num_cs_classes, 10)


def get_class_vector(tokenizer, config) -> torch.Tensor:
    """(diffusion_pipeline.py:65-109) multi-hot class vector of a GenerationConfig over the osu_diffusion Tokenizer's
    classes (the tokenizer object is the one pickled beside the checkpoint; only its attributes are read).  An
    unknown / missing value selects the family's `*_unk` class; the messages of the reference are kept."""
    v = torch.zeros(tokenizer.num_tokens)
    families = (("num_classes", "beatmap_id", "encode_style", "style_unk", "beatmap_idx", "Beatmap"),
                ("num_diff_classes", "difficulty", "encode_diff", "diff_unk", None, None),
                ("num_mapper_classes", "mapper_id", "encode_mapper", "mapper_unk", "mapper_idx", "Mapper"),
                ("num_cs_classes", "circle_size", "encode_cs", "cs_unk", None, None))
    for count, field, encode, unk, known, label in families:
        if getattr(tokenizer, count) <= 0:
            continue
        value = getattr(config, field, None)
        if value is None:
            v[getattr(tokenizer, unk)] = 1
            continue
        v[getattr(tokenizer, encode)(value)] = 1
        if known is not None and value not in getattr(tokenizer, known):
            print(f"{label} class {value} not found. Using default.")
    if tokenizer.num_descriptor_classes > 0:
        names = getattr(config, "descriptors", None)
        found = [d for d in names or () if d in tokenizer.descriptor_idx]
        if names and not found:
            print("Descriptor classes not found. Using default.")
        for d in names or ():
            if found and d not in tokenizer.descriptor_idx:
                print(f"Descriptor class {d} not found. Skipping.")
        if found:
            v[[tokenizer.encode_descriptor_name(d) for d in found]] = 1
        else:
            v[tokenizer.descriptor_unk] = 1
    return v


@dataclass
class DiffusionGenerationConfig:
    """The fields of the reference's GenerationConfig (osuT5/osuT5/inference/processor.py:26-43) this stage reads; any
    object with these attributes (the reference's own included) is accepted by `generate`."""
    beatmap_id: Optional[int] = None
    difficulty: Optional[float] = None
    mapper_id: Optional[int] = None
    circle_size: Optional[float] = None
    slider_multiplier: float = 1.4
    descriptors: Optional[Sequence] = None
    negative_descriptors: Optional[Sequence] = None


class DiffusionPipelineHIP:
    """Same knobs as the reference object (diffusion_pipeline.py:40-63 / config.py:98-106)."""

    def __init__(self, model: DiTHIP, *, timesteps, diffusion_steps: int = 1000, noise_schedule: str = "squaredcos_cap_v2",
                 seq_len: int = 128, max_seq_len: int = 1024, overlap_buffer: int = 128, cfg_scale: float = 1.0,
                 refine_model: Optional[DiTHIP] = None, refine_iters: int = 10, random_init: bool = False,
                 pad_sequence: bool = False, start_time: Optional[float] = None, end_time: Optional[float] = None,
                 tokenizer=None, types_first: bool = False, has_sv: bool = True):
        # pad_sequence (reference diffusion_pipeline.py:186-193) pads every window to max_seq_len with zero positions and zero
        # context, pads the band mask with "allowed" and builds a key_padding_mask -- which DiTBlock.forward never hands to
        # its attention (models.py:133-150): every real query then attends all the pad tokens, so padding CHANGES the real
        # positions (tests/test_oracle_pinned.py::test_reference_dit_padding_changes_real_positions).  Reproduced as it is:
        # BandMask(open_from=) / the attention kernels' `open_from`.
        if not 0 <= 2 * overlap_buffer < max_seq_len:
            raise ValueError("overlap_buffer must be less than half of max_seq_len")
        self.model, self.refine_model = model, refine_model
        self.device = model.device
        self.timesteps, self.diffusion_steps, self.noise_schedule = timesteps, diffusion_steps, noise_schedule
        self.seq_len, self.max_seq_len, self.overlap_buffer = seq_len, max_seq_len, overlap_buffer
        self.cfg_scale, self.refine_iters, self.random_init = cfg_scale, refine_iters, random_init
        self.start_time, self.end_time, self.pad_sequence = start_time, end_time, bool(pad_sequence)
        # what `generate` needs on top: the osu_diffusion Tokenizer pickled beside the checkpoint (class vectors) and the two
        # stream-format flags of the T5 stage (`args.train.data.types_first` / `.add_sv`, diffusion_pipeline.py:58-62)
        self.tokenizer, self.types_first, self.has_sv = tokenizer, bool(types_first), bool(has_sv)

    @torch.no_grad()
    def generate(self, events: Sequence, generation_config, timing: Optional[Sequence], verbose: bool = False,
                 noise_source: Optional[Callable] = None) -> list:
        """The reference's `DiffisionPipeline.generate` (diffusion_pipeline.py:111-287), same arguments and result:
        events with DISTANCE events -> events with POS_X / POS_Y events.  `generation_config` is read for
        `slider_multiplier`, the class fields of `get_class_vector`, and `negative_descriptors` (the null-class row of
        the guidance pair keeps difficulty and circle size, :151-155)."""
        seq_x, seq_o, seq_c, n, seq_indices, sliders = events_to_sequence(
            events, timing, generation_config.slider_multiplier, types_first=self.types_first, has_sv=self.has_sv)
        if verbose:
            print(f"seq len {n}")
        if n == 0:
            return events
        if self.tokenizer is None:
            raise ValueError("DiffusionPipelineHIP.generate needs the osu_diffusion tokenizer (class vectors)")
        null_config = DiffusionGenerationConfig(difficulty=getattr(generation_config, "difficulty", None),
                                       descriptors=getattr(generation_config, "negative_descriptors", None),
                                       circle_size=getattr(generation_config, "circle_size", None))
        positions = self.generate_positions(seq_x, seq_o, seq_c, get_class_vector(self.tokenizer, generation_config),
                                            get_class_vector(self.tokenizer, null_config), noise_source=noise_source,
                                            sliders=sliders)
        return events_with_pos(events, positions.squeeze(0), seq_indices)

    def to_positions(self, samples: torch.Tensor) -> torch.Tensor:
        """(:171-176) drop the null-class half, [-1, 1] -> playfield pixels, to the CPU.  (2B, 2, T) -> (B, 2, T)"""
        samples, _ = samples.clone().chunk(2, dim=0)
        samples += 1
        samples /= 2
        samples *= torch.tensor(PLAYFIELD, device=samples.device).repeat(1, 1).unsqueeze(2)
        return samples.cpu()

    @torch.no_grad()
    def generate_positions(self, seq_x: torch.Tensor, seq_o: torch.Tensor, seq_c: torch.Tensor,
                           class_vector: torch.Tensor, unk_class_vector: torch.Tensor,
                           noise_source: Optional[Callable] = None,
                           denoised_fn_factory: Optional[Callable] = None,
                           sliders: Optional[Sequence] = None) -> torch.Tensor:
        """seq_* as returned by `events_to_sequence`; class vectors (C,) multi-hot.  Returns positions (1, 2, T) on the
        CPU, what the reference hands to `events_with_pos`.  (= generate_positions_batch with one chunk.)

        sliders: the DiffusionSlider list `events_to_sequence` returns (its 6th value); their end points are
            re-projected onto the slider paths every denoising step, on the device.
        noise_source(n, shape) -> fp32 [n, *shape]: the gaussian noise of n consecutive p_sample calls (parity tests
            inject the reference's draws); default: torch.randn on the device, one draw per call like `th.randn_like`.
        denoised_fn_factory(mask, z_part, start, end) -> callable: replaces the in-paint-only `denoised_fn` (e.g. with
            the slider re-projection); forces the per-step host round trip."""
        if seq_x.shape[1] == 0:
            return torch.zeros(1, 2, 0)
        return self.generate_positions_batch(seq_x[None], seq_o[None], seq_c[None], class_vector[None],
                                             unk_class_vector[None], noise_source, denoised_fn_factory,
                                             None if sliders is None else [list(sliders)])

    @torch.no_grad()
    def generate_positions_batch(self, seq_x: torch.Tensor, seq_o: torch.Tensor, seq_c: torch.Tensor,
                                 class_vectors: torch.Tensor, unk_class_vectors: torch.Tensor,
                                 noise_source: Optional[Callable] = None,
                                 denoised_fn_factory: Optional[Callable] = None,
                                 sliders: Optional[Sequence[Sequence]] = None) -> torch.Tensor:
        """B song-chunks with the SAME number of points T through the diffusion stage as ONE denoiser batch:
        seq_x (B, 2, T), seq_o (B, T), seq_c (B, 272, T), class vectors (B, C).  Returns positions (B, 2, T) on the CPU.
        `sliders`: one DiffusionSlider list per chunk (indices count the chunk's own points).

        The reference runs its pipeline once per beatmap with `n = 1` (diffusion_pipeline.py:157-166): a CFG batch of
        2 rows, i.e. M = 2 T rows per GEMM -- 67 latency-bound launches per step at T = 128.  Chunks are independent,
        so the B conditional rows and the B null-class rows are stacked as [cond_0..cond_{B-1} | null_0..null_{B-1}]
        (the layout `forward_with_cfg` splits in halves, models.py:306-317): M = 2 B T rows reach the 128x128 MFMA
        tiles and every launch is shared by all chunks.  Row b of the result equals the single-chunk run of chunk b
        (bit for bit when both use the same GEMM tile family: option gemm_splitk_tiles = 0).
        `noise_source(n, shape)` is asked for shape (2B, 2, T_window): rows b and B + b are chunk b's pair."""
        dev = self.device
        B, _, seq_len = seq_x.shape
        if seq_len == 0:
            return torch.zeros(B, 2, 0)
        if sliders is not None and len(sliders) != B:
            raise ValueError(f"{len(sliders)} slider lists for {B} chunks")
        diffusion = create_diffusion(timestep_respacing=self.timesteps, diffusion_steps=self.diffusion_steps,
                                     noise_schedule=self.noise_schedule)
        z = seq_x.to(dev, torch.float32)
        c = seq_c.to(dev, torch.float32)
        y = class_vectors.to(dev, torch.float32)
        y_null = unk_class_vectors.to(dev, torch.float32)
        z = torch.cat([z, z], 0)
        c = torch.cat([c, c], 0)
        y = torch.cat([y, y_null], 0)
        if self.random_init:
            z = torch.randn(*z.shape, device=dev)
        seq_o = seq_o.to(torch.float32).cpu()
        if noise_source is None:
            def noise_source(k, shape):
                return torch.stack([torch.randn(*shape, device=dev) for _ in range(k)])

        def sample_part(zfull, start, end, start_mask_size=0):
            z_part = zfull[:, :, start:end].contiguous()
            c_part = c[:, :, start:end].contiguous()
            real = end - start
            pad = self.max_seq_len - real if self.pad_sequence else 0          # (:186-193)
            if pad > 0:
                z_part = torch.nn.functional.pad(z_part, (0, pad)).contiguous()
                c_part = torch.nn.functional.pad(c_part, (0, pad)).contiguous()
            T = real + max(pad, 0)
            # True means it will be generated (:223-234); per chunk, the pair of a chunk shares its mask
            mask = torch.full(z_part.shape, False, dtype=torch.bool, device=dev)
            mask[:, :, start_mask_size:] = True
            for b in range(B):
                o_part = seq_o[b, start:end].contiguous()
                if self.start_time is not None:
                    k0 = int(torch.searchsorted(o_part, self.start_time, right=False))
                    mask[b, :, :k0] = False
                    mask[B + b, :, :k0] = False
                if self.end_time is not None:
                    k1 = int(torch.searchsorted(o_part, self.end_time, right=True))
                    mask[b, :, k1:] = False
                    mask[B + b, :, k1:] = False
            if not bool(mask.any()):
                return z_part[:, :, :real]
            if denoised_fn_factory is not None:
                denoised_fn = denoised_fn_factory(mask, z_part, start, end)
            elif sliders is not None and any(len(sl) > 0 for sl in sliders):
                denoised_fn = SliderInpaintSpec(mask, z_part, sliders, start, end)
            else:
                denoised_fn = InpaintSpec(mask, z_part)
            z_part = denoised_fn(z_part)
            model_kwargs = dict(c=c_part, y=y, cfg_scale=self.cfg_scale,
                                attn_mask=BandMask(T, self.seq_len, open_from=real if pad > 0 else 0), key_padding_mask=None)
            samples = diffusion.p_sample_loop(self.model.forward_with_cfg, z_part.shape, z_part, denoised_fn=denoised_fn,
                                              clip_denoised=True, model_kwargs=model_kwargs, device=dev,
                                              step_noise=noise_source(diffusion.num_timesteps, tuple(z_part.shape)))
            if self.refine_model is not None:
                # the reference refines with `self.model.forward_with_cfg` (:261), not the refine model: kept as is
                for _ in range(self.refine_iters):
                    t = torch.tensor([0] * samples.shape[0], device=dev)
                    out = diffusion.p_sample(self.model.forward_with_cfg, samples, t, denoised_fn=denoised_fn,
                                             clip_denoised=True, model_kwargs=model_kwargs,
                                             noise=noise_source(1, tuple(samples.shape))[0])
                    samples = out["sample"]
            return samples[:, :, :real] if pad > 0 else samples

        full = z.clone()
        ob = self.overlap_buffer
        for i in range(0, seq_len - ob * 2, self.max_seq_len - ob * 2):
            end = min(i + self.max_seq_len, seq_len)
            if i > 0:
                # the first buffer is done; the second was generated but is regenerated from the initial values (:279-282)
                full[:, :, i + ob:i + ob * 2] = z[:, :, i + ob:i + ob * 2]
            full[:, :, i:end] = sample_part(full, i, end, start_mask_size=ob if i > 0 else 0)
        return self.to_positions(full)
