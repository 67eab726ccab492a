"""Fixed-vocabulary tokenizer with the reference's `Tokenizer` read API
(osuT5/osuT5/tokenizer.py:47-290: cumulative ranges -> ids, `encode`/`decode`, `event_start`/
`event_end`, `vocab_size_in/out`, pad/sos/eos ids, context sos/eos, JSON state).

The decode boundary only needs id ranges (EOS set, TIME_SHIFT range, SOS ids), so this mirror is
built either from the reference's own `tokenizer.json` state (`load_state_dict`) or from an
explicit range list (`from_ranges`); dataset-driven construction (mapper/descriptor tables, MMRS
metadata) stays with the reference and is out of scope.
Any object exposing the same attributes (e.g. the reference's own Tokenizer) can be passed to
`model_generate` instead.
"""
from __future__ import annotations

import json
from typing import Iterable

from .event import ContextType, Event, EventRange, EventType

MILISECONDS_PER_SECOND = 1000
MILISECONDS_PER_STEP = 10

# event types every reference vocabulary ends with (tokenizer.py:179-195)
_TAIL = ([(EventType.NEW_COMBO, 0, 0), (EventType.HITSOUND, 0, 2 ** 3 * 3 * 3), (EventType.VOLUME, 0, 100)] +
         [(t, 0, 0) for t in (EventType.CIRCLE, EventType.SPINNER, EventType.SPINNER_END, EventType.SLIDER_HEAD,
                              EventType.BEZIER_ANCHOR, EventType.PERFECT_ANCHOR, EventType.CATMULL_ANCHOR,
                              EventType.RED_ANCHOR, EventType.LAST_ANCHOR, EventType.SLIDER_END, EventType.BEAT,
                              EventType.MEASURE)])


class Tokenizer:
    def __init__(self):
        self.offset = 3
        self.context_sos: dict[ContextType, int] = {}
        self.context_eos: dict[ContextType, int] = {}
        self.event_ranges: list[EventRange] = []
        self.input_event_ranges: list[EventRange] = []
        self.num_classes = 0
        self.num_diff_classes = 0
        self.max_difficulty = 0
        self.num_mapper_classes = 0
        self.num_descriptor_classes = 0
        self.num_cs_classes = 0
        self._index()

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def from_ranges(cls, event_ranges: Iterable[tuple], input_event_ranges: Iterable[tuple] = (),
                    context_types: Iterable[ContextType] = ()):
        tok = cls()
        for ct in context_types:
            ct = ContextType(ct)
            if ct not in tok.context_sos:
                tok.context_sos[ct] = tok.offset
                tok.context_eos[ct] = tok.offset + 1
                tok.offset += 2
        tok.event_ranges = [EventRange(EventType(t) if not isinstance(t, EventType) else t, lo, hi)
                            for t, lo, hi in event_ranges]
        tok.input_event_ranges = [EventRange(EventType(t) if not isinstance(t, EventType) else t, lo, hi)
                                  for t, lo, hi in input_event_ranges]
        tok._index()
        return tok

    @classmethod
    def benchmark_vocab(cls, src_seq_len: int = 1251, hop_length: int = 128, sample_rate: int = 16000):
        """The minimal osu!standard vocabulary of the BASELINE configs (TIME_SHIFT 0..max,
        SNAPPING, DISTANCE + the fixed tail; no context tokens) -- 1849 ids for a 10 s window."""
        ms = (src_seq_len - 1) * hop_length * MILISECONDS_PER_SECOND / sample_rate
        max_ts = int(ms / MILISECONDS_PER_STEP)
        return cls.from_ranges([(EventType.TIME_SHIFT, 0, max_ts), (EventType.SNAPPING, 0, 16),
                                (EventType.DISTANCE, 0, 640)] + _TAIL)

    def _index(self):
        self.event_range = {er.type: er for er in self.event_ranges} | {er.type: er for er in self.input_event_ranges}
        self.event_start, self.event_end = {}, {}
        off = self.offset
        for er in list(self.event_ranges) + list(self.input_event_ranges):
            self.event_start[er.type] = off
            off += er.max_value - er.min_value + 1
            self.event_end[er.type] = off
        self.vocab_size_out = self.offset + sum(er.max_value - er.min_value + 1 for er in self.event_ranges)
        self.vocab_size_in = self.vocab_size_out + sum(er.max_value - er.min_value + 1
                                                       for er in self.input_event_ranges)

    # ---- reference read API -------------------------------------------------------------------
    @property
    def pad_id(self) -> int:
        return 0

    @property
    def sos_id(self) -> int:
        return 1

    @property
    def eos_id(self) -> int:
        return 2

    def decode(self, token_id: int) -> Event:
        off = self.offset
        for er in list(self.event_ranges) + list(self.input_event_ranges):
            n = er.max_value - er.min_value + 1
            if off <= token_id < off + n:
                return Event(type=er.type, value=er.min_value + token_id - off)
            off += n
        raise ValueError(f"id {token_id} is not mapped to any event")

    def encode(self, event: Event) -> int:
        if event.type not in self.event_range:
            raise ValueError(f"unknown event type: {event.type}")
        er = self.event_range[event.type]
        if not er.min_value <= event.value <= er.max_value:
            raise ValueError(f"event value {event.value} is not within range [{er.min_value}, {er.max_value}] "
                             f"for event type {event.type}")
        return self.event_start[event.type] + event.value - er.min_value

    def event_type_range(self, event_type: EventType) -> tuple[int, int]:
        if event_type not in self.event_range:
            raise ValueError(f"unknown event type: {event_type}")
        er = self.event_range[event_type]
        s = self.event_start[event_type]
        return s, s + (er.max_value - er.min_value)

    # ---- JSON state (same keys as the reference's tokenizer.json) ------------------------------------
    @staticmethod
    def _er_state(er: EventRange):
        return {"type": er.type.value, "min_value": er.min_value, "max_value": er.max_value}

    def state_dict(self):
        return {
            "offset": self.offset,
            "context_sos": {k.value: v for k, v in self.context_sos.items()},
            "context_eos": {k.value: v for k, v in self.context_eos.items()},
            "event_ranges": [self._er_state(er) for er in self.event_ranges],
            "input_event_ranges": [self._er_state(er) for er in self.input_event_ranges],
            "num_classes": self.num_classes, "num_diff_classes": self.num_diff_classes,
            "max_difficulty": self.max_difficulty,
            "event_range": {k.value: self._er_state(v) for k, v in self.event_range.items()},
            "event_start": {k.value: v for k, v in self.event_start.items()},
            "event_end": {k.value: v for k, v in self.event_end.items()},
            "vocab_size_out": self.vocab_size_out, "vocab_size_in": self.vocab_size_in,
            "num_mapper_classes": self.num_mapper_classes,
            "num_descriptor_classes": self.num_descriptor_classes, "num_cs_classes": self.num_cs_classes,
        }

    def load_state_dict(self, sd: dict):
        self.offset = sd.get("offset", 3)
        self.context_sos = {ContextType(k): v for k, v in sd.get("context_sos", {}).items()}
        self.context_eos = {ContextType(k): v for k, v in sd.get("context_eos", {}).items()}

        def er(d):
            return EventRange(EventType(d["type"]), d["min_value"], d["max_value"])

        self.event_ranges = [er(d) for d in sd.get("event_ranges", [])]
        self.input_event_ranges = [er(d) for d in sd.get("input_event_ranges", [])]
        for k in ("num_classes", "num_diff_classes", "max_difficulty", "num_mapper_classes",
                  "num_descriptor_classes", "num_cs_classes"):
            setattr(self, k, sd.get(k, 0))
        self._index()
        for key in ("vocab_size_out", "vocab_size_in"):
            if key in sd and sd[key] != getattr(self, key):
                raise ValueError(f"tokenizer state is inconsistent: {key}={sd[key]} but ranges give "
                                 f"{getattr(self, key)}")
        return self

    @classmethod
    def from_json(cls, path: str):
        with open(path, encoding="utf-8") as f:
            return cls().load_state_dict(json.load(f))

    def save_json(self, path: str):
        with open(path, "w", encoding="utf-8") as f:
            json.dump(self.state_dict(), f, ensure_ascii=False)


# ---- ids <-> Events with absolute times (reference Processor._encode / _decode, processor.py:1215-1268) --------
def encode_events(tokenizer: "Tokenizer", events, frame_time: float):
    """Events with ABSOLUTE TIME_SHIFT values (ms) -> int64 ids (1, n) relative to the window start `frame_time`:
    `int((ms - frame_time) / 10)` truncated toward zero and clipped to the TIME_SHIFT range (processor.py:1215-1224)."""
    import numpy as np
    import torch
    tokens = torch.empty((1, len(events)), dtype=torch.long)
    ts_range = tokenizer.event_range[EventType.TIME_SHIFT]
    for i, event in enumerate(events):
        if event.type == EventType.TIME_SHIFT:
            value = int((event.value - frame_time) / MILISECONDS_PER_STEP)
            value = int(np.clip(value, ts_range.min_value, ts_range.max_value))
            event = Event(type=event.type, value=value)
        tokens[0, i] = tokenizer.encode(event)
    return tokens


def decode_tokens(tokenizer: "Tokenizer", tokens, frame_time: float, allow_non_events: bool = False):
    """ids -> Events, TIME_SHIFT steps back to absolute ms: `frame_time + steps * 10 + 5` for steps >= 0 (the half-step
    de-bias of the truncating encoder), stops at eos unless `allow_non_events`, ids outside the vocabulary are skipped
    or kept as CONTROL events (processor.py:1226-1268)."""
    events = []
    for token in tokens:
        tid = int(token)
        if tid == tokenizer.eos_id and not allow_non_events:
            break
        try:
            event = tokenizer.decode(tid)
        except ValueError:
            if allow_non_events:
                events.append(Event(EventType.CONTROL, tid))
            continue
        if event.type == EventType.TIME_SHIFT:
            half_step = MILISECONDS_PER_STEP // 2 if event.value >= 0 else 0
            event.value = frame_time + event.value * MILISECONDS_PER_STEP + half_step
        events.append(event)
    return events
