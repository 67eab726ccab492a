"""Chunk-parallel sharding of the hot path over the GPUs of one node.

The path shards naturally: `generate_parallel` windows / songs are independent units
(osuT5/osuT5/inference/processor.py:370-419; preprocessor.py:20-21), every rank holds a full
replica of the weights and runs mel -> encode -> decode (-> DiT) on its block of chunks with NO
data-path collective.  The only exchange is one all_gather of the finished token streams
(B_local x max_length int32, <= 256 KB per rank: latency-bound over xGMI) so that rank 0 -- or
every rank -- can continue with the unchanged host loop.  Backend "nccl" is RCCL on ROCm; the same
code runs on "gloo" with CPU tensors (tests/test_sharding_gloo.py, world_size 2).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition; the first `n_items % world` ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_ragged(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """local: (n_local, W) on the backend's device; returns (n_total, W) with rows in global order.
    Shards are padded to the largest shard so one fixed-size all_gather suffices."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_total, rank, world)
    assert local.shape[0] == hi - lo, f"rank {rank}: expected {hi - lo} rows, got {local.shape[0]}"
    max_rows = -(-n_total // world)
    buf = local.new_zeros((max_rows,) + tuple(local.shape[1:]))
    buf[: local.shape[0]] = local
    out = local.new_empty((world * max_rows,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    parts = []
    for r in range(world):
        a, b = shard_bounds(n_total, r, world)
        parts.append(out[r * max_rows: r * max_rows + (b - a)])
    return torch.cat(parts, 0)


def sharded_generate(generate_fn: Callable, model_kwargs: dict, pad_id: int, max_length: int, group=None,
                     comm_device: Optional[torch.device] = None, refine_fn: Optional[Callable] = None):
    """Run `generate_fn(shard_of_model_kwargs) -> (LongTensor[n_local, <=max_length] (CPU), stats)` on this
    rank's block of chunks and all_gather the padded token streams.

    refine_fn(shard_of_model_kwargs, tokens) -> fp32 (n_local, 2, Tq): the diffusion stage of the rank's chunks
    (hit-object coordinates, SURVEY.md 8e "coords (32, 2, Tq) fp32").  Tq must be the same on every rank; the
    coordinates travel in the SAME all_gather as the tokens (bit-cast to int32 columns behind them), so a stage still
    costs one latency-bound collective.

    Sampling under sharding: a shard is its own batch, so (i) pass `conditional_temperature_per_row=True` in the
    generate kwargs when the ConditionalTemperature processor is on (the reference's batch form reads GLOBAL row 0,
    which other ranks do not hold), and (ii) `rng_row_offset` = the shard's first global row (model_kwargs key
    `_row_offset` is filled in here for generate_fn to forward) keeps the draws of different shards distinct.

    Returns (tokens int64 CPU (B, max_length) padded with pad_id, lengths int64 (B,), local_stats) and, with
    refine_fn, a 4th element: coords fp32 CPU (B, 2, Tq) in global chunk order."""
    grouped = dist.is_initialized()      # a formed group is USED even at world 1 (the RCCL path of a 1-GPU box = the N-GPU path)
    world = dist.get_world_size(group) if grouped else 1
    rank = dist.get_rank(group) if grouped else 0
    B = model_kwargs["inputs"].shape[0]
    lo, hi = shard_bounds(B, rank, world)
    shard = {k: (v[lo:hi] if isinstance(v, torch.Tensor) and v.shape[:1] == (B,) else v)
             for k, v in model_kwargs.items()}
    shard["_row_offset"] = lo
    if hi > lo:
        toks, stats = generate_fn(shard)
    else:
        toks, stats = torch.zeros((0, 1), dtype=torch.long), {"generated_tokens": 0}
    if toks.shape[1] > max_length:
        raise ValueError(f"generate_fn returned {toks.shape[1]} columns, max_length is {max_length}")
    # everything that travels is assembled ON the collective's device (RCCL: the GPU; gloo: the CPU): a generate_fn /
    # refine_fn that returns device tensors is never bounced through the host before the gather
    dev = _comm_device(group, comm_device) if grouped else toks.device
    local = torch.full((hi - lo, max_length + 1), pad_id, dtype=torch.int32, device=dev)
    local[:, : toks.shape[1]] = toks.to(device=dev, dtype=torch.int32)
    local[:, max_length] = toks.shape[1]          # column max_length carries the produced length
    n_coord = 0
    coord_shape = None
    if refine_fn is not None:
        coords = refine_fn(shard, toks) if hi > lo else None
        # every rank must agree on Tq even when its shard is empty: settle it with the first non-empty shard's shape
        shape_t = torch.tensor(list(coords.shape[1:]) if coords is not None else [0, 0], dtype=torch.int64, device=dev)
        if grouped:
            shapes = [torch.zeros_like(shape_t) for _ in range(world)]      # outputs live where the input lives
            dist.all_gather(shapes, shape_t, group=group)
            shape_t = max((s_.cpu() for s_ in shapes), key=lambda s_: int(s_.prod()))
        coord_shape = tuple(int(v) for v in shape_t.cpu())
        n_coord = coord_shape[0] * coord_shape[1]
        packed = torch.zeros((hi - lo, n_coord), dtype=torch.int32, device=dev)
        if coords is not None:
            if tuple(coords.shape[1:]) != coord_shape:
                raise ValueError(f"refine_fn returned {tuple(coords.shape)}; every rank must use the same (2, Tq) = {coord_shape}")
            packed = coords.to(device=dev, dtype=torch.float32).contiguous().reshape(hi - lo, n_coord).view(torch.int32)
        local = torch.cat([local, packed], 1)
    if not grouped:
        full = local.cpu()
    else:
        full = all_gather_ragged(local, B, group).cpu()
    out = (full[:, :max_length].to(torch.int64), full[:, max_length].to(torch.int64), stats)
    if refine_fn is not None:
        c = full[:, max_length + 1:].contiguous().view(torch.float32).reshape((B,) + coord_shape)
        out = out + (c,)
    return out


def _comm_device(group, comm_device):
    return comm_device or (torch.device("cuda", torch.cuda.current_device())
                           if dist.get_backend(group) == "nccl" else torch.device("cpu"))
