"""not gpu: the N>1 path (chunk sharding + all_gather of token streams) on 2 gloo ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mapperatorinator_amd.sharding import shard_bounds


def test_shard_bounds_cover_everything():
    for n in (0, 1, 5, 32, 255, 256):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _fake_generate(shard):
    """stand-in for model_generate: token stream is a deterministic function of the chunk's audio"""
    x = shard["inputs"]
    n = x.shape[0]
    lens = (x[:, 0].abs() * 10).long() % 7 + 2
    width = int(lens.max().item()) if n else 1
    out = torch.zeros((n, width), dtype=torch.long)
    for i in range(n):
        out[i, : lens[i]] = (x[i, : lens[i]] * 1000).long().abs() % 1800 + 3
    return out, {"generated_tokens": int(lens.sum())}


def _worker(rank, world, port, B, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mapperatorinator_amd.sharding import sharded_generate
        g = torch.Generator().manual_seed(0)
        audio = torch.randn(B, 64, generator=g)
        toks, lens, stats = sharded_generate(_fake_generate, dict(inputs=audio, flag=3), pad_id=0, max_length=12)
        ret[rank] = (toks, lens, stats)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 8, 1])
def test_two_rank_gather_equals_single_process(B):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, B, ret), nprocs=2, join=True)
    g = torch.Generator().manual_seed(0)
    audio = torch.randn(B, 64, generator=g)
    want, _ = _fake_generate(dict(inputs=audio))
    for r in range(2):
        toks, lens, _ = ret[r]
        assert toks.shape == (B, 12)
        for b in range(B):
            n = int(lens[b])
            assert n <= 12
        full = torch.zeros((B, 12), dtype=torch.long)
        # each shard pads to its own width; compare the non-pad prefix row by row
        for b in range(B):
            row = want[b][want[b] != 0]
            assert torch.equal(toks[b][: len(row)], row) and (toks[b][len(row):] == 0).all()
    assert torch.equal(ret[0][0], ret[1][0])
