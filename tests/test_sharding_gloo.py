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


MAXLEN, TQ = 12, 5


def _standin_generate(shard):
    """CPU stand-in for `model_generate` with its full contract: left-padded ragged prompts are kept in the output, rows
    stop at their own EOS and are padded with 0 up to the longest row OF THE SHARD (so shards return different widths,
    like batches that end at different steps), the token stream is a deterministic function of the chunk's audio and
    of its prompt, never of the batch it sits in."""
    x, prompt = shard["inputs"], shard["decoder_input_ids"]
    n, P = prompt.shape
    new = (x[:, 0].abs() * 10).long() % (MAXLEN - P) + 1
    width = P + (int(new.max().item()) if n else 0)
    out = torch.zeros((n, width), dtype=torch.long)
    out[:, :P] = prompt
    for i in range(n):
        body = (x[i, : new[i]] * 1000).long().abs() % 1800 + 3 + prompt[i].sum()
        out[i, P: P + new[i]] = body
    return out, {"generated_tokens": int(new.sum()), "row_offset": shard.get("_row_offset", 0)}


def _standin_refine(shard, toks):
    """stand-in for the diffusion stage: coordinates are a function of the chunk's audio and its tokens"""
    x = shard["inputs"]
    n = x.shape[0]
    base = x[:, : 2 * TQ].reshape(n, 2, TQ) * 100.0
    return base + toks.sum(1).to(torch.float32)[:, None, None] * 1e-3


def _inputs(B):
    g = torch.Generator().manual_seed(0)
    audio = torch.randn(B, 64, generator=g)
    prompt = torch.zeros((B, 3), dtype=torch.long)
    for b in range(B):                       # ragged, left-padded prompts
        k = 1 + b % 3
        prompt[b, 3 - k:] = torch.arange(1, k + 1) + b
    return audio, prompt


def _worker(rank, world, port, B, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mapperatorinator_amd.sharding import sharded_generate
        audio, prompt = _inputs(B)
        mk = dict(inputs=audio, decoder_input_ids=prompt, flag=3)
        toks, lens, stats, coords = sharded_generate(_standin_generate, mk, pad_id=0, max_length=MAXLEN,
                                                     refine_fn=_standin_refine)
        plain = sharded_generate(_standin_generate, mk, pad_id=0, max_length=MAXLEN)
        ret[rank] = (toks, lens, stats, coords, plain[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 8, 1])
def test_two_rank_gather_equals_single_process(B):
    """world_size 2 over gloo: tokens (ragged widths per shard) AND the diffusion coordinates of every chunk arrive in
    global order on both ranks and equal the single-process run; B = 1 leaves rank 1 with an empty shard."""
    from mapperatorinator_amd.sharding import shard_bounds
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, B, ret), nprocs=2, join=True)
    audio, prompt = _inputs(B)
    want, _ = _standin_generate(dict(inputs=audio, decoder_input_ids=prompt))
    want_c = _standin_refine(dict(inputs=audio), want)
    for r in range(2):
        toks, lens, stats, coords, plain = ret[r]
        assert toks.shape == (B, MAXLEN) and torch.equal(toks, plain)
        assert coords.shape == (B, 2, TQ) and coords.dtype == torch.float32
        lo, hi = shard_bounds(B, r, 2)
        if hi > lo:     # (an empty shard never calls generate_fn)
            assert stats["row_offset"] == lo
        for b in range(B):
            row = want[b]
            n = int((row != 0).nonzero().max()) + 1
            assert torch.equal(toks[b, :n], row[:n]) and (toks[b, n:] == 0).all()
        # the refine stand-in sees the shard-local padded tokens: same sum as the global ones (pads are zeros)
        assert torch.equal(coords, want_c), (coords - want_c).abs().max()
        lo, hi = shard_bounds(B, r, 2)
    assert torch.equal(ret[0][0], ret[1][0]) and torch.equal(ret[0][3], ret[1][3])
    # the produced lengths are per SHARD (each shard pads to its own longest row)
    for r in range(2):
        lo, hi = shard_bounds(B, r, 2)
        if hi > lo:
            assert int(ret[0][1][lo:hi].max()) == int((want[lo:hi] != 0).nonzero()[:, 1].max()) + 1


# ---- bench.py --gpus N is its own launcher: N ranks, never a silent single-rank run ---------------
def _run_bench(argv, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_bench_gpus_2_forms_two_ranks_by_itself():
    """`python bench.py --gpus 2` with NO launcher around it must come back from a 2-rank group (gloo here, RCCL on a GPU
    box): the launch path of the real bench, minus the workload."""
    p, lines = _run_bench(["--gpus", "2", "--selftest-launch"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks"] == [0, 1]


def test_bench_gpus_8_forms_eight_ranks_by_itself():
    """the shape of the driver's scaling run (N = 8, one rank per GPU of a node): eight ranks formed by bench.py itself, rank ids
    all-gathered in order (gloo here; the same code path takes RCCL on a GPU node)"""
    p, lines = _run_bench(["--gpus", "8", "--selftest-launch"], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    assert lines[0]["n_gpus"] == 8 and lines[0]["rccl_ranks"] == list(range(8))


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    """a launcher that formed 1 rank while the command says --gpus 2 is an error, not a 1-GPU measurement"""
    p, lines = _run_bench(["--gpus", "2", "--selftest-launch"], env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not lines
    assert "must never" in p.stderr
