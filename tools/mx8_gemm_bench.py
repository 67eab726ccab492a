#!/usr/bin/env python
"""Throughput of the MX-fp8 GEMM next to the bf16 three-stage GEMM on the shapes configs[4] runs (encoder / cross-KV of
osuT5-base and -large at 32 chunks, DiT-B block GEMMs at 32 chunks).  Prints TFLOP/s and the fraction of the dense peaks
(5 PFLOP/s MX-fp8, 2.5 PFLOP/s bf16; /opt/skills/guides/MI355X_MICROARCH.md)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mapperatorinator_amd import _lib as L  # noqa: E402
from mapperatorinator_amd import mx8 as host  # noqa: E402

SHAPES = [("base qkv", 40032, 2304, 768), ("base o", 40032, 768, 768), ("base wi", 40032, 4096, 768), ("base wo", 40032, 768, 2048),
          ("base cross-kv", 40032, 18432, 768), ("large wi", 40032, 5632, 1024), ("large wo", 40032, 1024, 2816),
          ("dit-b qkv", 8192, 2304, 768), ("dit-b out", 8192, 768, 768), ("dit-b fc1", 8192, 3072, 768), ("dit-b fc2", 8192, 768, 3072),
          ("square 8k", 8192, 8192, 8192)]


def bench(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    opts = [a for a in sys.argv[1:] if "=" in a]
    for o in opts:
        k, v = o.split("=")
        L.set_option(k, int(v))
    print(f"{'shape':16s} {'M':>6s} {'N':>6s} {'K':>5s} | {'bf16 us':>9s} {'TF':>7s} {'frac':>6s} | {'mx8 us':>9s} {'TF':>7s} {'frac':>6s} | speed-up")
    for name, M, N, K in SHAPES:
        g = torch.Generator().manual_seed(0)
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g)
        Ab, Wb = A.to(torch.bfloat16).cuda(), W.to(torch.bfloat16).cuda()
        out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        gb = L.MhGemm()
        gb.A, gb.lda, gb.W, gb.ldw, gb.C, gb.ldc = Ab.data_ptr(), K, Wb.data_ptr(), K, out.data_ptr(), N
        gb.M, gb.N, gb.K, gb.dtype, gb.epilogue = M, N, K, L.MH_BF16, L.EPI_STORE
        tb = bench(lambda: L.check(lib.mh_gemm(C.byref(gb), s), "bf16"))
        qa, sa = host.quantize_mx8(A.cuda())
        qw, sw = host.quantize_mx8(W.cuda())
        gm = L.MhGemm()
        gm.A, gm.lda, gm.W, gm.ldw, gm.C, gm.ldc = qa.data_ptr(), K, qw.data_ptr(), K, out.data_ptr(), N
        gm.a_scale, gm.w_scale = sa.data_ptr(), sw.data_ptr()
        gm.M, gm.N, gm.K, gm.dtype, gm.epilogue = M, N, K, L.MH_MX8, L.EPI_STORE
        tm = bench(lambda: L.check(lib.mh_gemm(C.byref(gm), s), "mx8"))
        fl = 2.0 * M * N * K
        print(f"{name:16s} {M:6d} {N:6d} {K:5d} | {tb * 1e6:9.1f} {fl / tb / 1e12:7.0f} {fl / tb / 2.5e15:6.3f} | {tm * 1e6:9.1f} {fl / tm / 1e12:7.0f} "
              f"{fl / tm / 5e15:6.3f} | {tb / tm:5.2f}x")
    # the quantiser pass an activation pays in front of the GEMM when its producer does not write MX directly
    x = torch.randn(40032, 2048, device="cuda").to(torch.bfloat16)
    q = torch.empty((40032, 2048), dtype=torch.uint8, device="cuda")
    sc = torch.empty((40032, 64), dtype=torch.uint8, device="cuda")
    tq = bench(lambda: L.check(lib.mh_quantize_mx8(x.data_ptr(), 2048, 40032, 2048, L.MH_BF16, q.data_ptr(), 2048, sc.data_ptr(), s), "q"))
    print(f"quantize 40032 x 2048 bf16 -> MX8: {tq * 1e6:.1f} us = {(40032 * 2048 * 3.03) / tq / 1e12:.2f} TB/s")


if __name__ == "__main__":
    main()
