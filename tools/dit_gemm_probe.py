#!/usr/bin/env python
"""Why does a 0.2-GFLOP fp32 GEMM of the one-chunk DiT step take 10-13 us?  Times mh_gemm on the step's shapes (M = 256 rows)
(a) back to back on ONE weight matrix (hot in L2), (b) cycling through 12 weight matrices + 12 activation buffers (the step's
access pattern: every operand cold in the XCD-private L2s), as eager launches and as a captured graph of 48 launches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapperatorinator_amd import _lib as L  # noqa: E402

SHAPES = [("qkv", 256, 1152, 384), ("out", 256, 384, 384), ("fc1", 256, 1536, 384), ("fc2", 256, 384, 1536)]


def main():
    lib = L.load()
    for o in [a for a in sys.argv[1:] if "=" in a]:
        k, v = o.split("=")
        L.set_option(k, int(v))
    s = torch.cuda.current_stream().cuda_stream
    g0 = torch.Generator().manual_seed(0)
    for name, M, N, K in SHAPES:
        Ws = [torch.randn(N, K, generator=g0).cuda() * 0.05 for _ in range(12)]
        As = [torch.randn(M, K, generator=g0).cuda() for _ in range(12)]
        bias = torch.zeros(N, device="cuda")
        out = torch.empty(M, N, device="cuda")

        def launch(i):
            gm = L.MhGemm()
            gm.A, gm.lda, gm.W, gm.ldw, gm.C, gm.ldc = As[i].data_ptr(), K, Ws[i].data_ptr(), K, out.data_ptr(), N
            gm.M, gm.N, gm.K, gm.dtype, gm.epilogue, gm.bias = M, N, K, L.MH_F32, L.EPI_STORE_F32, bias.data_ptr()
            L.check(lib.mh_gemm(C.byref(gm), s), "mh_gemm")

        def timed(fn, reps):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3

        hot = timed(lambda: [launch(0) for _ in range(48)], 20) / 48
        cold = timed(lambda: [launch(i % 12) for i in range(48)], 20) / 48
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(48):
                launch(i % 12)
        gcold = timed(graph.replay, 20) / 48
        print(f"{name:4s} M {M} N {N:5d} K {K:5d} | same operands {hot:6.2f} us | 12 operand sets {cold:6.2f} us | captured graph of the latter {gcold:6.2f} us per launch")


if __name__ == "__main__":
    main()
