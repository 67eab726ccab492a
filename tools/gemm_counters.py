#!/usr/bin/env python
"""Two bf16 GEMM shapes through the three-stage 256 x 128 kernel and the 256 x 256 kernel, a few launches each: a small workload
for `rocprofv3 --kernel-trace --pmc ...` counter passes (MfmaUtil, LdsBankConflict, MemUnitBusy, ...)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mapperatorinator_amd import _lib as L  # noqa: E402

lib = L.load()
s = torch.cuda.current_stream().cuda_stream
for M, N, K in ((8192, 8192, 8192), (40032, 4096, 768)):
    A = torch.randn(M, K).to(torch.bfloat16).cuda()
    W = torch.randn(N, K).to(torch.bfloat16).cuda()
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    g = L.MhGemm()
    g.A, g.lda, g.W, g.ldw, g.C, g.ldc = A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N
    g.M, g.N, g.K, g.dtype, g.epilogue = M, N, K, L.MH_BF16, L.EPI_STORE
    for thr in (0, 1):
        L.set_option("gemm_tile256sq_min", thr)
        for _ in range(3):
            L.check(lib.mh_gemm(C.byref(g), s), "gemm")
        torch.cuda.synchronize()
