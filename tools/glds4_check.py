#!/usr/bin/env python
"""The 256 x 256 bf16 GEMM tile (gemm_glds4_kernel) against the 256 x 128 three-stage kernel, bit for bit, on encoder-sized
shapes: the two accumulate the same products in the same order, so every output bit must agree; repeated to catch a race."""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mapperatorinator_amd import _lib as L  # noqa: E402

SHAPES = [(40032, 2304, 768), (40032, 768, 768), (40032, 768, 2048), (40032, 4096, 768), (10240, 1024, 2816), (8192, 8192, 1024),
          (2500, 768, 128), (2500, 768, 64), (4099, 1280, 192)]


def main():
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    bad = 0
    for M, N, K in SHAPES:
        g = torch.Generator().manual_seed(M + N + K)
        A = (torch.randn(M, K, generator=g)).to(torch.bfloat16).cuda()
        W = (torch.randn(N, K, generator=g)).to(torch.bfloat16).cuda()
        outs = {}
        for mode, thr in (("glds3", 0), ("glds4", 1)):
            L.set_option("gemm_tile256sq_min", thr)
            for rep in range(3):
                out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
                gb = L.MhGemm()
                gb.A, gb.lda, gb.W, gb.ldw, gb.C, gb.ldc = A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N
                gb.M, gb.N, gb.K, gb.dtype, gb.epilogue = M, N, K, L.MH_BF16, L.EPI_STORE
                L.check(lib.mh_gemm(C.byref(gb), s), mode)
                torch.cuda.synchronize()
                outs[(mode, rep)] = out
        ref = outs[("glds3", 0)]
        for key, o in outs.items():
            if not torch.equal(o.view(torch.int16), ref.view(torch.int16)):
                d = (o.float() - ref.float()).abs()
                idx = (o.view(torch.int16) != ref.view(torch.int16)).nonzero()
                bad += 1
                print(f"{M}x{N}x{K} {key}: {idx.shape[0]} elements differ, max |d| {d.max().item():.4g}; first at {idx[:4].tolist()}, "
                      f"rows {sorted(set((idx[:, 0] // 256).tolist()))[:8]} cols {sorted(set((idx[:, 1] // 256).tolist()))[:8]}")
        print(f"{M}x{N}x{K}: checked", flush=True)
    print("MISMATCHES" if bad else "all bit-equal")


if __name__ == "__main__":
    main()
