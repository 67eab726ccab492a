#!/usr/bin/env python
"""How exactly does v_mfma_scale_f32_16x16x128_f8f6f4 add?  K = 128 (one MFMA per output), operands whose quantisation is
exact, one k block of A scaled up by 2^s: the contribution of the OTHER blocks shrinks relative to it; the error of the device
result against the float64 product, in units of the largest single product, shows the width of the adder's alignment."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapperatorinator_amd import _lib as L  # noqa: E402
from oracle import mx8 as omx  # noqa: E402


def gemm(A, W):
    lib = L.load()
    M, K = A.shape
    N = W.shape[0]
    qa, sa = omx.quantize_mx8(A)
    qw, sw = omx.quantize_mx8(W)
    assert np.array_equal(omx.dequantize_mx8(qa, sa), A.astype(np.float64)) and np.array_equal(omx.dequantize_mx8(qw, sw), W.astype(np.float64)), "inputs must be exactly representable"
    t = [torch.from_numpy(v).cuda().contiguous() for v in (qa, sa, qw, sw)]
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    g = L.MhGemm()
    g.A, g.lda, g.W, g.ldw, g.a_scale, g.w_scale = t[0].data_ptr(), K, t[2].data_ptr(), K, t[1].data_ptr(), t[3].data_ptr()
    g.M, g.N, g.K, g.dtype, g.epilogue, g.C, g.ldc = M, N, K, L.MH_MX8, L.EPI_STORE_F32, out.data_ptr(), N
    L.check(lib.mh_gemm(C.byref(g), torch.cuda.current_stream().cuda_stream), "gemm")
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.float64), A.astype(np.float64) @ W.astype(np.float64).T


def main():
    rng = np.random.default_rng(0)
    M = N = 128
    grid = np.array([0.5, 0.75, 1, 1.25, 1.5, 1.75, 2, 2.5, 3, 3.5, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28])   # exact e4m3 values once a 256..448 element sets the scale
    for K in (128, 512):
        for s in (0, 4, 8, 12, 16, 20, 24, 28):
            A = rng.choice(grid, (M, K)) * rng.choice([-1.0, 1.0], (M, K))
            W = rng.choice(grid, (N, K)) * rng.choice([-1.0, 1.0], (N, K))
            A[:, ::32] = 256.0                      # every block's amax = 256 -> exponent 0: the grid values are exact e4m3
            W[:, ::32] = 256.0
            A[:, :32] *= 2.0 ** s                   # block 0 of every row of A weighs 2^s more
            got, ref = gemm(A.astype(np.float32), W.astype(np.float32))
            big = (2.0 ** s) * 256 * 256
            err = np.abs(got - ref)
            rel_out = np.max(err / np.abs(ref))
            print(f"K {K:4d} s {s:2d}: max |err| = {err.max():.4e} = 2^{np.log2(err.max() + 1e-300):6.1f}; largest product 2^{np.log2(big):.0f}; "
                  f"err / largest product = 2^{np.log2(err.max() / big + 1e-300):6.1f}; err / |result| max = {rel_out:.2e}; fp32 ulp(result) ~ 2^{np.log2(np.abs(ref).max()) - 23:.1f}")


if __name__ == "__main__":
    main()
