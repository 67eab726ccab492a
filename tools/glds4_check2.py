#!/usr/bin/env python
"""gemm_glds4_kernel against gemm_glds3_kernel bit for bit, every epilogue the T5 encoder uses, at encoder-sized M."""
import sys

import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import test_gpu_kernels as tk  # noqa: E402

L, lib = tk._lib()
M = 10032
g = torch.Generator().manual_seed(1)


def both(fn):
    outs = []
    for thr in (0, 1):
        L.set_option("gemm_tile256sq_min", thr)
        outs.append(fn())
    L.set_option("gemm_tile256sq_min", 440)
    a, b = outs
    if isinstance(a, tuple):
        return all(torch.equal(x, y) for x, y in zip(a, b)), max((x - y).abs().max().item() for x, y in zip(a, b))
    return torch.equal(a, b), (a - b).abs().max().item()


A768 = torch.randn(M, 768, generator=g)
A2048 = torch.randn(M, 2048, generator=g)
cases = {
    "STORE qkv": lambda: tk.run_gemm(A768, torch.randn(2304, 768, generator=torch.Generator().manual_seed(2)), L.EPI_STORE, L.MH_BF16),
    "RESID o": lambda: tk.run_gemm(A768, torch.randn(768, 768, generator=torch.Generator().manual_seed(3)) * 0.05, L.EPI_RESID, L.MH_BF16,
                                   C0=torch.randn(M, 768, generator=torch.Generator().manual_seed(4))),
    "RESID wo": lambda: tk.run_gemm(A2048, torch.randn(768, 2048, generator=torch.Generator().manual_seed(5)) * 0.05, L.EPI_RESID, L.MH_BF16,
                                    C0=torch.randn(M, 768, generator=torch.Generator().manual_seed(6))),
    "GEGLU wi": lambda: tk.run_gemm(A768, torch.randn(4096, 768, generator=torch.Generator().manual_seed(7)) * 0.05, L.EPI_GEGLU, L.MH_BF16),
    "QKV_VT": lambda: tk.run_gemm(A768[:8 * 1251], torch.randn(2304, 768, generator=torch.Generator().manual_seed(8)) * 0.05, L.EPI_QKV_VT, L.MH_BF16,
                                  kv=(8, 12, 1251), n_split=1536, Lpad=1280),
    "KV_SCATTER": lambda: tk.run_gemm(A768[:8 * 1251], torch.randn(2 * 2 * 768, 768, generator=torch.Generator().manual_seed(9)) * 0.05, L.EPI_KV_SCATTER,
                                      L.MH_BF16, kv=(8, 12, 1251)),
    "STORE_F32": lambda: tk.run_gemm(A768, torch.randn(768, 768, generator=torch.Generator().manual_seed(10)), L.EPI_STORE_F32, L.MH_BF16),
    "BIAS_GELU": lambda: tk.run_gemm(A768, torch.randn(3072, 768, generator=torch.Generator().manual_seed(11)) * 0.05, L.EPI_BIAS_GELU, L.MH_BF16,
                                     bias=torch.randn(3072, generator=torch.Generator().manual_seed(12))),
}
for name, fn in cases.items():
    try:
        eq, d = both(fn)
        print(f"{name:12s} bit-equal {eq}  max |d| {d:.4g}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name:12s} ERROR {e}", flush=True)
