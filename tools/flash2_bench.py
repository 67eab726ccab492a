#!/usr/bin/env python
"""Times the bf16 transposed-S attention kernel (csrc/attention.hip::flash2_bf16_kernel) alone on the shapes the library runs it
on: the T5 encoder at 32 chunks (relative bias), the batched DiT (band mask).  MAPPERHIP_LIB selects a probe build
(`tools/build_variant.sh NAME attention.hip -DMH_F2_PROBE=...`): parts of the kernel switched off to price them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapperatorinator_amd import _lib as L  # noqa: E402

SHAPES = [("t5-base encoder", 32, 12, 1251, True, 1.0, 0), ("t5-large encoder", 32, 16, 1251, True, 1.0, 0),
          ("dit-b batched", 64, 12, 128, False, 0.125, 128), ("dit 1024 window", 2, 12, 1024, False, 0.125, 128)]


def main():
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    for o in [a for a in sys.argv[1:] if "=" in a]:
        k, v = o.split("=")
        L.set_option(k, int(v))
    for name, B, H, Ln, has_bias, scale, band in SHAPES:
        g = torch.Generator().manual_seed(1)
        inner = H * 64
        qk = (torch.randn(B * Ln, 2 * inner, generator=g) * 0.6).to(torch.bfloat16).cuda()
        Lpad = (Ln + 63) // 64 * 64
        vt = torch.randn(B, H, 64, Lpad, generator=g).to(torch.bfloat16).cuda()
        bias = (torch.randn(H, 2 * Ln - 1, generator=g) * 0.5).cuda() if has_bias else None
        out = torch.zeros(B * Ln, inner, dtype=torch.bfloat16, device="cuda")

        def fn():
            L.check(lib.mh_attention(qk.data_ptr(), 2 * inner, inner, vt.data_ptr(), Lpad, L.ptr(bias), out.data_ptr(), inner, B, Ln, H,
                                     scale, band, L.MH_BF16, s), "mh_attention")
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        keys = min(Ln, 2 * band + 64) if band else Ln           # (banded: the tiles a query block visits, roughly)
        fl = 4.0 * B * H * Ln * keys * 64
        print(f"{name:18s} B {B:3d} H {H:2d} L {Ln:5d} | {t * 1e6:8.1f} us  {fl / t / 1e12:6.0f} TFLOP/s  {fl / t / 2.5e15:5.3f} of bf16 peak  checksum {out.float().sum().item():.4f}")


if __name__ == "__main__":
    main()
