#!/usr/bin/env python
"""Probe sweep of the decode cross-attention kernel variants (key splits x waves, unroll): one process per
configuration because the knobs are read once from the environment.  Prints GB/s (algorithmic bytes / time)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    from mapperatorinator_amd import _lib
    lib = _lib.load()
    B, H, L, nd = int(os.environ.get('PROBE_B', '32')), 12, 1251, 12
    cfg = _lib.MhT5Config(768, 64, 2048, H, 12, nd, 1849, 1849, 388, 416, L, 512, _lib.MH_BF16, 1e-6)
    kv = (torch.randn(nd, 2, B, H, L, 64, device="cuda") * 0.3).to(torch.bfloat16)
    ws = torch.empty(int(lib.mh_t5_decode_workspace_bytes(C.byref(cfg), B)), dtype=torch.uint8, device="cuda")
    ms = C.c_float(0)
    st = torch.cuda.Stream()
    best = 1e9
    for _ in range(3):
        _lib.check(lib.mh_t5_cross_attn_probe(C.byref(cfg), None, kv.data_ptr(), B, 20 * nd, C.byref(ms), ws.data_ptr(), ws.numel(), st.cuda_stream))
        best = min(best, ms.value)
    bytes_ = B * H * L * 64 * 2 * 2
    print(f"B={B} ({B*H} pairs) splits={os.environ.get('MH_CROSS_SPLITS')} U={os.environ.get('MH_CROSS_U')}: {best*1e3:.2f} us/launch  {bytes_/best/1e6:.0f} GB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        # batch sizes chosen for the occupancy question: 384 pairs = 1.5 per CU (B=32), 504 = ~2 per CU (B=42),
        # 252 = ~1 per CU (B=21), 768 = 3 per CU (B=64)
        for b in ("32", "21", "42", "64"):
            for sp in ("1", "2", "4"):
                for u in ("1", "2"):
                    if b != "32" and sp != "1":
                        continue
                    env = dict(os.environ, MH_CROSS_SPLITS=sp, MH_CROSS_U=u, PROBE_B=b)
                    subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env)
