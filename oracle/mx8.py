"""TEST INFRASTRUCTURE (never imported by the product path): numpy restatement of the MX-fp8 operand format of BASELINE
configs[4] -- the checker for csrc/mx8.hip (device quantisers), mapperatorinator_amd/mx8.py (host weight packer) and
gemm_mx8_kernel (csrc/gemm.hip).

The reference has NO fp8 path (inference.py:637-642 loads the DiT in fp32, `precision` offers fp32 / bf16 / amp only): this
mode is the run's own reduced-precision mode and its parity gates are ERROR BOUNDS against the fp32 reference goldens, exactly
like the bf16-operand mode of the DiT.  What is pinned here is the arithmetic the mode claims to perform:

  OCP Microscaling (MX) v1.0, MXFP8 with E4M3 elements: a block of 32 consecutive k shares one E8M0 scale 2^(s - 127); an
  element is an OCP e4m3 value (bias 7, no infinities, max 448, subnormals down to 2^-9).  The scale rule is ours (spec 6.3
  leaves it to the implementation): e = floor(log2 amax) - 8, + 1 when amax * 2^-e > 448 -- so that no element is clipped.
  A product row is sum over blocks of 2^(ea + ew) * sum_k a_k w_k, accumulated in fp32 by the matrix core; the oracle sums in
  float64 (the kernel must agree to fp32 accumulation noise)."""
from __future__ import annotations

import numpy as np


def e4m3_decode_table() -> np.ndarray:
    """float64 value of each of the 256 e4m3 bit patterns (0x7f / 0xff = NaN)"""
    t = np.zeros(256)
    for v in range(256):
        s, e, m = v >> 7, (v >> 3) & 15, v & 7
        if e == 15 and m == 7:
            x = np.nan
        elif e == 0:
            x = (m / 8.0) * 2.0 ** -6
        else:
            x = (1.0 + m / 8.0) * 2.0 ** (e - 7)
        t[v] = -x if s else x
    return t


_TAB = e4m3_decode_table()
_POS = _TAB[:127].copy()            # the 127 non-negative finite values, ascending


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """round-to-nearest-even onto the e4m3 grid (|x| <= 448 expected; larger magnitudes saturate) -> uint8 bit patterns"""
    a = np.minimum(np.abs(x).astype(np.float64), 448.0)
    hi = np.searchsorted(_POS, a, side="left").clip(0, 126)           # first grid point >= a
    lo = (hi - 1).clip(0, 126)
    dl, dh = a - _POS[lo], _POS[hi] - a
    pick_hi = (dh < dl) | ((dh == dl) & (hi % 2 == 0))                # ties to the even bit pattern
    idx = np.where(_POS[hi] == a, hi, np.where(pick_hi, hi, lo)).astype(np.uint8)
    return (idx | ((np.signbit(x)).astype(np.uint8) << 7)).astype(np.uint8)


def block_exponents(x: np.ndarray) -> np.ndarray:
    """x [rows, K] -> int [rows, K // 32]: the unbiased E8M0 exponent of every block"""
    rows, K = x.shape
    amax = np.abs(x.astype(np.float64)).reshape(rows, K // 32, 32).max(axis=2)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(amax > 0, amax, 1.0))).astype(np.int64) - 8
    e = np.where(amax * 2.0 ** (-e.astype(np.float64)) > 448.0, e + 1, e)
    return np.where(amax > 0, e, -127).clip(-127, 127)


def scale_byte_index(K: int) -> np.ndarray:
    b = np.arange(K // 32)
    kt, lg = b // 4, b % 4
    return (kt // 4) * 16 + lg * 4 + (kt % 4)


def quantize_mx8(x: np.ndarray):
    """-> (q uint8 [rows, K], scales uint8 [rows, 16 * ceil(K / 512)] in the device's lane-major layout)"""
    rows, K = x.shape
    assert K % 128 == 0
    e = block_exponents(x)
    scaled = x.astype(np.float64).reshape(rows, K // 32, 32) * 2.0 ** (-e[:, :, None].astype(np.float64))
    q = e4m3_encode(scaled).reshape(rows, K)
    scales = np.zeros((rows, 16 * ((K + 511) // 512)), np.uint8)
    scales[:, scale_byte_index(K)] = (e + 127).astype(np.uint8)
    return q, scales


def dequantize_mx8(q: np.ndarray, scales: np.ndarray) -> np.ndarray:
    rows, K = q.shape
    e = scales[:, scale_byte_index(K)].astype(np.int64) - 127
    return (_TAB[q].reshape(rows, K // 32, 32) * 2.0 ** e[:, :, None].astype(np.float64)).reshape(rows, K)


def mx8_matmul(qa, sa, qw, sw) -> np.ndarray:
    """C[m, n] = sum_k A[m, k] W[n, k] of the values the two MX operands stand for, float64"""
    return dequantize_mx8(qa, sa) @ dequantize_mx8(qw, sw).T


def fake_quant_torch(x):
    """torch restatement of quantise -> dequantise along the last axis (blocks of 32): the fp32 values an MX-fp8 operand made
    from `x` stands for.  Used by the model-level oracles (oracle/t5.py, oracle/dit.py) where numpy would be slow; pinned to
    the numpy form above in tests/test_host_cpu.py."""
    import torch
    shp = x.shape
    xb = x.detach().to(torch.float32).reshape(-1, shp[-1] // 32, 32)
    amax = xb.abs().amax(dim=2)
    _, ex = torch.frexp(amax)
    e = ex.to(torch.int32) - 9
    e = torch.where(torch.ldexp(amax, -e) > 448.0, e + 1, e)
    e = torch.where(amax > 0, e, torch.full_like(e, -127)).clamp(-127, 127)
    ee = e.unsqueeze(-1).expand_as(xb)
    q = torch.ldexp(xb, -ee).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
    return torch.ldexp(q, ee).reshape(shp)
