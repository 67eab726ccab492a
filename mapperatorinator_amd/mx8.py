"""MX-fp8 operands on the host side (BASELINE configs[4] "fp8 MFMA"): the quantiser that packs WEIGHTS for `mh_gemm` with
dtype MH_MX8 -- OCP e4m3 elements + one E8M0 scale byte per (row, 32 consecutive k) -- in exactly the layout and by exactly
the rule `mh_quantize_mx8` / csrc/mx8.hip use for activations on the device (include/mapperhip.h):

    amax = max |x| over the block;  e = floor(log2 amax) - 8, raised by one when amax * 2^-e > 448 (nothing is ever clipped)
    scale byte = e + 127 (all-zero block: 0);  element = RNE_e4m3(x * 2^-e)
    scales of a row: 16 * ceil(K / 512) bytes, byte (kt // 4) * 16 + lg * 4 + (kt % 4) = block kt * 4 + lg  (kt = k // 128)

Torch ops only (runs on the CPU or on the GPU the weights are packed on); `torch.float8_e4m3fn` does the element rounding."""
from __future__ import annotations

import torch


def scale_row_bytes(K: int) -> int:
    return 16 * ((K + 511) // 512)


def scale_byte_index(K: int) -> torch.Tensor:
    """int64 [K // 32]: position of block b's scale byte inside a row's scale bytes"""
    b = torch.arange(K // 32)
    kt, lg = b // 4, b % 4
    return (kt // 4) * 16 + lg * 4 + (kt % 4)


def quantize_mx8(x: torch.Tensor):
    """x [rows, K] (K % 128 == 0) float -> (q uint8 [rows, K] e4m3 bit patterns, scales uint8 [rows, scale_row_bytes(K)])"""
    rows, K = x.shape
    if K % 128:
        raise ValueError(f"MX-fp8 operands need K % 128 == 0 (K = {K})")
    xb = x.detach().to(torch.float32).reshape(rows, K // 32, 32)
    amax = xb.abs().amax(dim=2)
    _, ex = torch.frexp(amax)                                   # amax = m * 2^ex, m in [0.5, 1)  ->  floor(log2 amax) = ex - 1
    e = ex.to(torch.int32) - 1 - 8
    e = torch.where(torch.ldexp(amax, -e) > 448.0, e + 1, e)
    e = torch.where(amax > 0, e, torch.full_like(e, -127)).clamp(-127, 127)
    scaled = torch.ldexp(xb, (-e).unsqueeze(-1).expand_as(xb))
    q = scaled.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).reshape(rows, K)
    scales = torch.zeros((rows, scale_row_bytes(K)), dtype=torch.uint8, device=x.device)
    scales[:, scale_byte_index(K).to(x.device)] = (e + 127).to(torch.uint8)
    return q.contiguous(), scales.contiguous()


def dequantize_mx8(q: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """the fp32 values an MX-fp8 operand stands for (exact: e4m3 x 2^e fits fp32)"""
    rows, K = q.shape
    e = scales[:, scale_byte_index(K).to(q.device)].to(torch.int32) - 127
    v = q.view(torch.float8_e4m3fn).to(torch.float32).reshape(rows, K // 32, 32)
    return torch.ldexp(v, e.unsqueeze(-1).expand_as(v)).reshape(rows, K)
