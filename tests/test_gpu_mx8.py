"""-m gpu: the MX-fp8 operand mode (BASELINE configs[4] "fp8 MFMA") through the C ABI -- device quantisers, the
v_mfma_scale_f32_16x16x128_f8f6f4 GEMM and its epilogues -- against oracle/mx8.py (numpy restatement of the OCP MX format
and of the scale rule) and against the host packer mapperatorinator_amd/mx8.py.

The reference has no fp8 path: what is pinned is (i) that the device computes exactly the arithmetic the mode claims -- the
quantised bytes and scales BIT-EXACT against the oracle, the product within fp32 accumulation noise of the float64 product of
the dequantised operands -- and (ii) error bounds of the modes built on it against the fp32 reference goldens (test_gpu_dit /
test_gpu_t5)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from mapperatorinator_amd import _lib
    return _lib, _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def heavy_tailed(rows, K, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((rows, K)) * np.exp(rng.standard_normal((rows, K)) * 1.5)
    x[rows // 3, : K // 4] = 0.0                   # all-zero blocks
    x[rows // 2, 5] = 3.0e4                          # one outlier: its block's scale jumps
    return x.astype(np.float32)


def device_quantize(x_t, in_dtype):
    L, lib = _lib()
    rows, K = x_t.shape
    ks = int(lib.mh_mx8_scale_row_bytes(K))
    q = torch.full((rows, K), 0xAA, dtype=torch.uint8, device="cuda")
    s = torch.full((rows, ks), 0xAA, dtype=torch.uint8, device="cuda")
    L.check(lib.mh_quantize_mx8(x_t.data_ptr(), K, rows, K, in_dtype, q.data_ptr(), K, s.data_ptr(), _stream()), "mh_quantize_mx8")
    torch.cuda.synchronize()
    return q, s


@pytest.mark.parametrize("K", [128, 384, 768, 1024, 2816])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_quantizer_is_bit_exact_against_the_oracle_and_the_host_packer(K, dt):
    from mapperatorinator_amd import mx8 as host
    from oracle import mx8 as omx
    L, lib = _lib()
    x = heavy_tailed(67, K, seed=K)
    xt = torch.from_numpy(x)
    if dt == "bf16":
        xt = xt.to(torch.bfloat16)
        x = xt.float().numpy()
    q, s = device_quantize(xt.cuda().contiguous(), L.MH_BF16 if dt == "bf16" else L.MH_F32)
    qo, so = omx.quantize_mx8(x)
    assert np.array_equal(s.cpu().numpy(), so), "scale bytes differ from the oracle"
    assert np.array_equal(q.cpu().numpy(), qo), "e4m3 bytes differ from the oracle"
    qh, sh = host.quantize_mx8(torch.from_numpy(x))
    assert torch.equal(qh, q.cpu()) and torch.equal(sh, s.cpu()), "host packer and device quantiser disagree"
    # the format's promise: relative error of an element <= 2^-4 of its block's amax (3 mantissa bits), nothing clipped
    d = omx.dequantize_mx8(qo, so)
    amax = np.abs(x).reshape(67, K // 32, 32).max(2, keepdims=True).repeat(32, 2).reshape(67, K)
    assert np.all(np.abs(d - x) <= amax * 2.0 ** -4 + 1e-30)


def test_rmsnorm_mx8_equals_rmsnorm_then_quantize():
    """the fused producer writes the bytes the two-pass form (mh_rmsnorm to bf16, mh_quantize_mx8) writes"""
    L, lib = _lib()
    rows, d = 333, 768
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(rows, d, generator=g) * 3).cuda()
    w = (1 + 0.2 * torch.randn(d, generator=g)).cuda()
    ks = int(lib.mh_mx8_scale_row_bytes(d))
    for rd, td in ((L.MH_BF16, torch.bfloat16), (L.MH_F32, torch.float32)):
        y = torch.empty((rows, d), dtype=td, device="cuda")
        L.check(lib.mh_rmsnorm(x.data_ptr(), d, w.data_ptr(), y.data_ptr(), d, rows, d, 1e-6, rd, _stream()), "mh_rmsnorm")
        q2, s2 = device_quantize(y, rd)
        q1 = torch.empty((rows, d), dtype=torch.uint8, device="cuda")
        s1 = torch.empty((rows, ks), dtype=torch.uint8, device="cuda")
        L.check(lib.mh_rmsnorm_mx8(x.data_ptr(), d, w.data_ptr(), rows, d, 1e-6, rd, q1.data_ptr(), d, s1.data_ptr(), _stream()), "mh_rmsnorm_mx8")
        torch.cuda.synchronize()
        assert torch.equal(q1, q2) and torch.equal(s1, s2)


def run_mx8_gemm(A, W, epi, bias=None, C0=None, gate=None, rows_per_batch=0):
    """A [M, K], W [N, K] fp32 numpy -> (device result as fp32 numpy, float64 reference on the dequantised operands)"""
    from oracle import mx8 as omx
    L, lib = _lib()
    M, K = A.shape
    N = W.shape[0]
    qa, sa = omx.quantize_mx8(A)
    qw, sw = omx.quantize_mx8(W)
    ref = omx.mx8_matmul(qa, sa, qw, sw)
    t = [torch.from_numpy(v).cuda().contiguous() for v in (qa, sa, qw, sw)]
    g = L.MhGemm()
    g.A, g.lda, g.W, g.ldw, g.a_scale, g.w_scale = t[0].data_ptr(), K, t[2].data_ptr(), K, t[1].data_ptr(), t[3].data_ptr()
    g.M, g.N, g.K, g.dtype, g.epilogue = M, N, K, L.MH_MX8, epi
    keep = list(t)
    if bias is not None:
        b = torch.from_numpy(bias).cuda(); keep.append(b); g.bias = b.data_ptr()
    if epi == L.EPI_STORE_F32:
        out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    elif epi in (L.EPI_RESID, L.EPI_GATE_RESID):
        out = torch.from_numpy(C0).cuda().clone()
        if gate is not None:
            gt = torch.from_numpy(gate).cuda(); keep.append(gt)
            g.gate, g.gate_ld, g.rows_per_batch = gt.data_ptr(), gt.shape[1], rows_per_batch
    elif epi == L.EPI_GEGLU:
        out = torch.zeros((M, N // 2), dtype=torch.bfloat16, device="cuda")
    else:
        out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    g.C, g.ldc = out.data_ptr(), out.shape[1]
    L.check(lib.mh_gemm(C.byref(g), _stream()), "mh_gemm(MX8)")
    torch.cuda.synchronize()
    return out.float().cpu().numpy(), ref


# What the matrix core delivers (tools/mx8_precision_probe.py, profiles/r04_micro_mx8_precision.txt): inside one
# v_mfma_scale_f32_16x16x128_f8f6f4 the 128 products are aligned to the LARGEST of them and ~13 bits below it survive (exactly
# representable operands, equal scales: error 2^-10.8 of the largest product, whatever the scales) -- the fp8 matrix path does
# not add in fp32.  A dot product is therefore good to ~K * 2^-13 of its largest product <= 2^-6 of sum |a||w| in the worst
# case, 2^-12 measured on heavy-tailed data; a wrong byte or scale association is off by >= 2^-4 of it.
TOL_MAG = 2.0 ** -9


# shapes: K steps 1 .. 24 (every tail case of the unrolled loop), ragged M / N, both tile forms
SHAPES = [(256, 128, 128), (300, 200, 256), (512, 384, 384), (200, 136, 512), (1024, 768, 640), (2048, 256, 768),
          (640, 1024, 1024), (384, 2304, 1152), (4096, 512, 2816), (512, 768, 3072), (8192, 2304, 768)]


@pytest.mark.parametrize("shape", SHAPES)
def test_mx8_gemm_against_the_float64_product_of_the_dequantised_operands(shape):
    L, lib = _lib()
    M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    A = heavy_tailed(M, K, seed=1)
    W = (rng.standard_normal((N, K)) * rng.uniform(0.2, 3.0, (N, 1))).astype(np.float32)      # asymmetric: row scales differ
    got, ref = run_mx8_gemm(A, W, L.EPI_STORE_F32)
    # fp32 accumulation of K products whose partial sums reach |a|.|w|: bound the error by the row / column magnitudes
    from oracle import mx8 as omx
    qa, sa = omx.quantize_mx8(A)
    qw, sw = omx.quantize_mx8(W)
    mag = np.abs(omx.dequantize_mx8(qa, sa)) @ np.abs(omx.dequantize_mx8(qw, sw)).T
    err = np.abs(got - ref)
    assert np.all(err <= mag * TOL_MAG + 1e-20), f"worst {np.max(err / (mag + 1e-30)):.3e} of the magnitude at {np.unravel_index(np.argmax(err / (mag + 1e-30)), err.shape)}"


@pytest.mark.parametrize("K", [128, 384, 1024])
def test_mx8_gemm_is_exact_where_the_adder_window_holds_every_product(K):
    """values 8 .. 15 (exact e4m3), block scales 2^0 .. 2^3 per (row, block) on both operands: every product of a dot product
    lies within 2^10 of the largest, inside the matrix core's alignment window -- the result must be the exact integer.  This
    pins the byte -> k and the scale -> (row, block) association of both operands (a wrong one is off by whole products)."""
    L, lib = _lib()
    rng = np.random.default_rng(K)
    M, N = 160, 144

    def operand(rows):
        v = rng.integers(8, 16, (rows, K)).astype(np.float64) * rng.choice([-1.0, 1.0], (rows, K))
        v[:, ::32] = 15.0                                                  # every block's amax is 15 * 2^e
        e = rng.integers(0, 4, (rows, K // 32))
        return (v.reshape(rows, K // 32, 32) * 2.0 ** e[:, :, None]).reshape(rows, K).astype(np.float32)
    A, W = operand(M), operand(N)
    got, ref = run_mx8_gemm(A, W, L.EPI_STORE_F32)
    assert np.array_equal(ref, A.astype(np.float64) @ W.astype(np.float64).T), "the quantisation of these operands is exact"
    assert np.array_equal(got, ref)


def test_mx8_gemm_epilogues():
    L, lib = _lib()
    M, N, K = 520, 256, 768
    rng = np.random.default_rng(3)
    A = heavy_tailed(M, K, seed=2) * 0.05
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    gate = rng.standard_normal((M // 130, N)).astype(np.float32)

    def gelu(x):
        return 0.5 * x * (1 + np.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))
    # tolerance of the product itself: the matrix core's alignment window (TOL_MAG of sum |a||w|, see above)
    from oracle import mx8 as omx
    mag = np.abs(omx.dequantize_mx8(*omx.quantize_mx8(A))) @ np.abs(omx.dequantize_mx8(*omx.quantize_mx8(W))).T
    tol = TOL_MAG * mag + 1e-5

    def close(got, want, scale=1.0, rel=0.0):
        return np.all(np.abs(got - want) <= tol * scale + rel * np.abs(want))
    got, ref = run_mx8_gemm(A, W, L.EPI_STORE, bias=bias)
    assert close(got, ref + bias, rel=2.0 ** -8)                                                # (+ bf16 rounding of the output)
    got, ref = run_mx8_gemm(A, W, L.EPI_RESID, C0=C0)
    assert close(got, C0 + ref)
    got, ref = run_mx8_gemm(A, W, L.EPI_GATE_RESID, bias=bias, C0=C0, gate=gate, rows_per_batch=130)
    gfull = np.repeat(gate, 130, 0)
    assert close(got, C0 + gfull * (ref + bias), scale=np.abs(gfull) + 1e-3)
    got, ref = run_mx8_gemm(A, W, L.EPI_BIAS_GELU, bias=bias)
    assert close(got, gelu(ref + bias), scale=1.2, rel=2.0 ** -8)                                # |gelu'| <= 1.13
    got, ref = run_mx8_gemm(A, W, L.EPI_GEGLU)
    r = ref.reshape(M, N // 32, 2, 16)
    t = tol.reshape(M, N // 32, 2, 16)
    want = (gelu(r[:, :, 0]) * r[:, :, 1]).reshape(M, N // 2)
    bound = (1.2 * t[:, :, 0] * np.abs(r[:, :, 1]) + t[:, :, 1] * np.abs(gelu(r[:, :, 0])) + t[:, :, 0] * t[:, :, 1]).reshape(M, N // 2)
    assert np.all(np.abs(got - want) <= bound + 2.0 ** -8 * np.abs(want) + 1e-5)


def test_mx8_gemm_refuses_what_it_cannot_do():
    L, lib = _lib()
    a = torch.zeros((128, 192), dtype=torch.uint8, device="cuda")
    s = torch.zeros((128, 16), dtype=torch.uint8, device="cuda")
    out = torch.zeros((128, 128), dtype=torch.float32, device="cuda")
    g = L.MhGemm()
    g.A, g.lda, g.W, g.ldw, g.a_scale, g.w_scale = a.data_ptr(), 192, a.data_ptr(), 192, s.data_ptr(), s.data_ptr()
    g.M, g.N, g.K, g.dtype, g.epilogue, g.C, g.ldc = 128, 128, 192, L.MH_MX8, L.EPI_STORE_F32, out.data_ptr(), 128
    assert lib.mh_gemm(C.byref(g), _stream()) != 0 and b"K %% 128" not in lib.mh_last_error() and b"128" in lib.mh_last_error()
    g.K, g.lda, g.ldw, g.a_scale = 128, 192, 192, None
    assert lib.mh_gemm(C.byref(g), _stream()) != 0 and b"a_scale" in lib.mh_last_error()


@pytest.mark.parametrize("epi_name", ["BIAS_GELU", "GEGLU"])
@pytest.mark.parametrize("shape", [(520, 256, 768), (4099, 1024, 384), (8192, 3072, 768), (2048, 5632, 256)])
def test_fused_mx_output_is_the_quantiser_over_the_bf16_result(epi_name, shape):
    """ABI 9 (MhGemm.mx_out): the GELU / gated-GELU GEMM of the MX modes writes its result as the next GEMM's MX-fp8 operand
    itself.  Byte for byte (elements AND scales) what mh_quantize_mx8 makes of the bf16 matrix the plain epilogue writes -- both
    tile forms, ragged M, zero blocks and outliers included; rows beyond M and the bf16 output are not touched."""
    L, lib = _lib()
    from oracle import mx8 as omx
    M, N, K = shape
    epi = L.EPI_BIAS_GELU if epi_name == "BIAS_GELU" else L.EPI_GEGLU
    width = N if epi == L.EPI_BIAS_GELU else N // 2
    rng = np.random.default_rng(M + N)
    A = heavy_tailed(M, K, seed=K) * 0.05
    A[7] = 0.0                                                       # a whole zero row: zero blocks in the output
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    W[: N // 4] *= 40.0                                              # a wide range of block magnitudes across the output row
    bias = (rng.standard_normal(N).astype(np.float32) if epi == L.EPI_BIAS_GELU else None)
    qa, sa = omx.quantize_mx8(A)
    qw, sw = omx.quantize_mx8(W)
    t = [torch.from_numpy(v).cuda().contiguous() for v in (qa, sa, qw, sw)]
    b = torch.from_numpy(bias).cuda() if bias is not None else None

    def run(fused):
        g = L.MhGemm()
        g.A, g.lda, g.W, g.ldw, g.a_scale, g.w_scale = t[0].data_ptr(), K, t[2].data_ptr(), K, t[1].data_ptr(), t[3].data_ptr()
        g.M, g.N, g.K, g.dtype, g.epilogue = M, N, K, L.MH_MX8, epi
        if b is not None:
            g.bias = b.data_ptr()
        out = torch.full((M, width), 7.0, dtype=torch.bfloat16, device="cuda")
        g.C, g.ldc = out.data_ptr(), width
        ks = int(lib.mh_mx8_scale_row_bytes(width))
        q = torch.full((M + 3, width), 0xAA, dtype=torch.uint8, device="cuda")
        s = torch.zeros((M + 3, ks), dtype=torch.uint8, device="cuda")
        if fused:
            g.mx_out, g.mx_out_scales = q.data_ptr(), s.data_ptr()
        L.check(lib.mh_gemm(C.byref(g), _stream()), "mh_gemm(MX8)")
        torch.cuda.synchronize()
        return out, q, s
    plain, _, _ = run(False)
    untouched, q, s = run(True)
    assert bool((untouched == 7.0).all()), "the bf16 output must not be written in the fused form"
    assert bool((q[M:] == 0xAA).all()) and bool((s[M:] == 0).all()), "rows beyond M were written"
    want_q, want_s = device_quantize(plain, L.MH_BF16)
    if width % 512:      # the quantiser zeroes the unused tail bytes of the last scale group; so did the zero-filled buffer here
        assert int(lib.mh_mx8_scale_row_bytes(width)) == want_s.shape[1]
    assert torch.equal(q[:M], want_q), f"{int((q[:M] != want_q).sum())} element bytes differ"
    assert torch.equal(s[:M], want_s), f"{int((s[:M] != want_s).sum())} scale bytes differ"
