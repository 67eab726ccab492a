// MX-fp8 operand producers (BASELINE configs[4] "fp8 MFMA"): OCP e4m3 elements + one E8M0 scale per (row, 32 consecutive k),
// in the layouts gemm_mx8_kernel (gemm.hip) reads.  HBM-bound row kernels: one wave per row, a lane owns 4 consecutive values
// per 256-value chunk, so a 32-value MX block is 8 consecutive lanes (amax by three xor-shuffles) and the quantised row is
// written as one dword per lane (256 coalesced bytes per wave and chunk).
//
// Quantisation rule (include/mapperhip.h, mh_quantize_mx8; oracle/mx8.py restates it): amax -> e = floor(log2 amax) - 8, + 1
// when amax * 2^-e > 448 (so no element is ever clipped), scale byte e + 127, element = RNE_e4m3(x * 2^-e) by
// v_cvt_pk_fp8_f32.  Scale bytes are stored lane-major in groups of four 128-k steps: byte (kt/4)*16 + lg*4 + (kt%4).
#include "internal.hpp"

#define MH_TRY_RC(expr)           \
  do {                            \
    int rc_ = (expr);             \
    if (rc_ != MH_OK) return rc_; \
  } while (0)

namespace mh {
namespace {

// 4 fp32 values of a lane, block amax over the 8 lanes of its 32-value block -> 4 e4m3 bytes + the block's scale byte
__device__ inline uint32_t mx8_quant4(float a, float b, float c, float d, int& scale_byte) {
  float m = fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d)));
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  m = fmaxf(m, __shfl_xor(m, 4, 64));
  int e = m > 0.f ? mx8_exponent(m) : -127;
  e = e < -127 ? -127 : (e > 127 ? 127 : e);
  scale_byte = e + 127;
  const float inv = __uint_as_float((uint32_t)(127 - e < 1 ? 1 : (127 - e > 254 ? 254 : 127 - e)) << 23);   // 2^-e
  int o = __builtin_amdgcn_cvt_pk_fp8_f32(a * inv, b * inv, 0, false);
  o = __builtin_amdgcn_cvt_pk_fp8_f32(c * inv, d * inv, o, true);
  return (uint32_t)o;
}
template <typename T> __device__ inline float4 ld4(const T* p);
template <> __device__ inline float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ inline float4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}

template <typename T>
__global__ __launch_bounds__(256) void quant_mx8_kernel(const T* __restrict__ x, int ldx, int rows, int K, uint8_t* __restrict__ q, int ldq,
                                                       uint8_t* __restrict__ sc, int ks_b) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (long)row * ldx;
  for (int i = lane * 4; i < K; i += 256) {       // K % 128 == 0: a block never straddles the row end, all 8 lanes of a block are in
    const float4 v = ld4<T>(xr + i);
    int sb;
    const uint32_t o = mx8_quant4(v.x, v.y, v.z, v.w, sb);
    *reinterpret_cast<uint32_t*>(q + (long)row * ldq + i) = o;
    if ((lane & 7) == 0) sc[(long)row * ks_b + mx8_scale_index(i)] = (uint8_t)sb;
  }
}

// RMSNorm -> MX-fp8 (T5LayerNorm, custom_transformers/t5.py:50-62): the row is read once into registers (d <= 1024)
template <int NC, bool ROUND_BF16>
__global__ __launch_bounds__(256) void rmsnorm_mx8_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, int rows, int d,
                                                         float eps, uint8_t* __restrict__ q, int ldq, uint8_t* __restrict__ sc, int ks_b) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float4 v[NC], g[NC];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane * 4 + c * 256;
    const int ic = i < d ? i : 0;
    v[c] = *reinterpret_cast<const float4*>(xr + ic);
    g[c] = *reinterpret_cast<const float4*>(w + ic);
    if (i < d) ss += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
  }
  // the SAME summation order as rmsnorm_kernel (lane-strided partial sums, then the wave sum): identical fp32 statistics
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)d + eps);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane * 4 + c * 256;
    float o0 = g[c].x * (v[c].x * rs), o1 = g[c].y * (v[c].y * rs), o2 = g[c].z * (v[c].z * rs), o3 = g[c].w * (v[c].w * rs);
    if (ROUND_BF16) {
      const uint32_t h01 = pack_bf16x2(o0, o1), h23 = pack_bf16x2(o2, o3);
      o0 = __uint_as_float(h01 << 16); o1 = __uint_as_float(h01 & 0xffff0000u);
      o2 = __uint_as_float(h23 << 16); o3 = __uint_as_float(h23 & 0xffff0000u);
    }
    if (i >= d) { o0 = o1 = o2 = o3 = 0.f; }
    int sb;
    const uint32_t o = mx8_quant4(o0, o1, o2, o3, sb);           // (all 64 lanes take part in the shuffles)
    if (i < d) {
      *reinterpret_cast<uint32_t*>(q + (long)row * ldq + i) = o;
      if ((lane & 7) == 0) sc[(long)row * ks_b + mx8_scale_index(i)] = (uint8_t)sb;
    }
  }
}

// LayerNorm (no affine) + adaLN modulate -> MX-fp8 (osu_diffusion/utils/models.py:11-12,145,152): arithmetic of ln_modulate_kernel
template <int NC>
__global__ __launch_bounds__(256) void ln_modulate_mx8_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ shift,
                                                             const float* __restrict__ scale, int mod_ld, int rows_per_batch, int rows, int d,
                                                             float eps, uint8_t* __restrict__ q, int ldq, uint8_t* __restrict__ sc, int ks_b) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  const int bidx = row / rows_per_batch;
  const float* sh = shift + (long)bidx * mod_ld;
  const float* scl = scale + (long)bidx * mod_ld;
  float4 v[NC], a[NC], b[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane * 4 + c * 256;
    const int ic = i < d ? i : 0;
    v[c] = *reinterpret_cast<const float4*>(xr + ic);
    a[c] = *reinterpret_cast<const float4*>(sh + ic);
    b[c] = *reinterpret_cast<const float4*>(scl + ic);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
    if (lane * 4 + c * 256 < d) s += v[c].x + v[c].y + v[c].z + v[c].w;
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
    if (lane * 4 + c * 256 < d) {
      const float p = v[c].x - mean, r = v[c].y - mean, t = v[c].z - mean, u = v[c].w - mean;
      ss += p * p + r * r + t * t + u * u;
    }
  const float rs = rsqrtf(wave_sum(ss) / (float)d + eps);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane * 4 + c * 256;
    float o0 = (v[c].x - mean) * rs * (1.f + b[c].x) + a[c].x, o1 = (v[c].y - mean) * rs * (1.f + b[c].y) + a[c].y;
    float o2 = (v[c].z - mean) * rs * (1.f + b[c].z) + a[c].z, o3 = (v[c].w - mean) * rs * (1.f + b[c].w) + a[c].w;
    if (i >= d) { o0 = o1 = o2 = o3 = 0.f; }
    int sb;
    const uint32_t o = mx8_quant4(o0, o1, o2, o3, sb);
    if (i < d) {
      *reinterpret_cast<uint32_t*>(q + (long)row * ldq + i) = o;
      if ((lane & 7) == 0) sc[(long)row * ks_b + mx8_scale_index(i)] = (uint8_t)sb;
    }
  }
}

int check_mx8_out(int K, const void* q, int ldq, const void* sc, const char* who) {
  MH_REQUIRE(q && sc, "%s: null output", who);
  MH_REQUIRE(K > 0 && K % 128 == 0, "%s: K = %d must be a positive multiple of 128", who, K);
  MH_REQUIRE(ldq >= K && ldq % 16 == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)sc % 4) == 0, "%s: ldq %% 16, 16-byte aligned q, 4-byte aligned scales", who);
  return MH_OK;
}

}  // namespace

int quantize_mx8(const void* x, int ldx, int rows, int K, int in_dtype, uint8_t* q, int ldq, uint8_t* scales, hipStream_t s) {
  MH_REQUIRE(x && rows > 0, "mh_quantize_mx8: bad arguments");
  MH_TRY_RC(check_mx8_out(K, q, ldq, scales, "mh_quantize_mx8"));
  MH_REQUIRE(in_dtype == MH_F32 || in_dtype == MH_BF16, "mh_quantize_mx8: in_dtype must be MH_F32 or MH_BF16");
  // the kernel reads 16-byte vectors of fp32 (dwordx4) and 8-byte vectors of bf16: the rows must be aligned for THOSE loads
  MH_REQUIRE(ldx >= K && ldx % 4 == 0 && ((uintptr_t)x % (in_dtype == MH_F32 ? 16 : 8)) == 0,
             "mh_quantize_mx8: ldx %% 4 == 0 and x aligned to 16 bytes (fp32) / 8 bytes (bf16)");
  const int ks_b = mx8_scale_row_bytes(K);
  // bytes of the last scale group that no K step owns must be defined (the GEMM loads whole dwords): zero the array first
  // when K is not a multiple of 512
  if (K % 512 != 0 && hipMemsetAsync(scales, 0, (size_t)rows * ks_b, s) != hipSuccess) return check_launch("mh_quantize_mx8 memset");
  dim3 grid(ceil_div(rows, 4)), block(256);
  if (in_dtype == MH_BF16) hipLaunchKernelGGL(quant_mx8_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, ldx, rows, K, q, ldq, scales, ks_b);
  else hipLaunchKernelGGL(quant_mx8_kernel<float>, grid, block, 0, s, (const float*)x, ldx, rows, K, q, ldq, scales, ks_b);
  return check_launch("quant_mx8_kernel");
}

int rmsnorm_mx8(const float* x, int ldx, const float* w, int rows, int d, float eps, int round_dtype, uint8_t* q, int ldq,
                uint8_t* scales, hipStream_t s) {
  MH_REQUIRE(x && w && rows > 0, "mh_rmsnorm_mx8: bad arguments");
  MH_TRY_RC(check_mx8_out(d, q, ldq, scales, "mh_rmsnorm_mx8"));
  MH_REQUIRE(ldx % 4 == 0 && d <= 1024, "mh_rmsnorm_mx8: ldx %% 4 == 0 and d <= 1024 (the row lives in registers)");
  MH_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, "mh_rmsnorm_mx8: x and w must be 16-byte aligned (float4 loads)");
  MH_REQUIRE(round_dtype == MH_F32 || round_dtype == MH_BF16, "mh_rmsnorm_mx8: round_dtype must be MH_F32 or MH_BF16");
  const int ks_b = mx8_scale_row_bytes(d);
  if (d % 512 != 0 && hipMemsetAsync(scales, 0, (size_t)rows * ks_b, s) != hipSuccess) return check_launch("mh_rmsnorm_mx8 memset");
  dim3 grid(ceil_div(rows, 4)), block(256);
  const bool rb = round_dtype == MH_BF16;
#define MH_RN_CASE(NC) case NC: if (rb) hipLaunchKernelGGL((rmsnorm_mx8_kernel<NC, true>), grid, block, 0, s, x, ldx, w, rows, d, eps, q, ldq, scales, ks_b); \
                                else hipLaunchKernelGGL((rmsnorm_mx8_kernel<NC, false>), grid, block, 0, s, x, ldx, w, rows, d, eps, q, ldq, scales, ks_b); break;
  switch (ceil_div(d, 256)) { MH_RN_CASE(1) MH_RN_CASE(2) MH_RN_CASE(3) MH_RN_CASE(4) }
#undef MH_RN_CASE
  return check_launch("rmsnorm_mx8_kernel");
}

int ln_modulate_mx8(const float* x, int ldx, const float* shift, const float* scale, int mod_ld, int rows_per_batch, int rows, int d,
                    float eps, uint8_t* q, int ldq, uint8_t* scales, hipStream_t s) {
  MH_REQUIRE(x && shift && scale && rows > 0 && rows_per_batch > 0, "ln_modulate_mx8: bad arguments");
  MH_TRY_RC(check_mx8_out(d, q, ldq, scales, "ln_modulate_mx8"));
  MH_REQUIRE(ldx % 4 == 0 && mod_ld % 4 == 0 && d <= 1536, "ln_modulate_mx8: alignment / d <= 1536");
  MH_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)shift % 16) == 0 && ((uintptr_t)scale % 16) == 0,
             "ln_modulate_mx8: x, shift and scale must be 16-byte aligned (float4 loads)");
  const int ks_b = mx8_scale_row_bytes(d);
  if (d % 512 != 0 && hipMemsetAsync(scales, 0, (size_t)rows * ks_b, s) != hipSuccess) return check_launch("ln_modulate_mx8 memset");
  dim3 grid(ceil_div(rows, 4)), block(256);
#define MH_LN_CASE(NC) case NC: hipLaunchKernelGGL((ln_modulate_mx8_kernel<NC>), grid, block, 0, s, x, ldx, shift, scale, mod_ld, rows_per_batch, rows, d, eps, q, ldq, scales, ks_b); break;
  switch (ceil_div(d, 256)) { MH_LN_CASE(1) MH_LN_CASE(2) MH_LN_CASE(3) MH_LN_CASE(4) MH_LN_CASE(5) MH_LN_CASE(6) }
#undef MH_LN_CASE
  return check_launch("ln_modulate_mx8_kernel");
}

}  // namespace mh

extern "C" int64_t mh_mx8_scale_row_bytes(int K) { return K > 0 ? mh::mx8_scale_row_bytes(K) : -1; }
extern "C" int mh_quantize_mx8(const void* x, int ldx, int rows, int K, int in_dtype, uint8_t* q, int ldq, uint8_t* scales, void* stream) {
  return mh::quantize_mx8(x, ldx, rows, K, in_dtype, q, ldq, scales, (hipStream_t)stream);
}
extern "C" int mh_rmsnorm_mx8(const float* x, int ldx, const float* w, int rows, int d, float eps, int round_dtype, uint8_t* q, int ldq,
                              uint8_t* scales, void* stream) {
  return mh::rmsnorm_mx8(x, ldx, w, rows, d, eps, round_dtype, q, ldq, scales, (hipStream_t)stream);
}
