// Row-wise normalisation kernels (HBM-bound, one wave per row, 16-byte vector access).
//   rmsnorm      : T5LayerNorm  (HF modeling_t5 T5LayerNorm; restated custom_transformers/t5.py:50-62)
//   ln_modulate  : LayerNorm(no affine, eps) followed by adaLN modulate x*(1+scale)+shift
//                  (osu_diffusion/utils/models.py:11-12,110,117,140-155)
#include "internal.hpp"

namespace mh {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ w, T* __restrict__ y, int ldy,
                                                     int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float ss = 0.f;
  for (int i = lane * 4; i < d; i += 256) {
    float4 v = *reinterpret_cast<const float4*>(xr + i);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)d + eps);
  T* yr = y + (long)row * ldy;
  for (int i = lane * 4; i < d; i += 256) {
    float4 v = *reinterpret_cast<const float4*>(xr + i);
    float4 g = *reinterpret_cast<const float4*>(w + i);
    yr[i + 0] = Elem<T>::from_f32(g.x * (v.x * rs));
    yr[i + 1] = Elem<T>::from_f32(g.y * (v.y * rs));
    yr[i + 2] = Elem<T>::from_f32(g.z * (v.z * rs));
    yr[i + 3] = Elem<T>::from_f32(g.w * (v.w * rs));
  }
}

// nn.LayerNorm with affine parameters (HF Whisper's blocks, library arch 2): y = (x - mean) * rsqrt(var + eps) * w + b, biased
// variance of the CENTRED values (the two-pass arithmetic of F.layer_norm); one wave per row, the row is re-read from L1 / L2
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                       const float* __restrict__ b, T* __restrict__ y, int ldy, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float su = 0.f;
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + i);
    su += (v.x + v.y) + (v.z + v.w);
  }
  const float mu = wave_sum(su) / (float)d;
  float ss = 0.f;
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + i);
    const float p = v.x - mu, q = v.y - mu, r = v.z - mu, t = v.w - mu;
    ss += (p * p + q * q) + (r * r + t * t);
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)d + eps);
  T* yr = y + (long)row * ldy;
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + i);
    const float4 g = *reinterpret_cast<const float4*>(w + i);
    const float4 c = *reinterpret_cast<const float4*>(b + i);
    yr[i + 0] = Elem<T>::from_f32((v.x - mu) * rs * g.x + c.x);
    yr[i + 1] = Elem<T>::from_f32((v.y - mu) * rs * g.y + c.y);
    yr[i + 2] = Elem<T>::from_f32((v.z - mu) * rs * g.z + c.z);
    yr[i + 3] = Elem<T>::from_f32((v.w - mu) * rs * g.w + c.w);
  }
}

// x [rows, d] fp32; shift/scale: [n_batch, mod_ld] rows selected by row / rows_per_batch.  y: fp32 or bf16 (the operand
// rounding of the DiT's bf16 mode happens here, after the fp32 arithmetic).  One wave per row; the row (d <= 256 * NC floats)
// is read ONCE into registers -- mean, the centred sum of squares (same two-pass arithmetic as before, on the registers) and
// the modulated output all come from that copy; the shift / scale vectors are requested with the row.
template <typename T, int NC, bool SPLIT = false>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ scale, int mod_ld,
                                                         int rows_per_batch, T* __restrict__ y, int ldy,
                                                         int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  const int bidx = row / rows_per_batch;
  const float* sh = shift + (long)bidx * mod_ld;
  const float* sc = scale + (long)bidx * mod_ld;
  float4 v[NC], a[NC], b[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane * 4 + c * 256;
    const int ic = i < d ? i : 0;                      // clamped address, value masked below
    v[c] = *reinterpret_cast<const float4*>(xr + ic);
    a[c] = *reinterpret_cast<const float4*>(sh + ic);
    b[c] = *reinterpret_cast<const float4*>(sc + ic);
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
      const float p = v[c].x - mean, q = v[c].y - mean, r = v[c].z - mean, t = v[c].w - mean;
      ss += p * p + q * q + r * r + t * t;
    }
  const float rs = rsqrtf(wave_sum(ss) / (float)d + eps);
  T* yr = y + (long)row * ldy;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane * 4 + c * 256;
    if (i >= d) continue;
    float4 o;
    o.x = (v[c].x - mean) * rs * (1.f + b[c].x) + a[c].x;
    o.y = (v[c].y - mean) * rs * (1.f + b[c].y) + a[c].y;
    o.z = (v[c].z - mean) * rs * (1.f + b[c].z) + a[c].z;
    o.w = (v[c].w - mean) * rs * (1.f + b[c].w) + a[c].w;
    if constexpr (SPLIT) {   // fp32 values as [32 x bf16 hi | 32 x bf16 lo] per 32 (the pre-split A operand of gemm_s3g_kernel)
      const uint32_t h01 = pack_bf16x2(o.x, o.y), h23 = pack_bf16x2(o.z, o.w);
      const float r0 = o.x - __uint_as_float(h01 << 16), r1 = o.y - __uint_as_float(h01 & 0xffff0000u);
      const float r2 = o.z - __uint_as_float(h23 << 16), r3 = o.w - __uint_as_float(h23 & 0xffff0000u);
      char* dst = reinterpret_cast<char*>(yr) + (long)(i >> 5) * 128 + (i & 31) * 2;
      *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
      *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));
    } else if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(yr + i) = o;
    } else {
      ushort4 q;
      q.x = Elem<T>::from_f32(o.x); q.y = Elem<T>::from_f32(o.y); q.z = Elem<T>::from_f32(o.z); q.w = Elem<T>::from_f32(o.w);
      *reinterpret_cast<ushort4*>(yr + i) = q;
    }
  }
}

}  // namespace

int rmsnorm(const float* x, int ldx, const float* w, void* y, int ldy, int rows, int d, float eps, int out_dtype,
            hipStream_t s) {
  MH_REQUIRE(x && w && y && rows > 0 && d > 0, "mh_rmsnorm: bad arguments");
  MH_REQUIRE(d % 4 == 0 && ldx % 4 == 0, "mh_rmsnorm: d and ldx must be multiples of 4");
  dim3 grid(ceil_div(rows, 4)), block(256);
  if (out_dtype == MH_BF16)
    hipLaunchKernelGGL(rmsnorm_kernel<bf16_t>, grid, block, 0, s, x, ldx, w, (bf16_t*)y, ldy, rows, d, eps);
  else
    hipLaunchKernelGGL(rmsnorm_kernel<float>, grid, block, 0, s, x, ldx, w, (float*)y, ldy, rows, d, eps);
  return check_launch("rmsnorm_kernel");
}

int layernorm(const float* x, int ldx, const float* w, const float* b, void* y, int ldy, int rows, int d, float eps, int out_dtype,
              hipStream_t s) {
  MH_REQUIRE(x && w && b && y && rows > 0 && d > 0, "mh_layernorm: bad arguments");
  MH_REQUIRE(d % 4 == 0 && ldx % 4 == 0, "mh_layernorm: d and ldx must be multiples of 4");
  dim3 grid(ceil_div(rows, 4)), block(256);
  if (out_dtype == MH_BF16)
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, block, 0, s, x, ldx, w, b, (bf16_t*)y, ldy, rows, d, eps);
  else
    hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, s, x, ldx, w, b, (float*)y, ldy, rows, d, eps);
  return check_launch("layernorm_kernel");
}

template <typename T, int NC, bool SPLIT = false>
static void launch_ln_modulate(const float* x, int ldx, const float* shift, const float* scale, int mod_ld, int rows_per_batch,
                               void* y, int ldy, int rows, int d, float eps, hipStream_t s) {
  hipLaunchKernelGGL((ln_modulate_kernel<T, NC, SPLIT>), dim3(ceil_div(rows, 4)), dim3(256), 0, s, x, ldx, shift, scale, mod_ld,
                     rows_per_batch, (T*)y, ldy, rows, d, eps);
}

int ln_modulate(const float* x, int ldx, const float* shift, const float* scale, int mod_ld, int rows_per_batch,
                void* y, int ldy, int rows, int d, float eps, int out_dtype, hipStream_t s) {
  MH_REQUIRE(d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && mod_ld % 4 == 0, "ln_modulate: alignment");
  MH_REQUIRE(d <= 1536, "ln_modulate: d = %d > 1536 (the row lives in registers)", d);
  const int nc = ceil_div(d, 256);
  const bool lo = out_dtype == MH_BF16;
  MH_REQUIRE(out_dtype != MH_LN_SPLIT3 || (ldy % 32 == 0 && d % 32 == 0), "ln_modulate: the pre-split output needs d, ldy %% 32 == 0");
#define MH_LN_CASE(NC) case NC: if (out_dtype == MH_LN_SPLIT3) launch_ln_modulate<float, NC, true>(x, ldx, shift, scale, mod_ld, rows_per_batch, y, ldy, rows, d, eps, s); \
                                else if (lo) launch_ln_modulate<bf16_t, NC>(x, ldx, shift, scale, mod_ld, rows_per_batch, y, ldy, rows, d, eps, s); \
                                else launch_ln_modulate<float, NC>(x, ldx, shift, scale, mod_ld, rows_per_batch, y, ldy, rows, d, eps, s); break;
  switch (nc) { MH_LN_CASE(1) MH_LN_CASE(2) MH_LN_CASE(3) MH_LN_CASE(4) MH_LN_CASE(5) MH_LN_CASE(6) }
#undef MH_LN_CASE
  return check_launch("ln_modulate_kernel");
}

}  // namespace mh

extern "C" int mh_rmsnorm(const float* x, int ldx, const float* w, void* y, int ldy, int rows, int d, float eps,
                          int out_dtype, void* stream) {
  return mh::rmsnorm(x, ldx, w, y, ldy, rows, d, eps, out_dtype, (hipStream_t)stream);
}
extern "C" int mh_layernorm(const float* x, int ldx, const float* w, const float* b, void* y, int ldy, int rows, int d, float eps,
                            int out_dtype, void* stream) {
  return mh::layernorm(x, ldx, w, b, y, ldy, rows, d, eps, out_dtype, (hipStream_t)stream);
}
