// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libmapperhip.
// Wave = 64 lanes everywhere; MFMA atoms are the 16x16 family so that the C/D fragment layout
// (col = lane&15, row = (lane>>4)*4 + reg) is identical for the bf16 and the exact-f32 atom.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mapperhip.h"

namespace mh {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef uint16_t bf16_t;  // raw storage type of a bf16 element

// ---- error plumbing (api.hip) ------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---- tuning options (api.hip: default <- environment at first use <- mh_set_option at run time) ----
enum { OPT_GEMM_SPLITK_TILES = 0, OPT_DECODE_CHAINS, OPT_DECODE_PREFILL, OPT_DECODE_GEMV_COLS, OPT_DECODE_FUSED_PROJ, OPT_GEMM_TILE128_MIN, OPT_DIT_SPLIT3_MIN_ROWS, OPT_GEMM_GLDS, OPT_DECODE_LAUNCH_THREADS, OPT_DECODE_GRAPH_CACHE, OPT_GEMM_TILE256SQ_MIN, OPT_GEMM_2STAGE_MAX_K, OPT_DIT_SKINNY_MAX_ROWS, OPT_COUNT };
long option(int id);   // the calling thread's option set (OptionScope) first, then the process-wide value
// RAII: the entry points that take a config install its option set for the calling thread; launcher threads re-install it
struct OptionScope {
  const MhOptionSet* prev;
  explicit OptionScope(const MhOptionSet* set);
  ~OptionScope();
  OptionScope(const OptionScope&) = delete;
  OptionScope& operator=(const OptionScope&) = delete;
};
const MhOptionSet* current_option_set();

#define MH_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      ::mh::set_error(__VA_ARGS__);    \
      return MH_ERR_ARG;               \
    }                                  \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, NaN-preserving) -----------------------------------
__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)v) << 16;
  return c.f;
}
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint16_t, (__bf16)f);   // v_cvt_pk_bf16_f32 (round-to-nearest-even)
#endif
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two fp32 -> packed bf16x2 (lo in bits 0..15) with ONE v_cvt_pk_bf16_f32
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kVec = 4;  // elements per 16-byte vector
  __device__ static inline float to_f32(float v) { return v; }
  __device__ static inline float from_f32(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kVec = 8;
  __device__ static inline float to_f32(bf16_t v) { return bf16_to_f32(v); }
  __device__ static inline bf16_t from_f32(float v) { return f32_to_bf16(v); }
};

// ---- 16x16 MFMA atoms ----------------------------------------------------------------------
// Operand convention for both atoms (D = A * B^T-stored):
//   A fragment: lane l holds A[row = l&15][k = kbase + (l>>4)*KCH .. +KCH)
//   B fragment: lane l holds W[col = l&15][k = kbase + (l>>4)*KCH .. +KCH)   (W is [N][K] row-major)
//   C fragment: acc[r] = C[row = (l>>4)*4 + r][col = l&15]
template <typename T> struct Atom;
template <> struct Atom<bf16_t> {
  static constexpr int KCH = 8;   // contiguous k elements per lane
  static constexpr int KM = 32;   // k covered by one MFMA
  typedef bf16x8_t frag_t;
  __device__ static inline frag_t load(const bf16_t* p) {  // p 16-byte aligned
    return *reinterpret_cast<const frag_t*>(p);
  }
  __device__ static inline f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Atom<float> {
  static constexpr int KCH = 1;
  static constexpr int KM = 4;
  typedef float frag_t;
  __device__ static inline frag_t load(const float* p) { return *p; }
  __device__ static inline f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// ---- wave / block reductions -----------------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduce over groups of G consecutive lanes (G power of two <= 64)
template <int G> __device__ inline float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum via LDS scratch (scratch must hold >= nwaves floats); all threads get the result
__device__ inline float block_sum(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}
__device__ inline float block_max(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = -INFINITY;
  for (int i = 0; i < nw; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

// exp: accurate expf for the fp32 parity path, hardware exp2-based fast path for bf16 storage
template <typename T> __device__ inline float fexp(float x);
template <> __device__ inline float fexp<float>(float x) { return expf(x); }
template <> __device__ inline float fexp<bf16_t>(float x) { return __expf(x); }

// ---- activations (fp32 internals) ------------------------------------------------------------
// gelu_new == GELU(approximate='tanh'): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
// (HF NewGELUActivation, used by T5 gated-gelu; torch nn.GELU('tanh') used by the DiT Mlp)
__device__ inline float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
// the same function through one exponential: 0.5 x (1 + tanh u) = x / (1 + exp(-2u)) -- 7 instructions instead of tanhf's
// ~45 with two branches (raw v_exp_f32 / v_rcp_f32: ~1e-6 relative).  Used by the DiT's fc1 epilogues only (its parity
// gates are error bounds); the T5 gated-GELU keeps gelu_tanh, whose fp32 results decide bit-exact greedy ids.
__device__ inline float gelu_tanh_fast(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.885390081777927f * u));
}
// 0.5 x (1 + erf(x / sqrt 2)) through Abramowitz & Stegun 7.1.26 (|error of erf| <= 1.5e-7) on raw v_rcp_f32 / v_exp_f32: ~14
// instructions instead of erff's ~40 with branches.  For bf16 / MX-fp8 OUTPUTS only (the Whisper-family fc1 epilogue of the LDS-DMA GEMMs);
// the fp32 kernels keep erff.
__device__ inline float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);      // erf(|x| / sqrt 2)
  return 0.5f * x * (1.0f + copysignf(e, x));
}
__device__ inline float silu(float x) { return x / (1.0f + __expf(-x)); }

// T5 relative position bucket (restated from the published T5 formula; reference restatement at
// osuT5/osuT5/model/custom_transformers/t5.py:88-141).  rel = key_pos - query_pos.
__host__ __device__ inline int t5_bucket(int rel, bool bidirectional, int num_buckets, int max_distance) {
  int ret = 0;
  int n = num_buckets;
  int rp;
  if (bidirectional) {
    n /= 2;
    if (rel > 0) ret += n;
    rp = rel < 0 ? -rel : rel;
  } else {
    rp = rel < 0 ? -rel : 0;  // -min(rel, 0)
  }
  int max_exact = n / 2;
  if (rp < max_exact) return ret + rp;
  // max_exact + floor( log(rp/max_exact) / log(max_distance/max_exact) * (n - max_exact) )
  float v = logf((float)rp / (float)max_exact) / logf((float)max_distance / (float)max_exact) * (float)(n - max_exact);
  int large = max_exact + (int)v;
  if (large > n - 1) large = n - 1;
  return ret + large;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// MX-fp8 operands: E8M0 scale bytes per row (lane-major groups of four 128-k steps, 16 bytes each; mx8.hip)
inline int mx8_scale_row_bytes(int K) { return 16 * ((K + 511) / 512); }

// ---- MX-fp8 quantisation rule, shared by the producers (mx8.hip) and the GEMM epilogues that write an MX operand (gemm.hip) ----
// scale exponent of a block from its amax (> 0): returns e (unbiased), never clipping
__device__ inline int mx8_exponent(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;       // floor(log2(amax)) - 8 (amax normal; denormal blocks -> -135, clamped below)
  const float scaled = amax * __uint_as_float((uint32_t)(127 - e) << 23);     // amax * 2^-e in [256, 512)
  return scaled > 448.0f ? e + 1 : e;
}
__device__ inline long mx8_scale_index(int k) {          // byte index of the scale of the block holding column k
  const int kt = k >> 7, lg = (k >> 5) & 3;
  return (long)(kt >> 2) * 16 + lg * 4 + (kt & 3);
}

// 2^-e as a float (e clamped to the E8M0 range)
__device__ inline float mx8_inv_scale(int e) {
  return __uint_as_float((uint32_t)(127 - e < 1 ? 1 : (127 - e > 254 ? 254 : 127 - e)) << 23);
}

}  // namespace mh
