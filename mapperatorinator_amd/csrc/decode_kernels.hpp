// Device kernels of the KV-cached autoregressive T5 decode step (K5 / K6).
//
// One token step of one layer is SIX dependent kernels per row chain (a chain = up to 16 rows; two chains run side by
// side): [RMSNorm + this head's q / k / v projection + self-KV append + self-attention over the cache] -> [O GEMV +
// residual] -> [RMSNorm + this head's query projection + cross-attention over the 1251 encoder keys: THE HBM-bound
// kernel, B*H*L*64*2 elements per layer per step] -> [O GEMV + residual] -> [RMSNorm + wi GEMV + gated GELU] -> [wo GEMV
// + residual]; then final RMSNorm + lm_head GEMV and the sampler (t5.hip).  Every kernel reads the current position
// from device memory, so a chain's step is one replayable hipGraph.  (Option decode_fused_proj = 0 / 2 runs the q / k / v
// and cross-query projections as stand-alone GEMVs in front of dec_self_attn_kernel / dec_cross_attn_kernel instead.)
//
// What shapes these kernels (measured): a dependent kernel costs ~3.3 us before it does anything (launch,
// first-load latency, store flush), so each does its whole job in one or two memory round trips; the CU's load path is
// charged per 128-byte line touched, so operands are loaded as whole lines (weights of the attention kernels' own
// projections; the GEMVs' activations through a wave-private LDS patch) and only where that is impossible as MFMA-
// fragment-shaped pieces; everything handed to the next kernel is stored write-through.
//
// GEMVs ("skinny GEMMs", M = rows of the chain): a workgroup owns one 16-column MFMA tile (4, 8 or 16 real columns),
// its 4 or 8 waves split K, weight fragments go straight from L2 / HBM into registers (read once), the partial
// accumulators are added through LDS in wave order (deterministic, independent of the batch).
#pragma once
#include <type_traits>

#include "internal.hpp"

namespace mh {
namespace dec {

// ---- phase stamps (profiling build only: `make prof` defines MH_PHASE_STAMPS) --------------------------------------------
// Thread 0 of workgroup 0 adds (shader-clock ticks since its first instruction) to g_stamps[kernel][stamp] at marked
// points; tools/decode_phases.py prints the averages.  The production library contains none of this.
#ifdef MH_PHASE_STAMPS
__device__ unsigned long long g_stamps[16 * 16 * 2];   // [kernel id][stamp][sum of ticks, count]
#define MH_STAMP0() unsigned long long mh_t0_ = 0; if (blockIdx.x == 0 && threadIdx.x == 0) mh_t0_ = clock64()
#define MH_STAMP(kid, i)                                                              \
  do {                                                                                \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                        \
      atomicAdd(&g_stamps[((kid) * 16 + (i)) * 2], (unsigned long long)clock64() - mh_t0_); \
      atomicAdd(&g_stamps[((kid) * 16 + (i)) * 2 + 1], 1ull);                         \
    }                                                                                 \
  } while (0)
#else
#define MH_STAMP0() do {} while (0)
#define MH_STAMP(kid, i) do {} while (0)
#endif
enum { KID_GEMV = 0 /* + EPI * 2 + (NWV == 8) */, KID_SELF = 10, KID_CROSS = 11, KID_SAMPLE = 12 };

// ---- 8-element chunk helpers ----------------------------------------------------------------
template <typename T> __device__ inline void load8(const T* p, float (&o)[8]);
template <> __device__ inline void load8<bf16_t>(const bf16_t* p, float (&o)[8]) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(w[i] << 16);
    o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <> __device__ inline void load8<float>(const float* p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
// streaming (non-temporal) variants for data that is read once per launch (K/V rows of the decode attention):
// `global_load_dwordx4 ... nt` keeps the 1.5 GB/step K/V stream from evicting the decoder weights (226 MB) out of
// the 256 MB Infinity Cache between token steps.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
template <typename T> __device__ inline void load8_stream(const T* p, float (&o)[8]);
template <> __device__ inline void load8_stream<bf16_t>(const bf16_t* p, float (&o)[8]) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(v[i] << 16);
    o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
template <> __device__ inline void load8_stream<float>(const float* p, float (&o)[8]) {
  const u32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  const u32x4_t b = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p + 4));
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[i] = __uint_as_float(a[i]); o[4 + i] = __uint_as_float(b[i]); }
}
// OCP e4m3 rows (the fp8 copy of the cross-attention K / V): 8 elements = 8 bytes per lane
struct fp8_t { uint8_t v; };
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
template <> __device__ inline void load8_stream<fp8_t>(const fp8_t* p, float (&o)[8]) {
  const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[i], false);
    const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[i], true);
    o[4 * i] = lo[0]; o[4 * i + 1] = lo[1]; o[4 * i + 2] = hi[0]; o[4 * i + 3] = hi[1];
  }
}
template <typename T> __device__ inline void store8(T* p, const float (&o)[8]);
template <> __device__ inline void store8<bf16_t>(bf16_t* p, const float (&o)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
template <> __device__ inline void store8<float>(float* p, const float (&o)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

template <typename T> __device__ inline void store4(T* p, float a, float b, float c, float d);
template <> __device__ inline void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
template <> __device__ inline void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// Everything a decode kernel hands to the NEXT kernel is stored write-through ("agent scope", `global_store ... sc1`):
// the release at the end of a kernel writes back the dirty lines of every XCD's L2, and with two decode chains on the
// chip each chain's kernel boundaries also pay for the other chain's dirty lines.  With nothing dirty the boundary is
// cheaper: tools/micro/gemv_probe, o-projection 3.53 -> 3.37 us alone, 4.60 -> 4.08 us beside a second chain; the
// K = 2048 projection 6.35 -> 5.54 us (non-temporal stores: no gain).
// hipcc's scheduler sinks loads next to their first use when it believes registers are scarce (`load, s_waitcnt vmcnt(0),
// use, load, ...`: one memory round trip PER LOAD instead of one per kernel -- found in the 8-wave GEMV and, after an
// unrelated edit, in the 4-wave one: tools/isa_mem_signature.py shows it).  The cure is to tell it that registers are
// not scarce (amdgpu_waves_per_eu on the kernels below); __builtin_amdgcn_sched_barrier at this marker was tried and is
// worse: the kernel's private arrays are then promoted to LDS (a dispatch-packet read plus LDS round trips).
#define MH_LOADS_ISSUED() do {} while (0)

// (Naming the not-preloaded kernel arguments in an empty `asm volatile("" :: "s"(arg))` right after the vector loads, so
// that their s_load overlaps the vector round trip, was measured: 677 -> 840 us per token step.  Not kept.)

template <typename U>
__device__ inline void store_wt(U* p, U v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16- / 8-byte write-through stores (`__hip_atomic_store` stops at 8 bytes).  A 2- or 4-byte write-through store is one
// fabric write each (MI355X_MICROARCH.md: per byte, dword ~6x and short ~12.5x the dwordx4 time): the decode kernels stage
// what they hand on through LDS and write whole 16-byte pieces of a row.
__device__ inline void store16_wt(void* p, u32x4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
__device__ inline void store8_wt(void* p, u32x2_t v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
// n consecutive fp32 values (n = 4 or 8) -> one write-through store of n T elements when the destination is aligned to
// the piece, else element by element
#ifndef MH_PACK_ATTN
#define MH_PACK_ATTN 1     // (0: A/B builds only -- attention outputs / cache rows as 64 two-byte stores)
#endif
#ifndef MH_PACK_STORES
#define MH_PACK_STORES 1   // (0: A/B builds only -- element-wise write-through stores)
#endif
template <typename TO>
__device__ inline void store_piece_wt(TO* dst, const float* v, int n, int nvalid) {
  const bool whole = MH_PACK_STORES && nvalid == n && (reinterpret_cast<uintptr_t>(dst) & (n * sizeof(TO) - 1)) == 0;
  if (whole && sizeof(TO) == 4 && n == 4) {
    store16_wt(dst, u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])});
  } else if (whole && sizeof(TO) == 2 && n == 8) {
    store16_wt(dst, u32x4_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])});
  } else if (whole && sizeof(TO) == 2 && n == 4) {
    store8_wt(dst, u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
  } else {
    for (int i = 0; i < nvalid; ++i) store_wt(dst + i, Elem<TO>::from_f32(v[i]));
  }
}

// One wave holds the 64 values of a head row (lane = feature): through a wave-private 64-float LDS patch (one wave's LDS
// operations execute in order: no barrier) to 16-byte pieces, written by the first 8 (bf16) / 16 (fp32) lanes.
template <typename T>
__device__ inline void store_head_row_wt(T* dst, float val, float* patch) {
  const int lane = threadIdx.x & 63;
  constexpr int N = 16 / (int)sizeof(T);
  if (!MH_PACK_ATTN) { store_wt(dst + lane, Elem<T>::from_f32(val)); return; }
  patch[lane] = val;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < 64 / N) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = i < N ? patch[lane * N + i] : 0.f;
    store_piece_wt<T>(dst + lane * N, v, N, N);
  }
}
// the same from 64 floats that already sit in LDS (visible to the calling wave): lanes `lane0 .. lane0 + 64 / N` of the block
template <typename T>
__device__ inline void store_head_row_from_lds_wt(T* dst, const float* src, int t) {   // t = thread index relative to the first storing thread
  constexpr int N = 16 / (int)sizeof(T);
  if (!MH_PACK_ATTN) { if (t >= 0 && t < 64) store_wt(dst + t, Elem<T>::from_f32(src[t])); return; }
  if (t >= 0 && t < 64 / N) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = i < N ? src[t * N + i] : 0.f;
    store_piece_wt<T>(dst + t * N, v, N, N);
  }
}

// ---- decode GEMV ("skinny GEMM": M = batch rows <= 64) ------------------------------------------------------------
// out[b][n] = sum_k A[b][k] * W[n][k] on the 16x16 MFMA atoms.  One workgroup owns ONE 16-column tile of which `nv`
// columns are real (nv = 16, 8 or 4: a 768-column projection becomes 48, 96 or 192 workgroups; the other tile columns
// repeat real ones and are never stored), its NWV waves split K (wave w takes k-blocks w, w + NWV, ...), the partial
// accumulators are added through LDS in wave order (deterministic, independent of the batch).
//   * straight-line code: every load of a pass is issued before the first wait, addresses are clamped instead of
//     predicated, k-blocks beyond K are neutralised by AND-ing the activation fragment with 0 (no branches: a predicated
//     load or a branch around an MFMA made hipcc serialise the loads of the previous version into several round trips);
//   * PRO_RMSNORM: A is the fp32 residual stream; the workgroup holds all of its <= 64 rows x K values in registers at
//     once (single pass), so the RMSNorm statistics come from those registers (lane -> 16-lane shuffle -> LDS over the
//     waves): no statistics buffers, no extra memory round trip, one barrier;
//   * weights are read exactly once per workgroup straight into MFMA fragments (no LDS round trip for data used once).
//   * the activation fragments are "fragment-shaped" loads (16 rows x 64 bytes per wave instruction), which cost the
//     CU's load path about twice what whole lines cost (tools/micro/gemv_probe, mask 128: the same bytes as contiguous
//     1 KB runs take 3.35 -> 2.70 us alone, 4.13 -> 2.92 us beside a second chain).  Staging them through LDS with
//     whole-line loads was built and measured: the extra LDS write / barrier / read costs what the loads save (o-projection
//     3.61 us alone, 4.12 beside a second chain; whole step 37.4 k vs 38.1 k tok/s without it) -- not kept.
enum { PRO_PLAIN = 0, PRO_RMSNORM = 1, PRO_LAYERNORM = 2 };   // (LAYERNORM: HF Whisper's affine nn.LayerNorm, library arch 2)
enum { SK_STORE = 0, SK_QKV = 1, SK_GEGLU = 2, SK_RESID = 3, SK_LOGITS = 4, SK_GELU_ERF = 5 };   // (GELU_ERF: the Whisper family's fc1)

struct SkinnyP {
  const void* A; int lda;      // PRO_PLAIN: T [B, lda];  PRO_RMSNORM: fp32 residual stream [B, lda]
  const float* ln_w; float eps;
  const void* W; int ldw;      // [N, ldw] element type T
  int B, N, K;
  int nv;                      // real columns per 16-column tile: 16, 8 or 4 (GEGLU: 8 gate + 8 linear)
  void* out; int ldo;          // STORE: T [B, ldo]; GEGLU: T [B, ldo] (N/2 cols); LOGITS: f32 [B, ldo]
  float* h; int ldh;           // RESID: h[b][n] += acc
  void* kc; void* vc;          // QKV: this layer's self-attention caches [B][H][tgt_len][64]
  int H, tgt_len, inner;
  const int* pos;
  const float* bias;           // kernel template BIAS (the Whisper family's biased projections): fp32 [N], else unused
  const float* ln_b;           // PRO_LAYERNORM: the LayerNorm bias fp32 [K] (ln_w = its weight, eps = its epsilon)
};

template <typename T> struct VecOps;
template <> struct VecOps<bf16_t> {
  struct Raw { float4 a, b; };   // 8 consecutive fp32 values (one lane's k-chunk)
  __device__ static inline Raw load_raw(const float* p) {
    Raw r;
    r.a = *reinterpret_cast<const float4*>(p);
    r.b = *reinterpret_cast<const float4*>(p + 4);
    return r;
  }
  __device__ static inline float sumsq(const Raw& x) {
    return (x.a.x * x.a.x + x.a.y * x.a.y) + (x.a.z * x.a.z + x.a.w * x.a.w) + (x.b.x * x.b.x + x.b.y * x.b.y) +
           (x.b.z * x.b.z + x.b.w * x.b.w);
  }
  __device__ static inline float sum(const Raw& x) { return (x.a.x + x.a.y) + (x.a.z + x.a.w) + (x.b.x + x.b.y) + (x.b.z + x.b.w); }
  __device__ static inline Raw centred(const Raw& x, float mu) {
    Raw r;
    r.a = make_float4(x.a.x - mu, x.a.y - mu, x.a.z - mu, x.a.w - mu);
    r.b = make_float4(x.b.x - mu, x.b.y - mu, x.b.z - mu, x.b.w - mu);
    return r;
  }
  // nn.LayerNorm: (x - mean) * rstd * weight + bias (x arrives centred)
  __device__ static inline uint4 ln_frag(const Raw& x, const Raw& g, const Raw& b, float rs) {
    const uint32_t o0 = pack_bf16x2(x.a.x * rs * g.a.x + b.a.x, x.a.y * rs * g.a.y + b.a.y);
    const uint32_t o1 = pack_bf16x2(x.a.z * rs * g.a.z + b.a.z, x.a.w * rs * g.a.w + b.a.w);
    const uint32_t o2 = pack_bf16x2(x.b.x * rs * g.b.x + b.b.x, x.b.y * rs * g.b.y + b.b.y);
    const uint32_t o3 = pack_bf16x2(x.b.z * rs * g.b.z + b.b.z, x.b.w * rs * g.b.w + b.b.w);
    return make_uint4(o0, o1, o2, o3);
  }
  __device__ static inline uint4 norm_frag(const Raw& x, const Raw& g, float rs) {
    const uint32_t o0 = pack_bf16x2(g.a.x * (x.a.x * rs), g.a.y * (x.a.y * rs));
    const uint32_t o1 = pack_bf16x2(g.a.z * (x.a.z * rs), g.a.w * (x.a.w * rs));
    const uint32_t o2 = pack_bf16x2(g.b.x * (x.b.x * rs), g.b.y * (x.b.y * rs));
    const uint32_t o3 = pack_bf16x2(g.b.z * (x.b.z * rs), g.b.w * (x.b.w * rs));
    return make_uint4(o0, o1, o2, o3);
  }
  __device__ static inline f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
    union U { uint4 u; bf16x8_t f; };
    U ua, ub;
    ua.u = a; ub.u = b;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.f, ub.f, c, 0, 0, 0);
  }
};
template <> struct VecOps<float> {
  struct Raw { float4 a; };
  __device__ static inline Raw load_raw(const float* p) {
    Raw r;
    r.a = *reinterpret_cast<const float4*>(p);
    return r;
  }
  __device__ static inline float sumsq(const Raw& x) { return (x.a.x * x.a.x + x.a.y * x.a.y) + (x.a.z * x.a.z + x.a.w * x.a.w); }
  __device__ static inline float sum(const Raw& x) { return (x.a.x + x.a.y) + (x.a.z + x.a.w); }
  __device__ static inline Raw centred(const Raw& x, float mu) {
    Raw r;
    r.a = make_float4(x.a.x - mu, x.a.y - mu, x.a.z - mu, x.a.w - mu);
    return r;
  }
  __device__ static inline uint4 ln_frag(const Raw& x, const Raw& g, const Raw& b, float rs) {
    return make_uint4(__float_as_uint(x.a.x * rs * g.a.x + b.a.x), __float_as_uint(x.a.y * rs * g.a.y + b.a.y),
                      __float_as_uint(x.a.z * rs * g.a.z + b.a.z), __float_as_uint(x.a.w * rs * g.a.w + b.a.w));
  }
  __device__ static inline uint4 norm_frag(const Raw& x, const Raw& g, float rs) {
    return make_uint4(__float_as_uint(g.a.x * (x.a.x * rs)), __float_as_uint(g.a.y * (x.a.y * rs)),
                      __float_as_uint(g.a.z * (x.a.z * rs)), __float_as_uint(g.a.w * (x.a.w * rs)));
  }
  __device__ static inline f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
    // k permutation: element i of every lane's 4-float vector forms one 16x16x4 product
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

constexpr int kGemvCH = 8;   // k-blocks per wave whose loads are in flight at once (one pass)

// tools/micro/gemv_probe.hip compiles this kernel with parts switched OFF to price them (bit mask: 1 activation loads,
// 2 weight loads, 4 old residual values, 8 cross-wave reduction, 16 stores; 32 / 64: residual stores as agent-scope atomic /
// non-temporal stores); always 0 in the library
#ifndef MH_GEMV_PROBE
#define MH_GEMV_PROBE 0
#endif

// MF 16-row fragments (B <= 16 MF), NWV waves.  NWV depends on K only (never on the batch): a row's summation order,
// hence its rounding and its greedy tokens, must not depend on which other rows share the launch.  The k-block -> wave
// assignment is likewise a function of the prologue only: PRO_PLAIN gives wave w the k-block PAIRS w, w + NWV, ... (two
// consecutive k-blocks = one 128-byte line of a bf16 row) for every MF -- the whole-line path (MF == 1) and the
// fragment-shaped path (MF >= 2: a chain of more than 16 rows, e.g. a guidance batch) add the same products in the same
// order; PRO_RMSNORM gives wave w the single k-blocks w, w + NWV, ...
// Kernel arguments: the operands a workgroup needs to issue its first loads are LEADING SCALAR arguments -- the library is
// built with -amdgpu-kernarg-preload-count=14, so they arrive in SGPRs with the wave launch instead of through an s_load
// from the (cold) kernarg segment: one memory round trip off every dependent kernel (tools/micro/kernarg_preload.hip: a
// GEMV-like dependent kernel 2.49 -> 1.91 us alone, 3.58 -> 2.90 us beside a kernel that streams HBM).  The struct keeps
// the rest; its copies of the leading arguments are never read (lda = ldw = K: checked on the host).
// (12 dwords: the 13th and 14th preload slots do not arrive on this firmware -- the kernel body re-loads them)
#define MH_GEMV_LEAD_PARAMS const void *A_, const void *W_, float *h_, const float *lnw_, int K_, int B_, int N_, int nv_
#define MH_GEMV_LEAD_ARGS(p) (p).A, (p).W, (p).h, (p).ln_w, (p).K, (p).B, (p).N, (p).nv
#ifndef MH_GEMV_WPE
#define MH_GEMV_WPE 4
#endif
#if MH_GEMV_WPE
#define MH_GEMV_WPE_ATTR __attribute__((amdgpu_waves_per_eu(NWV / 4, MH_GEMV_WPE)))   // registers are free here: never trade a load for one
#else
#define MH_GEMV_WPE_ATTR
#endif
template <typename T, int MF, int NWV, int PRO, int EPI, bool BIAS = false>
__global__ __launch_bounds__(NWV * 64) MH_GEMV_WPE_ATTR
void gemv_kernel(MH_GEMV_LEAD_PARAMS, SkinnyP p) {
  p.A = A_; p.W = W_; p.h = h_; p.ln_w = lnw_; p.K = K_; p.lda = K_; p.ldw = K_; p.B = B_; p.N = N_; p.nv = nv_;
  if (EPI == SK_RESID) p.ldh = N_;   // the residual stream is dense [B, N] (checked on the host)
  constexpr int VEC = Elem<T>::kVec;   // elements per 16-byte vector (per lane per k-block)
  constexpr int KB = 4 * VEC;          // k elements per k-block (4 lane groups x 16 B)
  constexpr int CH = kGemvCH;
  typedef typename VecOps<T>::Raw Raw;
  // PRO_PLAIN, 16 rows (MF == 1): the activations are loaded as whole 128-byte lines -- 8 rows x 128 bytes per wave
  // instruction; wave w owns the k-block PAIRS w, w + NWV, ... -- and turned into MFMA fragments through a wave-private
  // 2.3 KB LDS patch: no barrier (the LDS operations of one wave execute in order, so the patch is reused line after
  // line), 2 ds_write_b128 + 2 ds_read_b128 per line.  Fragment-shaped global loads (16 rows x 64 bytes per instruction)
  // cost the CU's load path twice as much: tools/micro/gemv_probe, o-projection 3.35 -> 2.80 us alone, 4.13 -> 3.09 us
  // beside a second chain; K = 2048: 4.89 -> 3.66 / 5.54 -> 4.63; whole step 38.2 k -> 39.9 k tok/s.
  // PRO_RMSNORM under bf16 storage (RLINES): the fp32 residual rows are one line per row and k-block; the same patch
  // trick with the k-block -> wave assignment unchanged: RMSNorm + wi + gated GELU 5.80 -> 4.33 us alone, 6.80 -> 5.20 us
  // beside a second chain.  Loading the 16-row WEIGHT tile of that GEMV as lines as well was measured and lost (6.90 /
  // 8.69 us: 16 more LDS operations per wave than the load path saves) -- weights keep fragment-shaped loads.
#ifndef MH_GEMV_LINES
#define MH_GEMV_LINES 1   // (0: A/B builds only)
#endif
  constexpr bool LINES = (MF == 1 && PRO == PRO_PLAIN && MH_GEMV_LINES);
  constexpr int CP = CH / 2;                 // k-block pairs per wave and pass
  constexpr int PATCH = 16 * 144;            // 16 rows x (128 + 16) bytes: rows 16 bytes apart in the bank row
#ifndef MH_GEMV_RLINES
#define MH_GEMV_RLINES 1   // (0: A/B builds only)
#endif
  constexpr bool NORM = PRO != PRO_PLAIN, LN = PRO == PRO_LAYERNORM;
  constexpr bool RLINES = (MF == 1 && NORM && sizeof(T) == 2 && MH_GEMV_RLINES);
  __shared__ __attribute__((aligned(16))) unsigned char Lw[(LINES || RLINES) ? NWV * PATCH : 16];
  __shared__ f32x4_t red[NWV * MF * 64];
  __shared__ float ssw[NORM ? NWV : 1][MF * 16];
  __shared__ float ssw2[LN ? NWV : 1][MF * 16];    // LayerNorm: the centred sums of squares (second reduction)
  __shared__ __attribute__((aligned(16))) float lnw[NORM ? 1024 : 4];   // norm weight, staged once per workgroup
  __shared__ __attribute__((aligned(16))) float lnb[LN ? 1024 : 4];     // LayerNorm bias

  MH_STAMP0();
  constexpr int KID = KID_GEMV + EPI * 2 + (NWV == 8 ? 1 : 0);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nv = p.nv;
  // The vector-memory path of a CU moves 64 B per clock and charges every LANE of a load instruction, duplicates
  // included: a tile whose 16 columns repeat nv real ones must not load the repeats (exec-masked W loads below), and
  // the RMSNorm weight is fetched once per workgroup through LDS instead of once per 16-lane row group.
  float4 lnraw = make_float4(0.f, 0.f, 0.f, 0.f), lnbraw = make_float4(0.f, 0.f, 0.f, 0.f);
  if (NORM) {
    const int i4 = tid * 4 < p.K ? tid * 4 : 0;     // K <= 1024 <= 4 * (NWV * 64)
    lnraw = *reinterpret_cast<const float4*>(p.ln_w + i4);
    if (LN) lnbraw = *reinterpret_cast<const float4*>(p.ln_b + i4);
  }
  // weight row of this lane's tile column
  int wrow, ocol;      // ocol: output column of tile column l15 (GEGLU: of the gate / linear PAIR)
  bool wload;          // this lane fetches a weight fragment (tile columns >= nv are never stored: they keep zeros)
  if (EPI == SK_GEGLU) {
    // tile t = ff columns [8t, 8t + 8): tile columns 0..7 are their gate rows, 8..15 their linear rows; wi_0 / wi_1 are
    // interleaved in 16-row blocks [gate | linear] (PackedT5.interleave16)
    ocol = blockIdx.x * 8 + (l15 & 7);
    const int oc = ocol < p.N / 2 ? ocol : p.N / 2 - 1;
    wrow = (oc >> 4) * 32 + (l15 >> 3) * 16 + (oc & 15);
    wload = true;
  } else {
    ocol = blockIdx.x * nv + l15;
    wrow = ocol < p.N ? ocol : p.N - 1;
    wload = l15 < nv;
  }
  const T* Wp = reinterpret_cast<const T*>(p.W) + (long)wrow * p.ldw + lg * VEC;
  int arow[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) arow[f] = (f * 16 + l15) < p.B ? (f * 16 + l15) : p.B - 1;

  constexpr int UPW = (MF * 4 + NWV - 1) / NWV;
  const bool col_ok = (EPI == SK_GEGLU) ? (l15 < 8 && ocol < p.N / 2) : (l15 < nv && ocol < p.N);
  float oldh[UPW];
#pragma unroll
  for (int u = 0; u < UPW; ++u) oldh[u] = 0.f;
  constexpr int PROBE = MH_GEMV_PROBE;
  float bias_v = 0.f;
  if (BIAS) bias_v = p.bias[ocol < p.N ? ocol : p.N - 1];
  // (a macro, not a lambda: a by-reference capture of `oldh` made hipcc keep the array in memory and promote it to LDS --
  // 4 KB per wave, an LDS round trip per value: the residual GEMVs went from 5.4 to 17 us)
#define MH_LOAD_OLDH()                                                                                                        \
  do {                                                                                                                        \
    _Pragma("unroll") for (int u = 0; u < UPW; ++u) {                                                                         \
      const int unit = wid + u * NWV;                                                                                         \
      const int row = (unit >> 2) * 16 + lg * 4 + (unit & 3);                                                                 \
      if (unit < MF * 4) oldh[u] = p.h[(long)(row < p.B ? row : p.B - 1) * p.ldh + (ocol < p.N ? ocol : p.N - 1)];          \
    }                                                                                                                         \
  } while (0)
  if (EPI == SK_RESID && l15 < nv && !(PROBE & 4)) MH_LOAD_OLDH();   // requested before anything is waited for
  f32x4_t acc[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nkb = p.K / KB;

  // k-block of slot c of a pass that starts at kb0 (PRO_PLAIN: kb0 counts k-block PAIRS, see the header comment)
  auto kblock = [](int kb0, int c) -> int { return PRO == PRO_PLAIN ? 2 * (kb0 + NWV * (c >> 1)) + (c & 1) : kb0 + NWV * c; };
  int kb0 = wid;
  if constexpr (LINES) {
    const int npair = nkb >> 1;        // K is a multiple of 2 KB elements (checked on the host)
    unsigned char* patch = Lw + wid * PATCH;
    const int r8 = lane >> 3, c16 = (lane & 7) * 16;
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + c16;
    const long rx = (long)(r8 < p.B ? r8 : p.B - 1) * p.lda * (long)sizeof(T);
    const long ry = (long)(8 + r8 < p.B ? 8 + r8 : p.B - 1) * p.lda * (long)sizeof(T);
    for (int pw0 = wid; pw0 < npair + wid; pw0 += NWV * CP) {
      uint4 wv[CH], xa[CP], ya[CP];
#pragma unroll
      for (int c = 0; c < CH; ++c) wv[c] = make_uint4(0, 0, 0, 0);
      if (wload && !(PROBE & 2)) {   // ONE exec-masked region around all weight loads
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int pw = pw0 + NWV * (c >> 1);
          wv[c] = *reinterpret_cast<const uint4*>(Wp + (2 * (pw < npair ? pw : npair - 1) + (c & 1)) * KB);   // clamped address
        }
      }
#pragma unroll
      for (int cp = 0; cp < CP; ++cp) {
        const int pw = pw0 + NWV * cp;
        const long off = (long)(pw < npair ? pw : npair - 1) * 128;
        xa[cp] = *reinterpret_cast<const uint4*>(Ab + rx + off);
        ya[cp] = *reinterpret_cast<const uint4*>(Ab + ry + off);
      }
      MH_LOADS_ISSUED();
      MH_LOADS_ISSUED();
    MH_STAMP(KID, 0);   // loads issued
#pragma unroll
      for (int cp = 0; cp < CP; ++cp) {
        const uint32_t keep = (pw0 + NWV * cp < npair) ? 0xffffffffu : 0u;   // pairs beyond K contribute zeros
        *reinterpret_cast<uint4*>(patch + r8 * 144 + c16) = xa[cp];
        *reinterpret_cast<uint4*>(patch + (8 + r8) * 144 + c16) = ya[cp];
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // this lane's fragment (row l15, k group lg) of the pair's two k-blocks
          uint4 a = *reinterpret_cast<const uint4*>(patch + l15 * 144 + (j * 4 + lg) * 16);
          a = make_uint4(a.x & keep, a.y & keep, a.z & keep, a.w & keep);
          acc[0] = VecOps<T>::mma(a, wv[cp * 2 + j], acc[0]);
        }
      }
    }
  } else
  do {   // ONE pass for every RMSNorm shape and for K <= NWV * CH * KB; every wave runs at least one (its barrier)
    uint4 wv[CH];
    uint4 av[PRO == PRO_PLAIN ? CH : 1][PRO == PRO_PLAIN ? MF : 1];
    Raw hraw[NORM ? CH : 1][NORM ? MF : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) wv[c] = make_uint4(0, 0, 0, 0);
    if (wload && !(PROBE & 2)) {     // ONE exec-masked region around all weight loads (a branch per load would serialise them)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int kb = kblock(kb0, c);
        wv[c] = *reinterpret_cast<const uint4*>(Wp + (kb < nkb ? kb : nkb - 1) * KB);   // clamped address
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kb = kblock(kb0, c);
      const int kel = (kb < nkb ? kb : nkb - 1) * KB;
      if (PROBE & 1) {
#pragma unroll
        for (int f = 0; f < MF; ++f) {
          if (PRO == PRO_PLAIN) av[c][f] = make_uint4(kel, lane, c, f);
          else hraw[c][f] = VecOps<T>::load_raw(lnw + ((kel + lane) & 1020));
        }
      } else if (PRO == PRO_PLAIN && (PROBE & 128)) {   // timing only: the same bytes as whole 1 KB runs per wave instruction
#pragma unroll
        for (int f = 0; f < MF; ++f)
          av[c][f] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.A) + ((long)((c * NWV + wid) % (p.K * 16 / (64 * VEC))) * 64 + lane) * VEC);
      } else if (PRO == PRO_PLAIN) {
#pragma unroll
        for (int f = 0; f < MF; ++f)
          av[c][f] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.A) + (long)arow[f] * p.lda + kel + lg * VEC);
      } else if (!RLINES) {
#pragma unroll
        for (int f = 0; f < MF; ++f)
          hraw[c][f] = VecOps<T>::load_raw(reinterpret_cast<const float*>(p.A) + (long)arow[f] * p.lda + kel + lg * VEC);
      }
    }
    uint4 xr[RLINES ? CH : 1], yr[RLINES ? CH : 1];
    if (RLINES) {   // one 128-byte line of fp32 per row and k-block: rows 0..7 / 8..15 of the block as two whole-line loads
      const int r8 = lane >> 3;
      const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + (lane & 7) * 16;
      const long rx = (long)(r8 < p.B ? r8 : p.B - 1) * p.lda * 4, ry = (long)(8 + r8 < p.B ? 8 + r8 : p.B - 1) * p.lda * 4;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int kb = kb0 + NWV * c;
        const long off = (long)(kb < nkb ? kb : nkb - 1) * 128;
        xr[c] = *reinterpret_cast<const uint4*>(Ab + rx + off);
        yr[c] = *reinterpret_cast<const uint4*>(Ab + ry + off);
      }
    }
    MH_LOADS_ISSUED();
    MH_STAMP(KID, 0);   // loads issued
    if (RLINES) {
      unsigned char* patch = Lw + wid * PATCH;
      const int r8 = lane >> 3, c16 = (lane & 7) * 16;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        *reinterpret_cast<uint4*>(patch + r8 * 144 + c16) = xr[c];
        *reinterpret_cast<uint4*>(patch + (8 + r8) * 144 + c16) = yr[c];
        hraw[c][0] = VecOps<T>::load_raw(reinterpret_cast<const float*>(patch + l15 * 144 + lg * 32));
      }
    }
    float rsr[MF];
    if (LN) {
      // nn.LayerNorm statistics from the registers, two passes (mean, then the centred sum of squares -- the arithmetic of
      // F.layer_norm, no E[x^2] - mu^2 cancellation): lane -> the 4 lane groups of the wave -> the NWV waves, twice
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) q += (kb0 + NWV * c < nkb) ? VecOps<T>::sum(hraw[c][f]) : 0.f;
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (lg == 0) ssw[wid][f * 16 + l15] = q;
      }
      if (tid * 4 < p.K) { *reinterpret_cast<float4*>(lnw + tid * 4) = lnraw; *reinterpret_cast<float4*>(lnb + tid * 4) = lnbraw; }
      __syncthreads();
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        float su = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) su += ssw[w][f * 16 + l15];
        const float mu = su / (float)p.K;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          hraw[c][f] = VecOps<T>::centred(hraw[c][f], mu);
          q += (kb0 + NWV * c < nkb) ? VecOps<T>::sumsq(hraw[c][f]) : 0.f;
        }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (lg == 0) ssw2[wid][f * 16 + l15] = q;
      }
      __syncthreads();
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        float ss = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) ss += ssw2[w][f * 16 + l15];
        rsr[f] = rsqrtf(ss / (float)p.K + p.eps);
      }
    } else if (PRO == PRO_RMSNORM) {
      // RMSNorm statistics of the rows from the registers: lane -> the 4 lane groups of the wave -> the NWV waves
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) q += (kb0 + NWV * c < nkb) ? VecOps<T>::sumsq(hraw[c][f]) : 0.f;
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (lg == 0) ssw[wid][f * 16 + l15] = q;
      }
      if (tid * 4 < p.K) *reinterpret_cast<float4*>(lnw + tid * 4) = lnraw;
      __syncthreads();
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        float ss = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) ss += ssw[w][f * 16 + l15];
        rsr[f] = rsqrtf(ss / (float)p.K + p.eps);
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kbc = kblock(kb0, c);
      const uint32_t keep = (kbc < nkb) ? 0xffffffffu : 0u;   // k-blocks beyond K contribute zeros
      const int kc_el = ((kbc < nkb) ? kbc : nkb - 1) * KB + lg * VEC;
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        uint4 a;
        if (PRO == PRO_PLAIN) a = av[c][f];
        else if (LN) a = VecOps<T>::ln_frag(hraw[c][f], VecOps<T>::load_raw(lnw + kc_el), VecOps<T>::load_raw(lnb + kc_el), rsr[f]);
        else a = VecOps<T>::norm_frag(hraw[c][f], VecOps<T>::load_raw(lnw + kc_el), rsr[f]);
        a = make_uint4(a.x & keep, a.y & keep, a.z & keep, a.w & keep);
        acc[f] = VecOps<T>::mma(a, wv[c], acc[f]);
      }
    }
    kb0 += NWV * (PRO == PRO_PLAIN ? CP : CH);
  } while (kblock(kb0, 0) < nkb);

  if (!(PROBE & 8)) {
#pragma unroll
    for (int f = 0; f < MF; ++f) red[(wid * MF + f) * 64 + lane] = acc[f];
  }
  MH_STAMP(KID, 1);     // operands arrived, products done
  if (!(PROBE & 8)) __syncthreads();
  MH_STAMP(KID, 2);     // all waves done

  const int pos = (EPI == SK_QKV) ? *p.pos : 0;
  const float* redf = reinterpret_cast<const float*>(red);
#pragma unroll
  for (int u = 0; u < UPW; ++u) {
    const int unit = wid + u * NWV;
    if (unit >= MF * 4) break;
    const int ef = unit >> 2, r = unit & 3;
    float v = 0.f;
    if (PROBE & 8) {
      v = acc[0][0] + acc[MF - 1][3];
    } else {
#pragma unroll
      for (int w = 0; w < NWV; ++w) v += redf[((w * MF + ef) * 64 + lane) * 4 + r];    // wave order: deterministic
    }
    const int row = ef * 16 + lg * 4 + r;
    const bool ok = col_ok && row < p.B && (!(PROBE & 16) || p.B < 0);
    if (BIAS) v += bias_v;
    if (EPI == SK_GELU_ERF) {
      if (ok) store_wt(reinterpret_cast<T*>(p.out) + (long)row * p.ldo + ocol, Elem<T>::from_f32(0.5f * v * (1.0f + erff(v * 0.70710678118654752f))));
    } else if (EPI == SK_GEGLU) {
      const float lin = __shfl_down(v, 8, 16);   // the linear half of the pair sits 8 tile columns to the right
      if (ok) store_wt(reinterpret_cast<T*>(p.out) + (long)row * p.ldo + ocol, Elem<T>::from_f32(gelu_tanh(v) * lin));
    } else if (EPI == SK_STORE) {
      if (ok) store_wt(reinterpret_cast<T*>(p.out) + (long)row * p.ldo + ocol, Elem<T>::from_f32(v));
    } else if (EPI == SK_LOGITS) {
      if (ok) store_wt(reinterpret_cast<float*>(p.out) + (long)row * p.ldo + ocol, v);
    } else if (EPI == SK_RESID) {
      if (ok) store_wt(p.h + (long)row * p.ldh + ocol, oldh[u] + v);
    } else if (EPI == SK_QKV) {
      if (ok) {
        const int part = ocol / p.inner, c = ocol - part * p.inner;
        if (part == 0) {
          store_wt(reinterpret_cast<T*>(p.out) + (long)row * p.ldo + c, Elem<T>::from_f32(v));
        } else {
          const int hh = c >> 6, dd = c & 63;
          T* cache = reinterpret_cast<T*>(part == 1 ? p.kc : p.vc);
          store_wt(cache + (((long)row * p.H + hh) * p.tgt_len + pos) * 64 + dd, Elem<T>::from_f32(v));
        }
      }
    }
  }
  MH_STAMP(KID, 3);     // epilogue stores issued
#undef MH_LOAD_OLDH
}

// ---- single-query attention (online softmax in registers) -----------------------------------------
// lane = (key group g = lane>>3, dim chunk c = lane&7): 8 lanes cover the 64 dims of one key row
// (one 128-byte / 256-byte row per 8 lanes -> every wave load instruction is 8 consecutive rows).
struct Partial {
  float m, l;
  float acc[8];
};
__device__ inline void partial_init(Partial& s) {
  s.m = -1e30f; s.l = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s.acc[i] = 0.f;
}
template <typename T>
__device__ inline void partial_merge_groups(Partial& s) {  // across the 8 key groups of a wave
  // rescale to the wave-wide maximum first (ONE exponential per lane), then plain shuffle sums: no exponential sits in
  // the 3-level reduction chain
  float mw = s.m;
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
  const float f = fexp<T>(s.m - mw);
  s.l *= f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s.acc[i] *= f;
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    s.l += __shfl_xor(s.l, o, 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) s.acc[i] += __shfl_xor(s.acc[i], o, 64);
  }
  s.m = mw;
}

template <typename T, int U, typename E = T>
__device__ inline void attend_keys(Partial& st, const float (&q)[8], const E* kbase, const E* vbase, int j0,
                                   int jend, int jstride, const float* bias_row, int pos, const uint8_t* mask_row,
                                   int P, float scale, int jmin = 0) {   // keys below jmin are masked (sliding window)
  // processes keys j0, j0+jstride, ... < jend for this lane's key group, U at a time
  const int c8 = (threadIdx.x & 7) * 8;
  for (int j = j0; j < jend; j += jstride * U) {
    float s[U];
    float kv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = j + u * jstride;
      const int jc = jj < jend ? jj : j;
      load8_stream<E>(kbase + (long)jc * 64 + c8, kv[u]);
    }
    float vv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = j + u * jstride;
      const int jc = jj < jend ? jj : j;
      load8_stream<E>(vbase + (long)jc * 64 + c8, vv[u]);
    }
    // bias / mask values: wave-uniform branch on the pointers, unconditional loads from clamped indices
    float bv[U];
    int mv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { bv[u] = 0.f; mv[u] = 1; }
    if (bias_row) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = j + u * jstride;
        bv[u] = bias_row[pos - (jj < jend ? jj : j)];
      }
    }
    if (mask_row) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = j + u * jstride;
        const int jm = jj < P ? jj : P - 1;
        const int mval = mask_row[jm];
        mv[u] = (jj < P) ? mval : 1;
      }
    }
    float cmax = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = j + u * jstride;
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) d += q[i] * kv[u][i];
      d = group_sum<8>(d) * scale + bv[u];
      const bool ok = (jj < jend) && (jj >= jmin) && (mv[u] != 0);
      d = ok ? d : -INFINITY;
      s[u] = d;
      cmax = fmaxf(cmax, d);
    }
    const float mn = fmaxf(st.m, cmax);
    const float fa = fexp<T>(st.m - mn);
    st.l *= fa;
#pragma unroll
    for (int i = 0; i < 8; ++i) st.acc[i] *= fa;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float pu = fexp<T>(s[u] - mn);
      st.l += pu;
#pragma unroll
      for (int i = 0; i < 8; ++i) st.acc[i] += pu * vv[u][i];
    }
    st.m = mn;
  }
}

struct SelfAttnP {
  const void* q; int ldq;         // T [B, inner]
  const void* kc; const void* vc; // [B][H][tgt_len][64]
  const float* bias;              // fp32 [H][tgt_len] by distance pos - j
  const uint8_t* prompt_mask; int P;
  void* out; int ldo;             // T [B, inner]
  int B, H, tgt_len;
  const int* pos;
  // the Whisper family (kernel template WH; modeling_varwhisper.py:229-258,381-568): fp32 bias of this layer's fused
  // Wqkv [3 inner] or NULL; rotary table fp32 [tgt_len][64] = cos(32) | sin(32) of every position (built on the host with
  // the reference's formulas, already rounded to the storage type); score scale; window > 0: keys older than
  // pos - window are not attended (local layers under the flash-attention path, :330)
  const float* qkv_bias; const float* rope; float scale; int window;
};

// merge the 4 waves' partial (m, l, acc[64]) through LDS; threads 0..63 return the merged (m, l, a[d])
template <typename T, int NW = 4>
__device__ inline void block_merge(const Partial& st, float (*sm)[66], float& m, float& l, float& a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c8 = (lane & 7) * 8, g = lane >> 3;
  if (g == 0) {
    if (lane == 0) { sm[wid][0] = st.m; sm[wid][1] = st.l; }
#pragma unroll
    for (int i = 0; i < 8; ++i) sm[wid][2 + c8 + i] = st.acc[i];
  }
  __syncthreads();
  m = -1e30f; l = 0.f; a = 0.f;
  if (threadIdx.x < 64) {
    // two passes over the NW partials: the block-wide maximum, then NW INDEPENDENT exponentials (a running merge chains
    // 2 (NW - 1) dependent ones: ~0.7 us at NW = 16); wave order fixed => deterministic
    const int d = threadIdx.x;
    m = sm[0][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, sm[w][0]);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = fexp<T>(sm[w][0] - m);
      l += sm[w][1] * f;
      a += sm[w][2 + d] * f;
    }
  }
}

// one workgroup (4 waves) per (b, h); keys 0..pos interleaved over the 32 key groups of the block
template <typename T>
__global__ __launch_bounds__(256) void dec_self_attn_kernel(SelfAttnP p) {
  __shared__ float sm[4][66];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int pair = blockIdx.x;
  const int b = pair / p.H, h = pair % p.H;
  const int pos = *p.pos;
  const int c8 = (lane & 7) * 8, g = lane >> 3;
  float q[8];
  load8<T>(reinterpret_cast<const T*>(p.q) + (long)b * p.ldq + h * 64 + c8, q);
  const T* kb = reinterpret_cast<const T*>(p.kc) + ((long)b * p.H + h) * p.tgt_len * 64;
  const T* vb = reinterpret_cast<const T*>(p.vc) + ((long)b * p.H + h) * p.tgt_len * 64;
  Partial st;
  partial_init(st);
  attend_keys<T, 4>(st, q, kb, vb, wid * 8 + g, pos + 1, 32, p.bias + (long)h * p.tgt_len, pos,
                    p.prompt_mask ? p.prompt_mask + (long)b * p.P : nullptr, p.P, 1.0f);
  partial_merge_groups<T>(st);
  float m, l, a;
  block_merge<T>(st, sm, m, l, a);
  if (threadIdx.x < 64)   // (sm[0] was last read by this very wave)
    store_head_row_wt<T>(reinterpret_cast<T*>(p.out) + (long)b * p.ldo + h * 64, l > 0.f ? a / l : 0.f, &sm[0][0]);
}

struct CrossAttnP {
  const void* q; int ldq;   // T [B, inner]   (kernel without its own projection)
  const void* k; const void* v;  // this layer's [B][H][L][64]
  void* out; int ldo;       // T [B, inner]
  int B, H, L;
  int kv_B;                 // > 0: row b reads K/V row b % kv_B (CFG pairs share the encoder output);
                            // < 0: row b reads K/V row b / -kv_B (the beams of a chunk share it: rows are (chunk, beam))
  const float* kscale; const float* vscale;   // fp8 rows (kernel template F8): this layer's [kv rows][H] scales, else NULL
  // measurement only (mh_t5_decode_timing): [slots][2] = (earliest workgroup start, latest workgroup end) of THIS
  // launch in wall-clock ticks, slot = *pos * ts_layers + ts_layer for the first ts_ring positions; NULL in production
  unsigned long long* tstamp; const int* pos; int ts_ring, ts_layers, ts_layer;
  // the Whisper family (kernel template WH): fp32 bias of this layer's query projection [inner] or NULL, and the score scale
  const float* q_bias; float scale;
};

// (b, h) of workgroup `blk`; under CFG (B == 2 kv_B) the two rows that share K/V sit in adjacent workgroups
__device__ inline void cross_pair_of_block(const CrossAttnP& p, int blk, int& b, int& h) {
  if (p.kv_B > 0) {
    const int q2 = blk >> 1;
    b = (blk & 1) * p.kv_B + q2 / p.H;
    h = q2 % p.H;
  } else {
    b = blk / p.H;
    h = blk % p.H;
  }
}

// one 16-wave workgroup per (b, h): the waves interleave 8-key rows of the 1251 encoder keys, partial (max, sum, acc)
// merged through LDS in wave order -- the reduction order of a row never depends on the batch.  Used when the query was
// projected by its own GEMV (option decode_fused_proj = 0).
template <typename T, int U>
__global__ __launch_bounds__(1024) void dec_cross_attn_kernel(CrossAttnP p) {
  constexpr int NW = 16;
  __shared__ float sm[NW][66];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int b, h;
  cross_pair_of_block(p, blockIdx.x, b, h);
  const int c8 = (lane & 7) * 8, g = lane >> 3;
  float q[8];
  load8<T>(reinterpret_cast<const T*>(p.q) + (long)b * p.ldq + h * 64 + c8, q);
  const int kvb = p.kv_B > 0 ? b % p.kv_B : (p.kv_B < 0 ? b / -p.kv_B : b);
  const T* kb = reinterpret_cast<const T*>(p.k) + ((long)kvb * p.H + h) * p.L * 64;
  const T* vb = reinterpret_cast<const T*>(p.v) + ((long)kvb * p.H + h) * p.L * 64;
  Partial st;
  partial_init(st);
  attend_keys<T, U>(st, q, kb, vb, wid * 8 + g, p.L, 8 * NW, nullptr, 0, nullptr, 0, 1.0f);
  partial_merge_groups<T>(st);
  float m, l, a;
  block_merge<T, NW>(st, sm, m, l, a);
  if (threadIdx.x < 64)
    store_head_row_wt<T>(reinterpret_cast<T*>(p.out) + (long)b * p.ldo + h * 64, l > 0.f ? a / l : 0.f, &sm[0][0]);
}

// ---- attention kernels that project their own query (and the new self-attention key / value) --------------------
// A decode step is ~100 dependent kernels of a few microseconds each, so every kernel removed from the chain is worth
// more than the work it did.  The per-head projections are tiny (64 outputs x d inputs): each attention workgroup
// recomputes RMSNorm(h[b]) and its own 64-row slice of W (the slices of one head are shared by the B workgroups of
// that head through L2), which removes the stand-alone QKV and cross-Q GEMV launches from every layer.
struct HeadProjP {
  const float* h; int ldh;            // fp32 residual stream [B, ldh]
  const float* ln_w; float eps;
  const void* W; int ldw;             // [rows, ldw] element type T (cross: Wq [inner][d]; self: Wqkv [3 inner][d])
  int d;
  const float* ln_b;                  // kernel template LN (HF Whisper, library arch 2): affine nn.LayerNorm -- its bias; ln_w = its weight
};

template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 v;
  __device__ inline void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ inline void unpack(float (&o)[8]) const {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
};
template <> struct Raw8<float> {
  float4 a, b;
  __device__ inline void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ inline void unpack(float (&o)[8]) const {
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
};

// xn[k] = T-rounded ln_w[k] * (h[b][k] * rsqrt(mean(h[b]^2) + eps)) for k < d, by a 1024-thread workgroup (d <= 1024).
// The statistics come from the row itself (wave sums -> 16 LDS floats that every thread adds in the same order).
// Two halves: `issue` only REQUESTS the row and the weight (it needs nothing but preloaded kernel arguments), `finish`
// waits for them -- whatever else the kernel can request goes in between.
template <typename T, bool LN = false>
struct NormRow {
  float x, g, bb;
  __device__ inline void issue_weight(const HeadProjP& hp) {      // (independent of the predecessor)
    const int tid = threadIdx.x;
    g = hp.ln_w[tid < hp.d ? tid : hp.d - 1];
    if (LN) bb = hp.ln_b[tid < hp.d ? tid : hp.d - 1];
  }
  __device__ inline void issue_row(const HeadProjP& hp, int b) {   // the residual row the predecessor wrote
    const int tid = threadIdx.x;
    x = hp.h[(long)b * hp.ldh + (tid < hp.d ? tid : hp.d - 1)];
  }
  __device__ inline void issue(const HeadProjP& hp, int b) { issue_row(hp, b); issue_weight(hp); }
  __device__ inline void finish(const HeadProjP& hp, float* xn, float* red16) {
    const int tid = threadIdx.x;
    if (LN) {
      // nn.LayerNorm: mean first, then the centred sum of squares (two block reductions over the same 16 LDS floats)
      const float su = wave_sum(tid < hp.d ? x : 0.f);
      if ((tid & 63) == 0) red16[tid >> 6] = su;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) tot += red16[w];
      x -= tot / (float)hp.d;
      __syncthreads();      // (every thread has read red16 before it is written again)
    }
    const float sq = wave_sum(tid < hp.d ? x * x : 0.f);
    if ((tid & 63) == 0) red16[tid >> 6] = sq;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red16[w];
    const float rs = rsqrtf(tot / (float)hp.d + hp.eps);
    if (tid < hp.d) xn[tid] = Elem<T>::to_f32(Elem<T>::from_f32(LN ? x * rs * g + bb : g * (x * rs)));
    __syncthreads();
  }
};
template <typename T>
__device__ inline void norm_row_to_lds(const HeadProjP& hp, int b, float* xn, float* red16) {
  NormRow<T> nr;
  nr.issue(hp, b);
  nr.finish(hp, xn, red16);
}


// NP projections of 64 outputs each: out[p][o] = T-rounded sum_k xn[k] * W[(row0[p] + o) * ldw + k].
// 1024 threads: 16 consecutive lanes per output, KC 8-element chunks per lane (d = 128 KC), chunk c of lane ks = elements
// [128 c + 8 ks, + 8): the 16 lanes of an output read 256 (bf16) / 512 (fp32) CONTIGUOUS bytes per load instruction, whole
// 128-byte lines (lane-contiguous chunks -- 96-byte strides at d = 768 -- touch 12 lines per output row and instruction;
// the CU's load path is charged per line and lane).  The weight slice does not depend on the activations: `load` is
// called at kernel entry, `apply` after the normalised row is in LDS.
template <typename T, int KC, int NP>
struct HeadProj {
  Raw8<T> raw[NP][KC];
  __device__ inline void load(const HeadProjP& hp, const int (&row0)[NP]) {
    const int tid = threadIdx.x, o = tid >> 4, ks = tid & 15;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const T* wp = reinterpret_cast<const T*>(hp.W) + (long)(row0[q] + o) * hp.ldw + ks * 8;
#pragma unroll
      for (int c = 0; c < KC; ++c) raw[q][c].load(wp + c * 128);
    }
    MH_LOADS_ISSUED();
  }
  __device__ inline void load3(const HeadProjP& hp, const int (&row0)[3]) {   // (NP == 3 instantiations only)
    const int tid = threadIdx.x, o = tid >> 4, ks = tid & 15;
#pragma unroll
    for (int q = 0; q < (NP < 3 ? NP : 3); ++q) {
      const T* wp = reinterpret_cast<const T*>(hp.W) + (long)(row0[q] + o) * hp.ldw + ks * 8;
#pragma unroll
      for (int c = 0; c < KC; ++c) raw[q][c].load(wp + c * 128);
    }
  }
  // bias (fp32, indexed like the weight rows) or NULL: out = T-rounded (acc + bias[row0 + o]) -- nn.Linear with bias
  __device__ inline void apply(const float* xn, float (*out)[64], const float* bias = nullptr, const int* row0 = nullptr) const {
    const int tid = threadIdx.x, o = tid >> 4, ks = tid & 15;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        float w[8];
        raw[q][c].unpack(w);
        const float4 x0 = *reinterpret_cast<const float4*>(xn + c * 128 + ks * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(xn + c * 128 + ks * 8 + 4);
        acc += x0.x * w[0] + x0.y * w[1] + x0.z * w[2] + x0.w * w[3] + x1.x * w[4] + x1.y * w[5] + x1.z * w[6] + x1.w * w[7];
      }
      acc = group_sum<16>(acc);
      if (bias) acc += bias[row0[q] + o];
      if (ks == 0) out[q][o] = Elem<T>::to_f32(Elem<T>::from_f32(acc));
    }
  }
};

// cross-attention of one (b, h) with its own query projection; 16 waves, one key split (the default configuration
// of dec_cross_attn_kernel, same key interleave and merge order)
// F8: K / V are the e4m3 copy (64-byte rows, 8 bytes per lane; the scales multiply the scores and the output)
template <typename T, int KC, int U, bool F8 = false, bool WH = false, bool LN = false>
#ifndef MH_CROSS_WPE_MIN
#define MH_CROSS_WPE_MIN 8     // (A/B builds: 4 = up to 128 VGPRs, one 16-wave workgroup per CU)
#endif
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(MH_CROSS_WPE_MIN, 8)))   // <= 64 VGPRs: 2 workgroups per CU
void dec_cross_attn_q_kernel(const float* h_, const float* lnw_, const void* W_, const void* k_, const void* v_, int H_, int L_, int d_,
                             int kvB_, CrossAttnP p, HeadProjP hp) {   // leading scalars: preloaded kernel arguments (see gemv_kernel)
  hp.h = h_; hp.ln_w = lnw_; hp.W = W_; hp.ldh = d_; hp.ldw = d_; hp.d = d_;
  p.k = k_; p.v = v_; p.H = H_; p.L = L_; p.kv_B = kvB_;
  constexpr int NW = 16;
  __shared__ float sm[NW][66];
  __shared__ __attribute__((aligned(16))) float xn[1024];
  __shared__ float qs[1][64];
  __shared__ float red16[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int b, h;
  cross_pair_of_block(p, blockIdx.x, b, h);
  MH_STAMP0();
  const int c8 = (lane & 7) * 8, g = lane >> 3;
  const int row0[1] = {h * 64};
  const int kvb = p.kv_B > 0 ? b % p.kv_B : (p.kv_B < 0 ? b / -p.kv_B : b);
  typedef typename std::conditional<F8, fp8_t, T>::type E;
  const E* kb = reinterpret_cast<const E*>(p.k) + ((long)kvb * p.H + h) * p.L * 64;
  const E* vb = reinterpret_cast<const E*>(p.v) + ((long)kvb * p.H + h) * p.L * 64;
  HeadProj<T, KC, 1> proj;
  NormRow<T, LN> nrow;
  // everything the prologue needs is requested at once, from preloaded arguments only: the residual row, the RMSNorm
  // weight and (bf16: fp32 -- the parity path -- would not fit the 64-register budget of two workgroups per CU) this
  // head's 64 x d slice of Wq, which does not depend on the activations; then the remaining kernel arguments
  nrow.issue(hp, b);
  if (sizeof(T) == 2) proj.load(hp, row0);
  // (LDS-DMA prefetch of every wave's first 1-3 key iterations during this prologue -- 32 KB of LDS per iteration, issued as
  // inline asm behind shadow loads so that no wait of the prologue covers it -- was built and measured: 703 / 703 / 729 us
  // per token step for 1 / 2 / 3 iterations against 679 without.  Not kept.  Round 5, the same idea through REGISTERS (every lane
  // group's first key / value row, 32 bytes per lane, requested beside the prologue's operands): 285.8-286.5 ms per batch against
  // 276.7-277.0 -- the stream's requests queue in front of the weight slice the prologue waits for.  Not kept either.)
  unsigned long long t_start = 0;
  if (p.tstamp && threadIdx.x == 0) t_start = (unsigned long long)wall_clock64();
  nrow.finish(hp, xn, red16);
  MH_STAMP(KID_CROSS, 0);   // row normalised
  if (sizeof(T) != 2) proj.load(hp, row0);
  proj.apply(xn, qs, WH ? p.q_bias : nullptr, row0);
  __syncthreads();
  MH_STAMP(KID_CROSS, 1);   // query projected
  float q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = qs[0][c8 + i];
  const float ks = (F8 ? p.kscale[kvb * p.H + h] : 1.0f) * (WH ? p.scale : 1.0f), vs = F8 ? p.vscale[kvb * p.H + h] : 1.0f;
  Partial st;
  partial_init(st);
  attend_keys<T, U, E>(st, q, kb, vb, wid * 8 + g, p.L, 8 * NW, nullptr, 0, nullptr, 0, ks);
  MH_STAMP(KID_CROSS, 2);   // keys streamed (this wave)
  partial_merge_groups<T>(st);
  float m, l, a;
  block_merge<T, NW>(st, sm, m, l, a);
  MH_STAMP(KID_CROSS, 3);   // partials merged
  if (threadIdx.x < 64)
    store_head_row_wt<T>(reinterpret_cast<T*>(p.out) + (long)b * p.ldo + h * 64, l > 0.f ? a / l * vs : 0.f, &sm[0][0]);
  if (p.tstamp && threadIdx.x == 0 && *p.pos < p.ts_ring) {      // the first ts_ring positions of the call, one slot each
    unsigned long long* slot = p.tstamp + 2 * ((long)(*p.pos) * p.ts_layers + p.ts_layer);
    atomicMin(slot, t_start);
    atomicMax(slot + 1, (unsigned long long)wall_clock64());
  }
}

// self-attention of one (b, h) with its own q / k / v projections: appends the new key / value row to the caches and
// attends over keys 0 .. pos-1 from the cache plus the new key straight from LDS (merged last)
template <typename T, int KC, bool WH = false, bool LN = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))   // 16 waves = 4 per SIMD: the whole 128-register budget
void dec_self_attn_qkv_kernel(const float* h_, const float* lnw_, const void* W_, const int* pos_,
                                                                 const void* kc_, const void* vc_, int H_, int d_, SelfAttnP p,
                                                                 HeadProjP hp) {   // leading scalars: preloaded kernel arguments
  hp.h = h_; hp.ln_w = lnw_; hp.W = W_; hp.ldh = d_; hp.ldw = d_; hp.d = d_;
  p.pos = pos_; p.kc = kc_; p.vc = vc_; p.H = H_;
  const int inner = H_ * 64;
  constexpr int NW = 16;
  __shared__ float sm[NW][66];
  __shared__ __attribute__((aligned(16))) float xn[1024];
  __shared__ float qkv[3][64];
  __shared__ float red16[16];
  MH_STAMP0();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int c8 = (lane & 7) * 8, g = lane >> 3;
  // the first round trip carries everything that does not depend on another load: the residual row + RMSNorm weight, the
  // position, and the two loop-invariant values of the tail (the bias at distance 0 and the prompt mask of the new key:
  // they used to be two serial round trips at the END of the kernel)
  NormRow<T, LN> nrow;
  nrow.issue(hp, b);
  const int pos = *p.pos;
  constexpr bool kAllAtOnce = sizeof(T) == 2 && KC <= 7;
  HeadProj<T, KC, kAllAtOnce ? 3 : 1> proj;
  const int row03[3] = {h * 64, inner + h * 64, 2 * inner + h * 64};
  const float* bias_row = WH ? nullptr : p.bias + (long)h * p.tgt_len;       // (the Whisper family has no additive bias)
  const uint8_t* mask_row = p.prompt_mask ? p.prompt_mask + (long)b * p.P : nullptr;
  const float bias0 = WH ? 0.f : bias_row[0];
  float rope_c = 1.f, rope_s = 0.f;
  if (WH) {   // cos / sin of this position for rotary pair (threadIdx.x & 31)
    rope_c = p.rope[(long)pos * 64 + (threadIdx.x & 31)];
    rope_s = p.rope[(long)pos * 64 + 32 + (threadIdx.x & 31)];
  }
  const int new_key_mask = (mask_row && pos < p.P) ? (int)mask_row[pos < p.P ? pos : 0] : 1;
  nrow.finish(hp, xn, red16);
  const float* qb = WH ? p.qkv_bias : nullptr;
  if (kAllAtOnce) {
    proj.load3(hp, row03);   // (requested AFTER the normalisation: holding the 18 vectors across it measured 688 vs 677 us per token step)
    proj.apply(xn, qkv, qb, row03);
  } else {   // fp32 storage, or d_model = 1024 in bf16: one projection at a time (register budget of a 1024-thread workgroup)
#pragma unroll
    for (int q3 = 0; q3 < 3; ++q3) {
      const int row0[1] = {q3 * inner + h * 64};
      HeadProj<T, KC, 1> proj1;
      proj1.load(hp, row0);
      proj1.apply(xn, qkv + q3, qb, row0);
    }
  }
  __syncthreads();
  if (WH) {
    // rotate-half RoPE on q and k (apply_rotary_pos_emb, modeling_varwhisper.py:236-258): pair (i, i + 32) of a head ->
    // (x1 cos - x2 sin, x2 cos + x1 sin), T-rounded; threads 0..63 = q, 64..127 = k
    float rot = 0.f;
    const int t = threadIdx.x;
    if (t < 128) {
      const float* x = qkv[t >> 6];
      const int i = t & 31;
      const float x1 = x[i], x2 = x[i + 32];
      rot = (t & 32) ? x2 * rope_c + x1 * rope_s : x1 * rope_c - x2 * rope_s;
    }
    __syncthreads();
    if (t < 128) qkv[t >> 6][t & 63] = Elem<T>::to_f32(Elem<T>::from_f32(rot));
    __syncthreads();
  }
  MH_STAMP(KID_SELF, 0);    // q / k / v projected
  T* kcache = reinterpret_cast<T*>(const_cast<void*>(p.kc)) + ((long)b * p.H + h) * p.tgt_len * 64;
  T* vcache = reinterpret_cast<T*>(const_cast<void*>(p.vc)) + ((long)b * p.H + h) * p.tgt_len * 64;
  store_head_row_from_lds_wt<T>(kcache + (long)pos * 64, qkv[1], (int)threadIdx.x - 64);     // waves 1 / 2: whole 16-byte pieces
  store_head_row_from_lds_wt<T>(vcache + (long)pos * 64, qkv[2], (int)threadIdx.x - 128);
  float q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = qkv[0][c8 + i];
  Partial st;
  partial_init(st);
  const float sc = WH ? p.scale : 1.0f;
  int j_first = 0;
  if (WH && p.window > 0 && pos - p.window > 0) j_first = (pos - p.window) & ~(8 * NW - 1);   // whole key iterations below the window are skipped
  attend_keys<T, 2>(st, q, kcache, vcache, j_first + wid * 8 + g, pos, 8 * NW, bias_row, pos, mask_row, p.P, sc,
                    (WH && p.window > 0) ? pos - p.window : 0);
  MH_STAMP(KID_SELF, 1);    // cached keys attended (this wave)
  partial_merge_groups<T>(st);
  float m, l, a;
  block_merge<T, NW>(st, sm, m, l, a);
  MH_STAMP(KID_SELF, 2);    // partials merged
  if (threadIdx.x < 64) {
    const int d = threadIdx.x;
    float sn = wave_sum(qkv[0][d] * qkv[1][d]) * sc + bias0;
    if (new_key_mask == 0) sn = -INFINITY;
    const float mn = fmaxf(m, sn);
    const float fa = fexp<T>(m - mn), fb = fexp<T>(sn - mn);
    l = l * fa + fb;
    a = a * fa + qkv[2][d] * fb;
    store_head_row_wt<T>(reinterpret_cast<T*>(p.out) + (long)b * p.ldo + h * 64, l > 0.f ? a / l : 0.f, &sm[0][0]);
  }
}

}  // namespace dec
}  // namespace mh
