// Device kernels of the KV-cached autoregressive T5 decode step (K5 / K6).
//
// One decode step of one layer is: [RMSNorm + QKV GEMV -> q, self-KV append] -> self-attention over the
// cache -> [O GEMV + residual] -> [RMSNorm + cross-Q GEMV] -> cross-attention over the 1251 encoder
// keys (THE HBM-bound kernel: B*H*L*64*2 elements per layer per step) -> [O GEMV + residual] ->
// [RMSNorm + wi GEMV + gated GELU] -> [wo GEMV + residual]; then final RMSNorm + lm_head GEMV,
// logits processors + token selection.  Every kernel reads the current position from device memory so
// the whole step is one replayable hipGraph.
//
// "Skinny" GEMVs (M = batch <= 64 rows) stream each weight row exactly once: a workgroup owns NS
// 16-column strips of W, its 4 waves split K, MFMA 16x16 atoms do the (rows x 16) products with the
// weight fragment loaded straight from HBM into registers (no LDS round trip for data used once), and
// the 4 partial accumulators are combined through LDS in a fixed order (deterministic).
#pragma once
#include <type_traits>

#include "internal.hpp"

namespace mh {
namespace dec {

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

// ---- skinny GEMM ----------------------------------------------------------------------------
// Every kernel has exactly ONE dependent memory round trip: weights, activations, the old residual
// values and the RMSNorm statistics are all requested before anything is waited for.
//   RMSNorm without a grid-wide pass: the kernel that PRODUCES the residual stream (a RESID GEMV or the
//   sampler's embedding gather) also emits per-row partial sums of squares (one per 16-column strip,
//   fixed order => deterministic); the consumer adds the <= 64 partials, takes rsqrt and normalises its
//   fp32 A fragments in registers.
enum { PRO_PLAIN = 0, PRO_RMSNORM = 1 };
enum { SK_STORE = 0, SK_QKV = 1, SK_GEGLU = 2, SK_RESID = 3, SK_LOGITS = 4 };

struct SkinnyP {
  const void* A; int lda;      // PRO_PLAIN: T [B, lda];  PRO_RMSNORM: fp32 residual stream [B, lda]
  const float* ln_w; float eps;
  const float* ss_in; int ss_parts;   // PRO_RMSNORM: [ss_parts][64] partial sums of squares of the rows of A
  const void* W; int ldw;      // [N, ldw] element type T
  int B, N, K;
  void* out; int ldo;          // STORE: T [B, ldo]; GEGLU: T [B, ldo] (N/2 cols); LOGITS: f32 [B, ldo]
  float* h; int ldh;           // RESID: h[b][n] += acc
  float* ss_out;               // RESID: [N/16][64] partial sums of squares of the updated rows
  void* kc; void* vc;          // QKV: this layer's self-attention caches [B][H][tgt_len][64]
  int H, tgt_len, inner;
  const int* pos;
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
  static constexpr int kRawRegs = 8;
};
template <> struct VecOps<float> {
  struct Raw { float4 a; };
  __device__ static inline Raw load_raw(const float* p) {
    Raw r;
    r.a = *reinterpret_cast<const float4*>(p);
    return r;
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
  static constexpr int kRawRegs = 4;
};

// k-blocks per wave whose loads are all in flight at once, bounded by the register budget
template <typename T, int MF, int NS, int PRO>
constexpr int skinny_chunk() {
  if (PRO == PRO_PLAIN) return 8;
  const int per = 4 * NS + VecOps<T>::kRawRegs * (MF + 1);
  return 8 * per <= 200 ? 8 : (6 * per <= 200 ? 6 : (4 * per <= 200 ? 4 : 2));
}

// NWV waves per workgroup split K (4, or 8 for the long-K PLAIN GEMVs so that one pass covers K = 2048)
template <typename T, int MF, int NS, int PRO, int EPI, int NWV = 4>
__global__ __launch_bounds__(NWV * 64) void skinny_gemm_kernel(SkinnyP p) {
  static_assert(PRO == PRO_PLAIN || NWV == 4, "the RMSNorm prologue is written for 4 waves");
  constexpr int VEC = Elem<T>::kVec;   // elements per 16-byte vector (per lane per k-block)
  constexpr int KB = 4 * VEC;          // k elements per k-block (4 lane groups x 16 B)
  constexpr int CH = skinny_chunk<T, MF, NS, PRO>();
  constexpr int NSE = (EPI == SK_GEGLU) ? 1 : NS;
  typedef typename VecOps<T>::Raw Raw;
  __shared__ float ssp[4][64];
  __shared__ f32x4_t red[NWV * NS * MF * 64];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int strip0 = blockIdx.x * NS;
  // epilogue role of this thread: one f32x4 accumulator vector (4 rows x 1 column)
  const int ef = (tid >> 6) % MF, es = (tid >> 6) / MF;
  const bool epi_thread = tid < NSE * MF * 64;

  // ---- requests that do not depend on anything: old residual values, RMSNorm statistics ----
  float oldh[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI == SK_RESID && epi_thread) {
    const int colc = (strip0 + es) * 16 + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = ef * 16 + lg * 4 + r;
      oldh[r] = p.h[(long)(row < p.B ? row : p.B - 1) * p.ldh + (colc < p.N ? colc : p.N - 1)];
    }
  }

  f32x4_t acc[NS][MF];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int f = 0; f < MF; ++f) acc[s][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const T* Wp[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    int wr = (strip0 + s) * 16 + l15;
    wr = wr < p.N ? wr : p.N - 1;
    Wp[s] = reinterpret_cast<const T*>(p.W) + (long)wr * p.ldw + lg * VEC;
  }
  int arow[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) arow[f] = (f * 16 + l15) < p.B ? (f * 16 + l15) : p.B - 1;

  const int nkb = p.K / KB;
  float rsr[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) rsr[f] = 0.f;

  // wave w owns k-blocks w, w+4, w+8, ...; CH of them per pass, every load issued before the first use.
  // All addresses are clamped and the loads unconditional (a predicated load in an unrolled loop makes
  // hipcc branch around each load and wait for it: one L2 round trip per element).
  auto chunk = [&](int kb0, auto first_tag) {
    constexpr bool kFirst = decltype(first_tag)::value;
    uint4 wv[CH][NS];
    uint4 av[CH][MF];
    Raw hraw[PRO == PRO_RMSNORM ? CH : 1][PRO == PRO_RMSNORM ? MF : 1];
    Raw graw[PRO == PRO_RMSNORM ? CH : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kb = kb0 + NWV * c;
      const int kel = (kb < nkb ? kb : (wid < nkb ? wid : 0)) * KB;
#pragma unroll
      for (int s = 0; s < NS; ++s) wv[c][s] = *reinterpret_cast<const uint4*>(Wp[s] + kel);
      if (PRO == PRO_PLAIN) {
#pragma unroll
        for (int f = 0; f < MF; ++f)
          av[c][f] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.A) + (long)arow[f] * p.lda + kel + lg * VEC);
      } else {
        graw[c] = VecOps<T>::load_raw(p.ln_w + kel + lg * VEC);
#pragma unroll
        for (int f = 0; f < MF; ++f)
          hraw[c][f] = VecOps<T>::load_raw(reinterpret_cast<const float*>(p.A) + (long)arow[f] * p.lda + kel + lg * VEC);
      }
    }
    if (PRO == PRO_RMSNORM && kFirst) {
      // RMSNorm statistics: requested AFTER the operand loads (in program order) so that the one wait below
      // covers everything with a single round trip
      {
        const int row = tid & 63, pg = tid >> 6;
        float sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int part = pg + 4 * i;
          sv[i] = p.ss_in[(part < p.ss_parts ? part : 0) * 64 + row];
        }
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += (pg + 4 * i < p.ss_parts) ? sv[i] : 0.f;
        ssp[pg][row] = a;
      }
      __syncthreads();   // ssp complete (every wave reaches this exactly once)
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const int row = f * 16 + l15;
        const float ss = (ssp[0][row] + ssp[1][row]) + (ssp[2][row] + ssp[3][row]);
        rsr[f] = rsqrtf(ss / (float)p.K + p.eps);
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (kb0 + NWV * c < nkb) {
#pragma unroll
        for (int f = 0; f < MF; ++f) {
          uint4 a;
          if (PRO == PRO_PLAIN) a = av[c][f];
          else a = VecOps<T>::norm_frag(hraw[c][f], graw[c], rsr[f]);
          if (f * 16 + l15 >= p.B) a = make_uint4(0, 0, 0, 0);   // rows beyond the batch contribute zeros
#pragma unroll
          for (int s = 0; s < NS; ++s) acc[s][f] = VecOps<T>::mma(a, wv[c][s], acc[s][f]);
        }
      }
    }
  };
  chunk(wid, std::true_type{});
  for (int kb0 = wid + NWV * CH; kb0 < nkb; kb0 += NWV * CH) chunk(kb0, std::false_type{});

#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int f = 0; f < MF; ++f) red[((wid * NS + s) * MF + f) * 64 + lane] = acc[s][f];
  __syncthreads();
  if (!epi_thread) return;

  const int pos = (EPI == SK_QKV) ? *p.pos : 0;
  f32x4_t v = red[((0 * NS + es) * MF + ef) * 64 + lane];
#pragma unroll
  for (int w = 1; w < NWV; ++w) {
    const f32x4_t t = red[((w * NS + es) * MF + ef) * 64 + lane];
    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
  }
  f32x4_t u = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (EPI == SK_GEGLU) {
    u = red[((0 * NS + 1) * MF + ef) * 64 + lane];
#pragma unroll
    for (int w = 1; w < NWV; ++w) {
      const f32x4_t t = red[((w * NS + 1) * MF + ef) * 64 + lane];
      u[0] += t[0]; u[1] += t[1]; u[2] += t[2]; u[3] += t[3];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = ef * 16 + lg * 4 + r;
    const bool rok = row < p.B;
    if (EPI == SK_GEGLU) {
      const int col = (strip0 / 2) * 16 + l15;
      if (rok && col < p.N / 2)
        reinterpret_cast<T*>(p.out)[(long)row * p.ldo + col] = Elem<T>::from_f32(gelu_tanh(v[r]) * u[r]);
      continue;
    }
    const int col = (strip0 + es) * 16 + l15;
    const bool ok = rok && col < p.N;
    if (EPI == SK_STORE) {
      if (ok) reinterpret_cast<T*>(p.out)[(long)row * p.ldo + col] = Elem<T>::from_f32(v[r]);
    } else if (EPI == SK_LOGITS) {
      if (ok) reinterpret_cast<float*>(p.out)[(long)row * p.ldo + col] = v[r];
    } else if (EPI == SK_RESID) {
      const float hn = oldh[r] + v[r];
      if (ok) p.h[(long)row * p.ldh + col] = hn;
      // partial sum of squares of this strip for the next RMSNorm (all 64 lanes take part in the shuffle)
      const float sq = group_sum<16>(ok ? hn * hn : 0.f);
      if (l15 == 0) p.ss_out[(long)(strip0 + es) * 64 + row] = sq;
    } else if (EPI == SK_QKV) {
      if (ok) {
        const int part = col / p.inner, c = col - part * p.inner;
        if (part == 0) {
          reinterpret_cast<T*>(p.out)[(long)row * p.ldo + c] = Elem<T>::from_f32(v[r]);
        } else {
          const int hh = c >> 6, dd = c & 63;
          T* cache = reinterpret_cast<T*>(part == 1 ? p.kc : p.vc);
          cache[(((long)row * p.H + hh) * p.tgt_len + pos) * 64 + dd] = Elem<T>::from_f32(v[r]);
        }
      }
    }
  }
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
__device__ inline void partial_merge(Partial& a, float m2, float l2, const float (&acc2)[8]) {
  const float mn = fmaxf(a.m, m2);
  const float fa = fexp<T>(a.m - mn), fb = fexp<T>(m2 - mn);
  a.l = a.l * fa + l2 * fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) a.acc[i] = a.acc[i] * fa + acc2[i] * fb;
  a.m = mn;
}
template <typename T>
__device__ inline void partial_merge_groups(Partial& s) {  // across the 8 key groups of a wave
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(s.m, o, 64), l2 = __shfl_xor(s.l, o, 64);
    float a2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a2[i] = __shfl_xor(s.acc[i], o, 64);
    partial_merge<T>(s, m2, l2, a2);
  }
}

template <typename T, int U>
__device__ inline void attend_keys(Partial& st, const float (&q)[8], const T* kbase, const T* vbase, int j0,
                                   int jend, int jstride, const float* bias_row, int pos, const uint8_t* mask_row,
                                   int P, float scale) {
  // processes keys j0, j0+jstride, ... < jend for this lane's key group, U at a time
  const int c8 = (threadIdx.x & 7) * 8;
  for (int j = j0; j < jend; j += jstride * U) {
    float s[U];
    float kv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = j + u * jstride;
      const int jc = jj < jend ? jj : j;
      load8_stream<T>(kbase + (long)jc * 64 + c8, kv[u]);
    }
    float vv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = j + u * jstride;
      const int jc = jj < jend ? jj : j;
      load8_stream<T>(vbase + (long)jc * 64 + c8, vv[u]);
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
      const bool ok = (jj < jend) && (mv[u] != 0);
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
    const int d = threadIdx.x;
    m = sm[0][0]; l = sm[0][1]; a = sm[0][2 + d];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float m2 = sm[w][0], l2 = sm[w][1], a2 = sm[w][2 + d];
      const float mn = fmaxf(m, m2);
      const float fa = fexp<T>(m - mn), fb = fexp<T>(m2 - mn);
      l = l * fa + l2 * fb;
      a = a * fa + a2 * fb;
      m = mn;
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
  if (threadIdx.x < 64)
    reinterpret_cast<T*>(p.out)[(long)b * p.ldo + h * 64 + threadIdx.x] = Elem<T>::from_f32(l > 0.f ? a / l : 0.f);
}

struct CrossAttnP {
  const void* q; int ldq;   // T [B, inner]
  const void* k; const void* v;  // this layer's [B][H][L][64]
  void* out; int ldo;       // T [B, inner] (splits == 1)
  float* part;              // fp32 [B][H][splits][66] = (m, l, acc[64]) (splits > 1)
  int* ticket;              // [B*H] arrival counters, zero between launches; non-null: the last-arriving split
                            // of a (b, h) pair merges the partials in-kernel (no separate merge launch)
  int B, H, L, splits;
  int kv_B;                 // > 0: row b reads K/V row b % kv_B (CFG pairs share the encoder output)
};

// one workgroup (NW waves) per (b, h, split); the waves interleave 8-key rows of the split's key range.
// NW = 4 with 4 key splits (+ merge kernel) or NW = 16 with one split (no merge launch): both keep the
// reduction order of a row independent of the batch.
template <typename T, int NW, int U = 4>
__global__ __launch_bounds__(NW * 64) void dec_cross_attn_kernel(CrossAttnP p) {
  __shared__ float sm[NW][66];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int split = blockIdx.x % p.splits;
  const int blk = blockIdx.x / p.splits;
  int b, h;
  if (p.kv_B > 0) {   // CFG (B == 2 kv_B): the two rows that share K/V sit in adjacent workgroups
    const int q2 = blk >> 1;
    b = (blk & 1) * p.kv_B + q2 / p.H;
    h = q2 % p.H;
  } else {
    b = blk / p.H;
    h = blk % p.H;
  }
  const int pair = b * p.H + h;
  const int c8 = (lane & 7) * 8, g = lane >> 3;
  const int per = (p.L + p.splits - 1) / p.splits;
  const int k_lo = split * per;
  int k_hi = k_lo + per;
  k_hi = k_hi < p.L ? k_hi : p.L;
  float q[8];
  load8<T>(reinterpret_cast<const T*>(p.q) + (long)b * p.ldq + h * 64 + c8, q);
  const int kvb = p.kv_B > 0 ? b % p.kv_B : b;   // row b reads K/V row b % kv_B
  const T* kb = reinterpret_cast<const T*>(p.k) + ((long)kvb * p.H + h) * p.L * 64;
  const T* vb = reinterpret_cast<const T*>(p.v) + ((long)kvb * p.H + h) * p.L * 64;
  Partial st;
  partial_init(st);
  attend_keys<T, U>(st, q, kb, vb, k_lo + wid * 8 + g, k_hi, 8 * NW, nullptr, 0, nullptr, 0, 1.0f);
  partial_merge_groups<T>(st);
  float m, l, a;
  block_merge<T, NW>(st, sm, m, l, a);
  if (threadIdx.x < 64) {
    const int d = threadIdx.x;
    if (p.splits == 1) {
      reinterpret_cast<T*>(p.out)[(long)b * p.ldo + h * 64 + d] = Elem<T>::from_f32(l > 0.f ? a / l : 0.f);
    } else if (p.ticket) {
      float* pp = p.part + ((long)pair * p.splits + split) * 66;
      if (d == 0) {
        __hip_atomic_store(pp + 0, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(pp + 2 + d, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      float* pp = p.part + ((long)pair * p.splits + split) * 66;
      if (d == 0) { pp[0] = m; pp[1] = l; }
      pp[2 + d] = a;
    }
  }
  if (p.splits > 1 && p.ticket) {
    // In-kernel merge by the last-arriving split of a (b, h) pair.  Hand-off in the write-through form of
    // cdna_hip_programming.md guideline 16: the 66-float partial is stored with sc1 (agent-scope relaxed atomic
    // stores lower to `global_store_dword sc1`), every wave drains its stores, one relaxed agent-scope ticket is
    // taken, and the block that draws the last ticket reads all partials with sc1 loads (L1 bypass).  No L2
    // write-back fence (an agent-scope release in each of the ~800 workgroups cost far more than a merge launch).
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = __hip_atomic_fetch_add(p.ticket + pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (t == p.splits - 1) ? 1 : 0;
      if (last) __hip_atomic_store(p.ticket + pair, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
      s_last = last;
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) {
      const int d = threadIdx.x;
      const float* pp = p.part + (long)pair * p.splits * 66;
      float mm = __hip_atomic_load(pp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float ll = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float aa = __hip_atomic_load(pp + 2 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int s2 = 1; s2 < p.splits; ++s2) {   // fixed split order: deterministic
        const float* ps = pp + s2 * 66;
        const float m2 = __hip_atomic_load(ps + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float l2 = __hip_atomic_load(ps + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float a2 = __hip_atomic_load(ps + 2 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float mn = fmaxf(mm, m2);
        const float fa = fexp<T>(mm - mn), fb = fexp<T>(m2 - mn);
        ll = ll * fa + l2 * fb;
        aa = aa * fa + a2 * fb;
        mm = mn;
      }
      reinterpret_cast<T*>(p.out)[(long)b * p.ldo + h * 64 + d] = Elem<T>::from_f32(ll > 0.f ? aa / ll : 0.f);
    }
  }
}

// ---- attention kernels that project their own query (and the new self-attention key / value) --------------------
// A decode step is ~100 dependent kernels of a few microseconds each, so every kernel removed from the chain is worth
// more than the work it did.  The per-head projections are tiny (64 outputs x d inputs): each attention workgroup
// recomputes RMSNorm(h[b]) and its own 64-row slice of W (the slices of one head are shared by the B workgroups of
// that head through L2), which removes the stand-alone QKV and cross-Q GEMV launches from every layer.
struct HeadProjP {
  const float* h; int ldh;            // fp32 residual stream [B, ldh]
  const float* ln_w; float eps;
  const float* ss_in; int ss_parts;   // [ss_parts][64] partial sums of squares of the rows of h
  const void* W; int ldw;             // [rows, ldw] element type T (cross: Wq [inner][d]; self: Wqkv [3 inner][d])
  int d;
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

// xn[k] = T-rounded ln_w[k] * (h[b][k] * rsqrt(mean(h[b]^2) + eps)) for k < d, by a 1024-thread workgroup (d <= 1024);
// the operand loads are requested before the statistics are waited for.  `stat` is one LDS float.
template <typename T>
__device__ inline void norm_row_to_lds(const HeadProjP& hp, int b, float* xn, float* stat) {
  const int tid = threadIdx.x;
  const int kc = tid < hp.d ? tid : hp.d - 1;
  const float x = hp.h[(long)b * hp.ldh + kc];
  const float g = hp.ln_w[kc];
  if (tid < 64) {
    const float sv = hp.ss_in[(tid < hp.ss_parts ? tid : 0) * 64 + b];
    const float tot = wave_sum(tid < hp.ss_parts ? sv : 0.f);
    if (tid == 0) *stat = rsqrtf(tot / (float)hp.d + hp.eps);
  }
  __syncthreads();
  const float rs = *stat;
  if (tid < hp.d) xn[tid] = Elem<T>::to_f32(Elem<T>::from_f32(g * (x * rs)));
  __syncthreads();
}

// NP projections of 64 outputs each: out[p][o] = T-rounded sum_k xn[k] * W[(row0[p] + o) * ldw + k].
// 1024 threads: 16 consecutive lanes per output, KC 8-element chunks per lane (d = 128 KC); all loads up front.
template <typename T, int KC, int NP>
__device__ inline void head_proj(const HeadProjP& hp, const int (&row0)[NP], const float* xn, float (*out)[64]) {
  const int tid = threadIdx.x, o = tid >> 4, ks = tid & 15;
  constexpr int kper = KC * 8;
  Raw8<T> raw[NP][KC];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const T* wp = reinterpret_cast<const T*>(hp.W) + (long)(row0[q] + o) * hp.ldw + ks * kper;
#pragma unroll
    for (int c = 0; c < KC; ++c) raw[q][c].load(wp + c * 8);
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      float w[8];
      raw[q][c].unpack(w);
      const float4 x0 = *reinterpret_cast<const float4*>(xn + ks * kper + c * 8);
      const float4 x1 = *reinterpret_cast<const float4*>(xn + ks * kper + c * 8 + 4);
      acc += x0.x * w[0] + x0.y * w[1] + x0.z * w[2] + x0.w * w[3] + x1.x * w[4] + x1.y * w[5] + x1.z * w[6] + x1.w * w[7];
    }
    acc = group_sum<16>(acc);
    if (ks == 0) out[q][o] = Elem<T>::to_f32(Elem<T>::from_f32(acc));
  }
}

// cross-attention of one (b, h) with its own query projection; 16 waves, one key split (the default configuration
// of dec_cross_attn_kernel, same key interleave and merge order)
template <typename T, int KC, int U>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))   // <= 64 VGPRs: 2 workgroups per CU
void dec_cross_attn_q_kernel(CrossAttnP p, HeadProjP hp) {
  constexpr int NW = 16;
  __shared__ float sm[NW][66];
  __shared__ __attribute__((aligned(16))) float xn[1024];
  __shared__ float qs[1][64];
  __shared__ float stat;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int blk = blockIdx.x;
  int b, h;
  if (p.kv_B > 0) {
    const int q2 = blk >> 1;
    b = (blk & 1) * p.kv_B + q2 / p.H;
    h = q2 % p.H;
  } else {
    b = blk / p.H;
    h = blk % p.H;
  }
  const int c8 = (lane & 7) * 8, g = lane >> 3;
  norm_row_to_lds<T>(hp, b, xn, &stat);
  const int row0[1] = {h * 64};
  head_proj<T, KC, 1>(hp, row0, xn, qs);
  __syncthreads();
  float q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = qs[0][c8 + i];
  const int kvb = p.kv_B > 0 ? b % p.kv_B : b;
  const T* kb = reinterpret_cast<const T*>(p.k) + ((long)kvb * p.H + h) * p.L * 64;
  const T* vb = reinterpret_cast<const T*>(p.v) + ((long)kvb * p.H + h) * p.L * 64;
  Partial st;
  partial_init(st);
  attend_keys<T, U>(st, q, kb, vb, wid * 8 + g, p.L, 8 * NW, nullptr, 0, nullptr, 0, 1.0f);
  partial_merge_groups<T>(st);
  float m, l, a;
  block_merge<T, NW>(st, sm, m, l, a);
  if (threadIdx.x < 64)
    reinterpret_cast<T*>(p.out)[(long)b * p.ldo + h * 64 + threadIdx.x] = Elem<T>::from_f32(l > 0.f ? a / l : 0.f);
}

// self-attention of one (b, h) with its own q / k / v projections: appends the new key / value row to the caches and
// attends over keys 0 .. pos-1 from the cache plus the new key straight from LDS (merged last)
template <typename T, int KC>
__global__ __launch_bounds__(1024) void dec_self_attn_qkv_kernel(SelfAttnP p, HeadProjP hp, int inner) {
  constexpr int NW = 16;
  __shared__ float sm[NW][66];
  __shared__ __attribute__((aligned(16))) float xn[1024];
  __shared__ float qkv[3][64];
  __shared__ float stat;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int pos = *p.pos;
  const int c8 = (lane & 7) * 8, g = lane >> 3;
  norm_row_to_lds<T>(hp, b, xn, &stat);
  if (sizeof(T) == 2) {
    const int row0[3] = {h * 64, inner + h * 64, 2 * inner + h * 64};
    head_proj<T, KC, 3>(hp, row0, xn, qkv);
  } else {   // fp32 storage: one projection at a time (register budget of a 1024-thread workgroup)
#pragma unroll
    for (int q3 = 0; q3 < 3; ++q3) {
      const int row0[1] = {q3 * inner + h * 64};
      head_proj<T, KC, 1>(hp, row0, xn, qkv + q3);
    }
  }
  __syncthreads();
  T* kcache = reinterpret_cast<T*>(const_cast<void*>(p.kc)) + ((long)b * p.H + h) * p.tgt_len * 64;
  T* vcache = reinterpret_cast<T*>(const_cast<void*>(p.vc)) + ((long)b * p.H + h) * p.tgt_len * 64;
  if (threadIdx.x >= 64 && threadIdx.x < 128) kcache[(long)pos * 64 + (threadIdx.x - 64)] = Elem<T>::from_f32(qkv[1][threadIdx.x - 64]);
  if (threadIdx.x >= 128 && threadIdx.x < 192) vcache[(long)pos * 64 + (threadIdx.x - 128)] = Elem<T>::from_f32(qkv[2][threadIdx.x - 128]);
  float q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = qkv[0][c8 + i];
  const float* bias_row = p.bias + (long)h * p.tgt_len;
  const uint8_t* mask_row = p.prompt_mask ? p.prompt_mask + (long)b * p.P : nullptr;
  Partial st;
  partial_init(st);
  attend_keys<T, 2>(st, q, kcache, vcache, wid * 8 + g, pos, 8 * NW, bias_row, pos, mask_row, p.P, 1.0f);
  partial_merge_groups<T>(st);
  float m, l, a;
  block_merge<T, NW>(st, sm, m, l, a);
  if (threadIdx.x < 64) {
    const int d = threadIdx.x;
    float sn = wave_sum(qkv[0][d] * qkv[1][d]) + bias_row[0];
    if (mask_row && pos < p.P && mask_row[pos < p.P ? pos : 0] == 0) sn = -INFINITY;
    const float mn = fmaxf(m, sn);
    const float fa = fexp<T>(m - mn), fb = fexp<T>(sn - mn);
    l = l * fa + fb;
    a = a * fa + qkv[2][d] * fb;
    reinterpret_cast<T*>(p.out)[(long)b * p.ldo + h * 64 + d] = Elem<T>::from_f32(l > 0.f ? a / l : 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(64) void dec_cross_merge_kernel(CrossAttnP p) {
  const int pair = blockIdx.x, d = threadIdx.x;
  const int b = pair / p.H, h = pair % p.H;
  const float* pp = p.part + (long)pair * p.splits * 66;
  float m = pp[0], l = pp[1], a = pp[2 + d];
  for (int s = 1; s < p.splits; ++s) {
    const float* ps = pp + s * 66;
    const float m2 = ps[0], l2 = ps[1], a2 = ps[2 + d];
    const float mn = fmaxf(m, m2);
    const float fa = fexp<T>(m - mn), fb = fexp<T>(m2 - mn);
    l = l * fa + l2 * fb;
    a = a * fa + a2 * fb;
    m = mn;
  }
  reinterpret_cast<T*>(p.out)[(long)b * p.ldo + h * 64 + d] = Elem<T>::from_f32(l > 0.f ? a / l : 0.f);
}

}  // namespace dec
}  // namespace mh
