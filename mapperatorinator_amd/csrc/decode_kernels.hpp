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
template <typename T> __device__ inline void store8(T* p, const float (&o)[8]);
template <> __device__ inline void store8<bf16_t>(bf16_t* p, const float (&o)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(o[2 * i]) | ((uint32_t)f32_to_bf16(o[2 * i + 1]) << 16);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
template <> __device__ inline void store8<float>(float* p, const float (&o)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

template <typename T> __device__ inline void store4(T* p, float a, float b, float c, float d);
template <> __device__ inline void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(p) = make_uint2((uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16),
                                            (uint32_t)f32_to_bf16(c) | ((uint32_t)f32_to_bf16(d) << 16));
}
template <> __device__ inline void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// ---- skinny GEMM ----------------------------------------------------------------------------
enum { PRO_PLAIN = 0, PRO_RMSNORM = 1 };
enum { SK_STORE = 0, SK_QKV = 1, SK_GEGLU = 2, SK_RESID = 3, SK_LOGITS = 4 };

struct SkinnyP {
  const void* A; int lda;      // PRO_PLAIN: T [B, lda];  PRO_RMSNORM: fp32 residual stream [B, lda]
  const float* ln_w; float eps;
  const void* W; int ldw;      // [N, ldw] element type T
  int B, N, K;
  void* out; int ldo;          // STORE: T [B, ldo]; GEGLU: T [B, ldo] (N/2 cols); LOGITS: f32 [B, ldo]
  float* h; int ldh;           // RESID: h[b][n] += acc
  void* kc; void* vc;          // QKV: this layer's self-attention caches [B][H][tgt_len][64]
  int H, tgt_len, inner;
  const int* pos;
};

template <typename T> struct VecOps;
template <> struct VecOps<bf16_t> {
  static constexpr int NMMA = 1;  // MFMAs per 16-byte fragment vector
  __device__ static inline f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
    union U { uint4 u; bf16x8_t f; };
    U ua, ub;
    ua.u = a; ub.u = b;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.f, ub.f, c, 0, 0, 0);
  }
  // normalised fp32 values -> one 16-byte fragment (8 k's)
  __device__ static inline uint4 norm_frag(const float* hrow, const float* w, float rs) {
    float x[8], g[8], y[8];
    load8<float>(hrow, x);
    load8<float>(w, g);
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = g[i] * (x[i] * rs);
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (uint32_t)f32_to_bf16(y[2 * i]) | ((uint32_t)f32_to_bf16(y[2 * i + 1]) << 16);
    return make_uint4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct VecOps<float> {
  static constexpr int NMMA = 4;
  __device__ static inline f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
    // k permutation: element i of every lane's 4-float vector forms one 16x16x4 product
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
  __device__ static inline uint4 norm_frag(const float* hrow, const float* w, float rs) {
    const float4 x = *reinterpret_cast<const float4*>(hrow);
    const float4 g = *reinterpret_cast<const float4*>(w);
    return make_uint4(__float_as_uint(g.x * (x.x * rs)), __float_as_uint(g.y * (x.y * rs)),
                      __float_as_uint(g.z * (x.z * rs)), __float_as_uint(g.w * (x.w * rs)));
  }
};


// Dynamic LDS layout of skinny_gemm_kernel:
//   [ normalised A tile : MF*16 rows x (K*sizeof(T) + 16) bytes ]   (PRO_RMSNORM only)
//   [ cross-wave reduction buffer : 4 x NS x MF x 64 x f32x4 ]      (aliases the A tile after the K loop)
template <typename T, int MF, int NS, int PRO>
inline size_t skinny_smem_bytes(int K) {
  const size_t a = PRO == PRO_RMSNORM ? (size_t)MF * 16 * ((size_t)K * sizeof(T) + 16) : 0;
  const size_t r = (size_t)4 * NS * MF * 64 * 16;
  return a > r ? a : r;
}

template <typename T, int MF, int NS, int PRO, int EPI>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyP p) {
  constexpr int VEC = Elem<T>::kVec;   // elements per 16-byte vector (per lane per k-block)
  constexpr int KB = 4 * VEC;          // k elements per k-block (4 lane groups x 16 B)
  constexpr int CH = 8;                // k-blocks per wave whose loads are all issued before the first MFMA
  extern __shared__ __attribute__((aligned(16))) char sk_smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int strip0 = blockIdx.x * NS;
  const int a_stride = p.K * (int)sizeof(T) + 16;   // bytes per LDS A row

  if (PRO == PRO_RMSNORM) {
    // 8 threads per row, 32 rows per pass; a thread keeps its slice of the fp32 residual row in registers
    // (<= 32 float4 = d_model 1024), so h is read from L2 exactly once per workgroup, all loads in flight.
    constexpr int NV = 32;
    const int sub = tid & 7;
    const int nvec = p.K / 32;           // float4 per thread (K multiple of 32)
#pragma unroll 1
    for (int pass = 0; pass < (MF * 16 + 31) / 32; ++pass) {
      const int row = pass * 32 + (tid >> 3);
      const bool live = row < p.B && row < MF * 16;
      const float* hr = reinterpret_cast<const float*>(p.A) + (long)(live ? row : 0) * p.lda + sub * 4;
      // UNCONDITIONAL loads from clamped addresses, masked afterwards: a predicated load inside an unrolled
      // loop makes hipcc branch around every load and wait for each one (32 serial L2 round trips)
      float4 hv[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) hv[i] = *reinterpret_cast<const float4*>(hr + (i < nvec ? i : nvec - 1) * 32);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float q2 = hv[i].x * hv[i].x + hv[i].y * hv[i].y + hv[i].z * hv[i].z + hv[i].w * hv[i].w;
        ss += (i < nvec) ? q2 : 0.f;
      }
      ss = group_sum<8>(ss);
      const float rs = live ? rsqrtf(ss / (float)p.K + p.eps) : 0.f;
      if (row < MF * 16) {
        T* ar = reinterpret_cast<T*>(sk_smem + (long)row * a_stride);
        float4 gv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) gv[i] = *reinterpret_cast<const float4*>(p.ln_w + (i < nvec ? i : nvec - 1) * 32 + sub * 4);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          if (i < nvec) {   // LDS stores only (no loads under the branch)
            const int k = i * 32 + sub * 4;
            store4<T>(ar + k, gv[i].x * (hv[i].x * rs), gv[i].y * (hv[i].y * rs), gv[i].z * (hv[i].z * rs),
                      gv[i].w * (hv[i].w * rs));
          }
        }
      }
    }
    __syncthreads();
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

  const int nkb = p.K / KB;
  // wave w owns k-blocks w, w+4, w+8, ...; CH of them per pass with every load issued up front
  for (int kb0 = wid; kb0 < nkb; kb0 += 4 * CH) {
    uint4 wv[CH][NS];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kb = kb0 + 4 * c;
      const int kel = (kb < nkb ? kb : kb0) * KB;
#pragma unroll
      for (int s = 0; s < NS; ++s) wv[c][s] = *reinterpret_cast<const uint4*>(Wp[s] + kel);
    }
    uint4 av[CH][MF];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kb = kb0 + 4 * c;
      const int kel = (kb < nkb ? kb : kb0) * KB + lg * VEC;
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const int row = f * 16 + l15;
        if (PRO == PRO_PLAIN) {
          const int rc = row < p.B ? row : p.B - 1;   // clamped address, masked value (no predicated load)
          uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.A) + (long)rc * p.lda + kel);
          const uint32_t keep = row < p.B ? 0xffffffffu : 0u;
          av[c][f] = make_uint4(t.x & keep, t.y & keep, t.z & keep, t.w & keep);
        } else {
          av[c][f] = *reinterpret_cast<const uint4*>(sk_smem + (long)row * a_stride + (long)kel * sizeof(T));
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (kb0 + 4 * c < nkb) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int f = 0; f < MF; ++f) acc[s][f] = VecOps<T>::mma(av[c][f], wv[c][s], acc[s][f]);
      }
    }
  }

  __syncthreads();   // every wave is done with the A tile: the reduction buffer may alias it
  f32x4_t* red = reinterpret_cast<f32x4_t*>(sk_smem);   // [4][NS][MF][64]
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int f = 0; f < MF; ++f) red[((wid * NS + s) * MF + f) * 64 + lane] = acc[s][f];
  __syncthreads();

  const int pos = (EPI == SK_QKV) ? *p.pos : 0;
  constexpr int NSE = (EPI == SK_GEGLU) ? 1 : NS;
  for (int idx = tid; idx < NSE * MF * 64; idx += 256) {
    const int ln = idx & 63, f = (idx >> 6) % MF, s = (idx >> 6) / MF;
    f32x4_t v = red[((0 * NS + s) * MF + f) * 64 + ln];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4_t t = red[((w * NS + s) * MF + f) * 64 + ln];
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    f32x4_t u = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (EPI == SK_GEGLU) {
      u = red[((0 * NS + 1) * MF + f) * 64 + ln];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f32x4_t t = red[((w * NS + 1) * MF + f) * 64 + ln];
        u[0] += t[0]; u[1] += t[1]; u[2] += t[2]; u[3] += t[3];
      }
    }
    const int c15 = ln & 15;
    float oldh[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == SK_RESID) {   // unconditional loads of the residual stream from clamped addresses
      const int colc = (strip0 + s) * 16 + c15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = f * 16 + (ln >> 4) * 4 + r;
        oldh[r] = p.h[(long)(row < p.B ? row : p.B - 1) * p.ldh + (colc < p.N ? colc : p.N - 1)];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = f * 16 + (ln >> 4) * 4 + r;
      if (row >= p.B) continue;
      if (EPI == SK_GEGLU) {
        const int col = (strip0 / 2) * 16 + c15;
        if (col < p.N / 2)
          reinterpret_cast<T*>(p.out)[(long)row * p.ldo + col] = Elem<T>::from_f32(gelu_tanh(v[r]) * u[r]);
        continue;
      }
      const int col = (strip0 + s) * 16 + c15;
      if (col >= p.N) continue;
      if (EPI == SK_STORE) {
        reinterpret_cast<T*>(p.out)[(long)row * p.ldo + col] = Elem<T>::from_f32(v[r]);
      } else if (EPI == SK_LOGITS) {
        reinterpret_cast<float*>(p.out)[(long)row * p.ldo + col] = v[r];
      } else if (EPI == SK_RESID) {
        p.h[(long)row * p.ldh + col] = oldh[r] + v[r];
      } else if (EPI == SK_QKV) {
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
      load8<T>(kbase + (long)jc * 64 + c8, kv[u]);
    }
    float vv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = j + u * jstride;
      const int jc = jj < jend ? jj : j;
      load8<T>(vbase + (long)jc * 64 + c8, vv[u]);
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
template <typename T>
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
    for (int w = 1; w < 4; ++w) {
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
  int B, H, L, splits;
};

// one workgroup (4 waves) per (b, h, split); the waves interleave 8-key rows of the split's key range
template <typename T>
__global__ __launch_bounds__(256) void dec_cross_attn_kernel(CrossAttnP p) {
  __shared__ float sm[4][66];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int split = blockIdx.x % p.splits;
  const int pair = blockIdx.x / p.splits;
  const int b = pair / p.H, h = pair % p.H;
  const int c8 = (lane & 7) * 8, g = lane >> 3;
  const int per = (p.L + p.splits - 1) / p.splits;
  const int k_lo = split * per;
  int k_hi = k_lo + per;
  k_hi = k_hi < p.L ? k_hi : p.L;
  float q[8];
  load8<T>(reinterpret_cast<const T*>(p.q) + (long)b * p.ldq + h * 64 + c8, q);
  const T* kb = reinterpret_cast<const T*>(p.k) + ((long)b * p.H + h) * p.L * 64;
  const T* vb = reinterpret_cast<const T*>(p.v) + ((long)b * p.H + h) * p.L * 64;
  Partial st;
  partial_init(st);
  attend_keys<T, 4>(st, q, kb, vb, k_lo + wid * 8 + g, k_hi, 32, nullptr, 0, nullptr, 0, 1.0f);
  partial_merge_groups<T>(st);
  float m, l, a;
  block_merge<T>(st, sm, m, l, a);
  if (threadIdx.x < 64) {
    const int d = threadIdx.x;
    if (p.splits == 1) {
      reinterpret_cast<T*>(p.out)[(long)b * p.ldo + h * 64 + d] = Elem<T>::from_f32(l > 0.f ? a / l : 0.f);
    } else {
      float* pp = p.part + ((long)pair * p.splits + split) * 66;
      if (d == 0) { pp[0] = m; pp[1] = l; }
      pp[2 + d] = a;
    }
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
