// Dense layer C = A[M,K] * W[N,K]^T on the gfx950 matrix cores with fused epilogues.
//
// Structure: LDS-tiled, register-prefetched, double-buffered LDS, one barrier per K step.
//   block = 256 threads = 4 waves as 2(M) x 2(N); tile BMxBN in {128x128, 64x64};
//   K step = 128 bytes per row (64 bf16 / 32 f32) staged with 16-byte loads; LDS rows padded by 16 B.
//   MFMA atom: v_mfma_f32_16x16x32_bf16 (bf16 storage) or the exact v_mfma_f32_16x16x4_f32 (f32).
// Both operands are K-contiguous ("B^T input"), which is the nn.Linear weight layout -- no transposes.
#include <type_traits>

#include "common.hpp"

namespace mh {

namespace {

// K-step bytes per tile row: 128 for the big tiles; 512 for the 32x32 tile, whose problems (DiT, M = 256) are
// bound by the global-load latency of a K step (~1 us per step whatever the tile: measured), so it takes 4x
// fewer, 4x fatter steps with 8 loads in flight per thread; 1024 for the 16x16 split-K tile (77.8 -> 74.4 ms per
// 100 DiT-S steps; its two stages still fit four workgroups per CU).
template <int BM> struct RowBytes { static constexpr int v = (BM == 16) ? 1024 : (BM == 32) ? 512 : 128; };

struct GemmP {
  const char* A; long lda_b;  // leading dimension in BYTES
  const char* W; long ldw_b;
  void* C; int ldc;
  int M, N, K;
  const float* bias;
  const float* gate; int gate_ld; int rows_per_batch;
  int kv_B, kv_H, kv_L;
  void* C2; int n_split; int kv_Lpad;
  void* C3; void* C4; int cache_len;
  float* stats_out;                                   // producer: [ceil(N/16)][M][2] row sums / sums of squares
  const float* ln_stats; int ln_strips;               // consumer: LayerNorm + modulate applied to the A operand (f32)
  const float* ln_shift; const float* ln_scale; int ln_ld; float ln_eps;
  bool ascending_k;
  int split3;   // fp32 only: W holds [hi bf16 x32 | lo bf16 x32] per 32-float K block, A is split on the way into LDS
  const uint8_t* a_scale; const uint8_t* w_scale; int ks_b;   // MH_MX8: E8M0 scales (lane-major groups, mx8.hip), bytes per row
  uint8_t* mxq; uint8_t* mxs; int mxs_b;                        // MX-fp8 image of the epilogue's result (template MXO): bytes [M][ldc], scales [M][mxs_b]
};

// `v` already contains the bias; `old` = previous C value (RESID / GATE_RESID), `g` = gate value, `v2` = paired
// linear value (GEGLU).  All global LOADS feeding this function are issued unconditionally by the caller
// (clamped addresses): predicated loads in an unrolled epilogue serialise into one round trip each.
template <typename T, int EPI>
__device__ inline void epilogue_store(const GemmP& p, int row, int col, float v, float v2, float old, float g) {
  if (row >= p.M) return;
  if (EPI == MH_EPI_GEGLU) {
    if (col >= p.N / 2) return;   // col is the output column in [0, N/2)
    reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = Elem<T>::from_f32(gelu_tanh(v) * v2);
    return;
  }
  if (col >= p.N) return;
  if (EPI == MH_EPI_STORE) {
    reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = Elem<T>::from_f32(v);
  } else if (EPI == MH_EPI_STORE_F32) {
    reinterpret_cast<float*>(p.C)[(long)row * p.ldc + col] = v;
  } else if (EPI == MH_EPI_RESID) {
    reinterpret_cast<float*>(p.C)[(long)row * p.ldc + col] = old + v;
  } else if (EPI == MH_EPI_BIAS_GELU) {
    reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = Elem<T>::from_f32(gelu_tanh(v));
  } else if (EPI == MH_EPI_BIAS_GELU_ERF) {
    // exact GELU (nn.functional.gelu default) + optional position row (passed in `g`)
    reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = Elem<T>::from_f32(0.5f * v * (1.0f + erff(v * 0.70710678118654752f)) + g);
  } else if (EPI == MH_EPI_GATE_RESID) {
    reinterpret_cast<float*>(p.C)[(long)row * p.ldc + col] = old + g * v;
  } else if (EPI == MH_EPI_KV_SCATTER) {
    // col = ((layer*2 + kv) * H + h) * 64 + dd ; row = b * L + key
    const int dd = col & 63;
    const int h = (col >> 6) % p.kv_H;
    const int lk = (col >> 6) / p.kv_H;  // layer*2 + kv
    const int b = row / p.kv_L, key = row - b * p.kv_L;
    long dst = ((((long)lk * p.kv_B + b) * p.kv_H + h) * p.kv_L + key) * 64 + dd;
    reinterpret_cast<T*>(p.C)[dst] = Elem<T>::from_f32(v);
  } else if (EPI == MH_EPI_QKV_VT) {
    if (col < p.n_split) {
      reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = Elem<T>::from_f32(v);
    } else {
      const int c2 = col - p.n_split;  // h*64 + dd
      const int b = row / p.kv_L, key = row - b * p.kv_L;
      long dst = ((long)b * p.kv_H * 64 + c2) * p.kv_Lpad + key;
      reinterpret_cast<T*>(p.C2)[dst] = Elem<T>::from_f32(v);
    }
  } else if (EPI == MH_EPI_QKV_CACHE) {
    if (col < p.n_split) {
      reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = Elem<T>::from_f32(v);
    } else {
      const int inner = p.kv_H * 64;
      const int c2 = col - p.n_split;
      const int kv = c2 / inner, c = c2 - kv * inner;   // 0 = k, 1 = v ; c = h*64 + dd
      const int b = row / p.kv_L, i = row - b * p.kv_L;
      const T tv = Elem<T>::from_f32(v);
      const long dst = (((long)b * p.kv_H + (c >> 6)) * p.cache_len + i) * 64 + (c & 63);
      if (kv == 0) {
        reinterpret_cast<T*>(p.C2)[dst] = tv;
      } else {
        reinterpret_cast<T*>(p.C3)[dst] = tv;
        reinterpret_cast<T*>(p.C4)[((long)b * p.kv_H * 64 + c) * p.kv_Lpad + i] = tv;
      }
    }
  }
}

// S3 ("bf16 x 3"): an fp32 GEMM on the bf16 matrix cores.  a = a_hi + a_lo, w = w_hi + w_lo with hi = bf16(x),
// lo = bf16(x - hi); a w ~= a_hi w_hi + a_hi w_lo + a_lo w_hi (the dropped a_lo w_lo and the 16-bit significands leave a
// relative error of ~2^-16 per product, fp32 accumulation).  Three v_mfma_f32_16x16x32_bf16 replace eight exact
// v_mfma_f32_16x16x4_f32: 1/5 of the matrix-core time at the SAME operand bytes (weights are stored pre-split, 128 B per
// 32-float block either way; activations stay fp32 in HBM and are split while they are staged into LDS).
// GL ("global_load_lds"): bf16, 128 x 128 tile, no operand transform -- the encoder GEMMs.  Both operand tiles go from
// HBM / L2 straight into LDS (`global_load_lds_dwordx4`: no staging registers, no ds_write pass); an LDS-DMA writes
// wave-uniform base + lane * 16, so the stage is lane-linear (128-byte rows, no padding) and the bank-conflict-free
// layout comes from the SOURCE side: LDS chunk c of row r holds global chunk c ^ (r & 7) (still one whole line per row).
__device__ __attribute__((aligned(16))) uint4 g_zero16 = {0, 0, 0, 0};   // source of K-tail chunks

template <typename T, int BM, int BN, int EPI, bool S3 = false, bool GL = false>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmP p) {
  static_assert(!S3 || (std::is_same<T, float>::value && BM == 64 && BN == 64), "split-3 path: fp32, 64x64 tile");
  static_assert(!GL || (sizeof(T) == 2 && BM == 128 && BN == 128 && !S3), "LDS-DMA path: bf16, 128x128 tile");
  constexpr int VEC = Elem<T>::kVec;          // elements per 16 B
  constexpr int kRowBytes = RowBytes<BM>::v;
  constexpr int kRowStride = GL ? kRowBytes : kRowBytes + 16;  // LDS row stride in bytes (padded unless lane-linear)
  constexpr int CPR = kRowBytes / 16;         // 16-byte chunks per tile row
  constexpr int RPP = 256 / CPR;              // tile rows covered by one pass of the 256 threads
  constexpr int BK = kRowBytes / (int)sizeof(T);
  constexpr int KM = Atom<T>::KM;
  constexpr int KCH = Atom<T>::KCH;
  // BM = BN = 16: ONE 16x16 output tile per workgroup, its 4 waves split every K step four ways and the partial
  // accumulators are added through LDS in wave order (deterministic).  For the tiny fp32 GEMMs of the DiT (M = 256
  // rows, N = 384) this gives 4x the workgroups of the 32x32 tile and a quarter of the per-wave MFMA time.
  constexpr bool kSplitK = (BM == 16);
  static_assert(!kSplitK || BN == 16, "split-K tile is 16x16");
  constexpr int WM = kSplitK ? 16 : BM / 2, WN = kSplitK ? 16 : BN / 2;     // wave tile
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int A_CHUNKS = BM * CPR / 256;    // 16-byte chunks per thread for the A tile
  constexpr int B_CHUNKS = BN * CPR / 256;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kBufBytes = (BM + BN) * kRowStride;  // one (A tile, B tile) stage

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = kSplitK ? 0 : wid >> 1, wc = kSplitK ? 0 : wid & 1;

  // XCD-aware super-tile order.  Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with
  // its own 4 MB L2: the bijective remap below gives every XCD a CONTIGUOUS range of a work list, and the work list
  // walks the tile grid in groups of GM row panels: inside a group the row panel is the fastest index, then the column
  // tile.  The ~64 tiles an XCD runs at once are GM A panels x a few W panels (<= ~3 MB, L2-resident), every W panel is
  // fetched once per GROUP instead of once per A panel (the cross-K/V GEMM re-streamed its 28 MB of weights 313 times:
  // 9.1 GB of L2 misses for 90 MB of operands).  GM is sized so that GM panels of A fit in ~2.5 MB.
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int bm, bn;
  {
    const long panel_bytes = (long)BM * p.K * (long)sizeof(T);
    int GM = (int)((5L << 19) / (panel_bytes > 0 ? panel_bytes : 1));   // 2.5 MB of A panels
    GM = GM < 2 ? 2 : (GM > 16 ? 16 : GM);
    const int per_group = GM * nbn;
    const int grp = bid / per_group, rem = bid - grp * per_group;
    const int gm = (nbm - grp * GM) < GM ? (nbm - grp * GM) : GM;       // panels in this (possibly last, partial) group
    bn = rem / gm;
    bm = grp * GM + (rem - bn * gm);
  }
  const int m0 = bm * BM, n0 = bn * BN;

  const int cchunk = tid % CPR;   // 16-byte chunk within the K step
  const int crow = tid / CPR;     // 0..RPP-1

  // Register ring of D in-flight K tiles (tile t lives in slot t % D) in front of the double-buffered LDS stage.
  // Measured: D = 4 made the small (M = 256) fp32 GEMMs of the DiT SLOWER (140 -> 188 ms per 100 steps): those are
  // bound by the per-wave MFMA issue time of a K step (32 x 32-cycle f32 MFMAs), not by load latency; what helps
  // them is more workgroups (the 32x32 tile below).  D stays 1.
  constexpr int D = 1;   // (D = 3 on the 32x32 tile, K = 384 all in flight: 97.1 vs 95.1 ms per 100 DiT-S steps)
  uint4 ra[D][A_CHUNKS], rb[D][B_CHUNKS];
  const int nk = (p.K + BK - 1) / BK;

  // fused LayerNorm + adaLN modulate on the A operand (fp32 activations of the DiT): mean / rstd of this tile's rows
  // from the producer's per-strip sums, kept in LDS behind the two tile stages
  constexpr bool kLnCapable = std::is_same<T, float>::value && !S3;   // (the bf16 x 3 form takes its LayerNorm from the stand-alone pass: refused in gemm())
  const bool ln = kLnCapable && p.ln_stats != nullptr;   // block-uniform
  float* lnst = reinterpret_cast<float*>(smem + 2 * kBufBytes);
  uint4 rsc[D][kLnCapable ? A_CHUNKS : 1], rsh[D][kLnCapable ? A_CHUNKS : 1];
  auto load_tiles = [&](int kt, uint4 (&xa)[A_CHUNKS], uint4 (&xb)[B_CHUNKS], uint4* xsc, uint4* xsh) {
    const int k_el = kt * BK + cchunk * VEC;
    const bool kin = k_el < p.K;
    const long k_off = (long)(kin ? k_el : 0) * sizeof(T);   // clamped address, value masked below (no predicated load)
    const uint32_t keep = kin ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      int r = m0 + crow + RPP * i;
      r = r < p.M ? r : p.M - 1;
      const uint4 t = *reinterpret_cast<const uint4*>(p.A + (long)r * p.lda_b + k_off);
      xa[i] = make_uint4(t.x & keep, t.y & keep, t.z & keep, t.w & keep);
      if (kLnCapable && ln) {   // block-uniform branch; modulation vectors of this row's batch entry, same k range
        const long mo = (long)(r / p.rows_per_batch) * p.ln_ld + (kin ? k_el : 0);
        xsc[i] = *reinterpret_cast<const uint4*>(p.ln_scale + mo);
        xsh[i] = *reinterpret_cast<const uint4*>(p.ln_shift + mo);
      }
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      int r = n0 + crow + RPP * i;
      r = r < p.N ? r : p.N - 1;
      const uint4 t = *reinterpret_cast<const uint4*>(p.W + (long)r * p.ldw_b + k_off);
      xb[i] = make_uint4(t.x & keep, t.y & keep, t.z & keep, t.w & keep);
    }
  };
  auto store_tiles = [&](int buf, int kt, const uint4 (&xa)[A_CHUNKS], const uint4 (&xb)[B_CHUNKS], const uint4* xsc,
                         const uint4* xsh) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      uint4 v = xa[i];
      if (kLnCapable && ln) {
        const bool kin = kt * BK + cchunk * VEC < p.K;   // K-tail columns stay zero
        float mu = lnst[2 * (crow + RPP * i)], rs = lnst[2 * (crow + RPP * i) + 1];
        const float m = kin ? 1.f : 0.f;
        uint4 qsc = xsc[i], qsh = xsh[i];
        v.x = __float_as_uint(m * ((__uint_as_float(v.x) - mu) * rs * (1.f + __uint_as_float(qsc.x)) + __uint_as_float(qsh.x)));
        v.y = __float_as_uint(m * ((__uint_as_float(v.y) - mu) * rs * (1.f + __uint_as_float(qsc.y)) + __uint_as_float(qsh.y)));
        v.z = __float_as_uint(m * ((__uint_as_float(v.z) - mu) * rs * (1.f + __uint_as_float(qsc.z)) + __uint_as_float(qsh.z)));
        v.w = __float_as_uint(m * ((__uint_as_float(v.w) - mu) * rs * (1.f + __uint_as_float(qsc.w)) + __uint_as_float(qsh.w)));
      }
      if constexpr (S3) {
        // 4 fp32 -> 4 hi + 4 lo bf16: hi half of the 128-byte row block at cchunk * 8, lo half 64 bytes further
        const float f0 = __uint_as_float(v.x), f1 = __uint_as_float(v.y), f2 = __uint_as_float(v.z), f3 = __uint_as_float(v.w);
        const uint32_t h01 = pack_bf16x2(f0, f1), h23 = pack_bf16x2(f2, f3);
        const float r0 = f0 - __uint_as_float(h01 << 16), r1 = f1 - __uint_as_float(h01 & 0xffff0000u);
        const float r2 = f2 - __uint_as_float(h23 << 16), r3 = f3 - __uint_as_float(h23 & 0xffff0000u);
        char* rowp = smem + buf * kBufBytes + (crow + RPP * i) * kRowStride;
        *reinterpret_cast<uint2*>(rowp + cchunk * 8) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(rowp + 64 + cchunk * 8) = make_uint2(pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));
      } else {
        *reinterpret_cast<uint4*>(smem + buf * kBufBytes + (crow + RPP * i) * kRowStride + cchunk * 16) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i)
      *reinterpret_cast<uint4*>(smem + buf * kBufBytes + (BM + crow + RPP * i) * kRowStride + cchunk * 16) = xb[i];
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // GL: K tile kt of both operands -> LDS stage `buf`: 4 + 4 wave instructions of 8 rows x 128 bytes each
  auto glds_tiles = [&](int buf, int kt) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int r8 = lane >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (i * 4 + wid) * 8 + r8;                  // tile row of this lane
      const int k_el = kt * BK + (((lane & 7) ^ (r & 7)) * VEC);   // pre-swizzled source chunk (r & 7 == r8)
      const bool kin = k_el < p.K;
      int ra_ = m0 + r; ra_ = ra_ < p.M ? ra_ : p.M - 1;
      int rb_ = n0 + r; rb_ = rb_ < p.N ? rb_ : p.N - 1;
      const char* sa = kin ? p.A + (long)ra_ * p.lda_b + (long)k_el * sizeof(T) : reinterpret_cast<const char*>(&g_zero16);
      const char* sb = kin ? p.W + (long)rb_ * p.ldw_b + (long)k_el * sizeof(T) : reinterpret_cast<const char*>(&g_zero16);
      char* da = smem + buf * kBufBytes + (i * 4 + wid) * 1024;          // wave-uniform
      char* db = da + BM * kRowStride;
      __builtin_amdgcn_global_load_lds((gptr_t)sa, (lptr_t)da, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)db, 16, 0, 0);
    }
  };
  if constexpr (GL) {
    glds_tiles(0, 0);
  } else {
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nk) load_tiles(d, ra[d], rb[d], rsc[d], rsh[d]);
  }
  // (the LayerNorm statistics below are requested after the first operand tiles: one round trip covers both)
  // small tile: the epilogue's old C / gate values do not depend on the product either -- requested here, not after it
  constexpr bool kReadsC = (EPI == MH_EPI_RESID || EPI == MH_EPI_GATE_RESID);
  constexpr bool kPreEpi = kReadsC && BM <= 32;
  const int erow0 = m0 + wr * WM + (lane >> 4) * 4;
  const int ecol0 = n0 + wc * WN + (lane & 15);
  f32x4_t pre_old[kPreEpi ? MI : 1][kPreEpi ? NI : 1], pre_g[kPreEpi ? MI : 1][kPreEpi ? NI : 1];
  if constexpr (kPreEpi) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = erow0 + i * 16 + r;
        row = row < p.M ? row : p.M - 1;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          int c = ecol0 + j * 16;
          c = c < p.N ? c : p.N - 1;
          pre_old[i][j][r] = reinterpret_cast<const float*>(p.C)[(long)row * p.ldc + c];
          pre_g[i][j][r] = (EPI == MH_EPI_GATE_RESID) ? p.gate[(long)(row / p.rows_per_batch) * p.gate_ld + c] : 0.f;
        }
      }
  }
  if (ln) {
    // 256 threads = NG strip groups x BM rows; every thread has its (<= 8 per pass) strip loads in flight at once
    // (a serial loop over the strips costs one L2 round trip per strip: measured +6 ms per 100 DiT steps)
    constexpr int NG = 256 / BM;
    float* red = lnst + 2 * BM;   // [NG][BM][2]
    const int row_l = tid % BM, sg = tid / BM;
    const int row = (m0 + row_l) < p.M ? (m0 + row_l) : p.M - 1;
    float s1 = 0.f, s2 = 0.f;
    for (int base = 0; base < p.ln_strips; base += 8 * NG) {
      float2 v[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int st = base + sg + it * NG;
        v[it] = *reinterpret_cast<const float2*>(p.ln_stats + ((long)(st < p.ln_strips ? st : 0) * p.M + row) * 2);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const bool ok = base + sg + it * NG < p.ln_strips;
        s1 += ok ? v[it].x : 0.f;
        s2 += ok ? v[it].y : 0.f;
      }
    }
    red[(sg * BM + row_l) * 2] = s1;
    red[(sg * BM + row_l) * 2 + 1] = s2;
    __syncthreads();
    if (tid < BM) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) { t1 += red[(gq * BM + tid) * 2]; t2 += red[(gq * BM + tid) * 2 + 1]; }
      const float mu = t1 / (float)p.K;
      const float var = fmaxf(t2 / (float)p.K - mu * mu, 0.f);
      lnst[2 * tid] = mu;
      lnst[2 * tid + 1] = rsqrtf(var + p.ln_eps);
    }
    __syncthreads();
  }
  if constexpr (GL) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    store_tiles(0, 0, ra[0], rb[0], rsc[0], rsh[0]);
  }
  __syncthreads();

  const int frow = lane & 15, fk = (lane >> 4) * KCH;
  if constexpr (GL) {
    const int sw = frow & 7, lgc = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) glds_tiles(cur ^ 1, kt + 1);   // the stage read two iterations ago (barrier below) is free
      const char* a_row = smem + cur * kBufBytes + (wr * WM + frow) * kRowStride;
      const char* b_row = smem + cur * kBufBytes + (BM + wc * WN + frow) * kRowStride;
#pragma unroll
      for (int ks = 0; ks < BK / KM; ++ks) {
        const int coff = ((ks * 4 + lgc) ^ sw) * 16;
        typename Atom<T>::frag_t af[MI], bf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = Atom<T>::load(reinterpret_cast<const T*>(a_row + i * 16 * kRowStride + coff));
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[j] = Atom<T>::load(reinterpret_cast<const T*>(b_row + j * 16 * kRowStride + coff));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = Atom<T>::mma(af[i], bf[j], acc[i][j]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of the next stage has landed
      __syncthreads();
    }
  } else
  for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int kt = kt0 + d;
      if (kt >= nk) break;
      const int cur = kt & 1;
      if (kt + D < nk) load_tiles(kt + D, ra[d], rb[d], rsc[d], rsh[d]);   // slot d was drained into LDS one iteration ago
      const char* a_base = smem + cur * kBufBytes + (wr * WM + frow) * kRowStride + fk * (int)sizeof(T);
      const char* b_base = smem + cur * kBufBytes + (BM + wc * WN + frow) * kRowStride + fk * (int)sizeof(T);
      constexpr int kKs = BK / KM, kKsW = kSplitK ? kKs / 4 : kKs;
      static_assert(!kSplitK || kKs % 4 == 0, "K step not divisible among the 4 waves");
      const int ks0 = kSplitK ? wid * kKsW : 0;
      if constexpr (S3) {
        // one 32-deep bf16 k-step per 128-byte row block: fragments = 16 bytes at (lane >> 4) * 16 of the hi / lo half
        const char* a3 = smem + cur * kBufBytes + (wr * WM + frow) * kRowStride + (lane >> 4) * 16;
        const char* b3 = smem + cur * kBufBytes + (BM + wc * WN + frow) * kRowStride + (lane >> 4) * 16;
        bf16x8_t ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          ah[i] = *reinterpret_cast<const bf16x8_t*>(a3 + i * 16 * kRowStride);
          al[i] = *reinterpret_cast<const bf16x8_t*>(a3 + i * 16 * kRowStride + 64);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          bh[j] = *reinterpret_cast<const bf16x8_t*>(b3 + j * 16 * kRowStride);
          bl[j] = *reinterpret_cast<const bf16x8_t*>(b3 + j * 16 * kRowStride + 64);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            // small terms first, the dominant product last
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      } else if constexpr (std::is_same<T, float>::value && !GL) {
        // fp32 (round 5): k slot lg of MFMA step (g, e) <-> k index g*16 + lg*4 + e of the K step, for BOTH operands: the same products
        // in another order, and the fragments of four MFMA steps are ONE ds_read_b128 each instead of four ds_read_b32 (2 reads per
        // 16x16x4 MFMA were the inner loop's issue bound).  Every fp32 tile shape enumerates k this way, so `ascending_k` keeps
        // its meaning: one summation order whatever the row count.
        static_assert(kKsW % 4 == 0, "groups of four k steps");
        const char* a4 = smem + cur * kBufBytes + (wr * WM + frow) * kRowStride + (lane >> 4) * 16;
        const char* b4 = smem + cur * kBufBytes + (BM + wc * WN + frow) * kRowStride + (lane >> 4) * 16;
#pragma unroll
        for (int gq = 0; gq < kKsW / 4; ++gq) {
          const int g = ks0 / 4 + gq;
          float4 af[MI], bf[NI];
#pragma unroll
          for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(a4 + i * 16 * kRowStride + g * 64);
#pragma unroll
          for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const float4*>(b4 + j * 16 * kRowStride + g * 64);
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
            }
        }
      } else
#pragma unroll
      for (int kq = 0; kq < kKsW; ++kq) {
        const int ks = ks0 + kq;
        typename Atom<T>::frag_t af[MI], bf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
          af[i] = Atom<T>::load(reinterpret_cast<const T*>(a_base + i * 16 * kRowStride + ks * KM * (int)sizeof(T)));
#pragma unroll
        for (int j = 0; j < NI; ++j)
          bf[j] = Atom<T>::load(reinterpret_cast<const T*>(b_base + j * 16 * kRowStride + ks * KM * (int)sizeof(T)));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = Atom<T>::mma(af[i], bf[j], acc[i][j]);
      }
      if (kt + 1 < nk) store_tiles(cur ^ 1, kt + 1, ra[(d + 1) % D], rb[(d + 1) % D], rsc[(d + 1) % D], rsh[(d + 1) % D]);
      __syncthreads();
    }
  }

  if constexpr (kSplitK) {   // (the K loop ended with a barrier: the tile stages are free)
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);
    red[wid * 64 + lane] = acc[0][0];
    __syncthreads();
    if (wid != 0) return;
    f32x4_t v = red[lane];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      const f32x4_t t = red[w2 * 64 + lane];
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    acc[0][0] = v;
  }
  // epilogue: acc[i][j][r] = C[m0 + wr*WM + i*16 + (lane>>4)*4 + r][n0 + wc*WN + j*16 + (lane&15)]
  constexpr bool kPos = (EPI == MH_EPI_BIAS_GELU_ERF);
  float bias_v[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) bias_v[j] = 0.f;
  if (EPI != MH_EPI_GEGLU && p.bias) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int c = ecol0 + j * 16;
      bias_v[j] = p.bias[c < p.N ? c : p.N - 1];
    }
  }
  // The bias goes into the accumulators HERE and the result is pinned: left inside the guarded per-element stores below,
  // hipcc sinks the add -- and with it the wait for the bias load -- behind the stores, and since the load counter also
  // counts stores every element then waits for the store before it (one write round trip per element: 64 per lane on
  // the 128 x 128 tile).  The same holds for the old C / gate values of a row block.
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      acc[i][j][0] += bias_v[j]; acc[i][j][1] += bias_v[j]; acc[i][j][2] += bias_v[j]; acc[i][j][3] += bias_v[j];
      asm volatile("" : "+v"(acc[i][j]));
    }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    f32x4_t oldv[NI], gv[NI];
    if (kPos) {
#pragma unroll
      for (int j = 0; j < NI; ++j) gv[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (p.gate) {   // position table row = row % rows_per_batch (wave-uniform branch, clamped unconditional loads)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = erow0 + i * 16 + r;
          row = row < p.M ? row : p.M - 1;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            int c = ecol0 + j * 16;
            c = c < p.N ? c : p.N - 1;
            gv[j][r] = p.gate[(long)(row % p.rows_per_batch) * p.gate_ld + c];
          }
        }
      }
    }
    if constexpr (kPreEpi) {
#pragma unroll
      for (int j = 0; j < NI; ++j) { oldv[j] = pre_old[i][j]; gv[j] = pre_g[i][j]; }
    } else if (kReadsC) {   // unconditional loads from clamped addresses, all in flight before the first store
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = erow0 + i * 16 + r;
        row = row < p.M ? row : p.M - 1;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          int c = ecol0 + j * 16;
          c = c < p.N ? c : p.N - 1;
          oldv[j][r] = reinterpret_cast<const float*>(p.C)[(long)row * p.ldc + c];
          gv[j][r] = (EPI == MH_EPI_GATE_RESID) ? p.gate[(long)(row / p.rows_per_batch) * p.gate_ld + c] : 0.f;
        }
      }
    }
    if constexpr (kReadsC) {
#pragma unroll
      for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(oldv[j]), "+v"(gv[j]));   // loaded before the first store of the block
    } else if constexpr (kPos) {
#pragma unroll
      for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(gv[j]));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = erow0 + i * 16 + r;
      if (EPI == MH_EPI_GEGLU) {
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
          // 16-row weight blocks alternate wi_0 / wi_1: fragment j is the gate, j+1 the linear half
          const int ocol = (n0 + wc * WN) / 2 + (j / 2) * 16 + (lane & 15);
          epilogue_store<T, EPI>(p, row, ocol, acc[i][j][r], acc[i][j + 1][r], 0.f, 0.f);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NI; ++j)
          epilogue_store<T, EPI>(p, row, ecol0 + j * 16, acc[i][j][r], 0.f, kReadsC ? oldv[j][r] : 0.f,
                                 (kReadsC || kPos) ? gv[j][r] : 0.f);
        if constexpr (EPI == MH_EPI_STORE_F32 || EPI == MH_EPI_GATE_RESID) {
          if (p.stats_out) {   // block-uniform: row sums of the values just stored, one (sum, sum of squares) per 16 columns
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              const float lin = acc[i][j][r];
              float o = (EPI == MH_EPI_GATE_RESID) ? oldv[j][r] + gv[j][r] * lin : lin;
              o = (row < p.M && ecol0 + j * 16 < p.N) ? o : 0.f;
              const float s1 = group_sum<16>(o), s2 = group_sum<16>(o * o);
              if ((lane & 15) == 0 && row < p.M) {
                const int strip = (n0 + wc * WN + j * 16) >> 4;
                *reinterpret_cast<float2*>(p.stats_out + ((long)strip * p.M + row) * 2) = make_float2(s1, s2);
              }
            }
          }
        }
      }
    }
  }
}

// ---- epilogue of the three-stage kernels (transposed accumulators: a lane owns 4 CONSECUTIVE columns of one row) ---------
// acc[j][i][r] = C[row m0 + wr*16*MI + i*16 + (lane & 15)][col n0 + wc*64 + j*16 + (lane >> 4)*4 + r].  Shared by the bf16 form
// (gemm_glds3_kernel) and the MX-fp8 form (gemm_mx8_kernel): the C / D layout of the 16x16 MFMAs does not depend on the operand type.
template <int EPI, int MI, bool MXO = false>
__device__ __forceinline__ void g3_epilogue(const GemmP& p, f32x4_t (&acc)[4][MI], int m0, int n0, int wr, int wc, int lane) {
  static_assert(!MXO || EPI == MH_EPI_GEGLU || EPI == MH_EPI_BIAS_GELU, "MX-fp8 output: the epilogues whose result is the next GEMM's A operand");
  using T = bf16_t;
  constexpr int WM = 16 * MI, WN = 64, NI = 4;
  const int lgc = lane >> 4;
  // ---- epilogue: 4 consecutive columns per lane; every load it needs is in flight before the first store of its row
  // block (a load consumed behind a store makes hipcc wait for that store: one write round trip per element) ----------
  constexpr bool kReadsC = (EPI == MH_EPI_RESID || EPI == MH_EPI_GATE_RESID);
  constexpr bool kPos = (EPI == MH_EPI_BIAS_GELU_ERF);
  const int l15 = lane & 15;
  const int ecol_base = n0 + wc * WN + lgc * 4;     // + j * 16
  const int erow_base = m0 + wr * WM + l15;         // + i * 16
  auto f4 = [](const float* q) { return *reinterpret_cast<const float4*>(q); };
  auto st_bf16x4 = [](void* base, long idx, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(reinterpret_cast<T*>(base) + idx) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  };
  auto st_f32x4 = [](void* base, long idx, const float (&v)[4]) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = make_float4(v[0], v[1], v[2], v[3]);
  };
  float4 bias4[NI];
#pragma unroll
  for (int jj = 0; jj < NI; ++jj) {
    int c = ecol_base + jj * 16;
    c = c < p.N ? c : p.N - 4;
    bias4[jj] = (EPI != MH_EPI_GEGLU && p.bias) ? f4(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // (the bias is added per row block below, in front of that block's pinned values: hipcc otherwise sinks the add -- and the wait
  // for the bias load -- behind the stores, and every store then waits for the one before it.  Adding it to ALL accumulators up
  // front pinned 16 MI of them in VGPRs at once: with AGPR accumulators and a 128 x 64 wave tile that spilled.)
#pragma unroll
  for (int jj = 0; jj < NI; ++jj)   // the bias loads are waited for HERE
    asm volatile("" : "+v"(bias4[jj].x), "+v"(bias4[jj].y), "+v"(bias4[jj].z), "+v"(bias4[jj].w));
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = erow_base + i * 16;
    const bool rok = row < p.M;
    const int rowc = rok ? row : p.M - 1;
    float4 old4[NI], g4[NI];
    if constexpr (kReadsC || kPos) {
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) {
        int c = ecol_base + jj * 16;
        c = c < p.N ? c : p.N - 4;
        old4[jj] = kReadsC ? f4(reinterpret_cast<const float*>(p.C) + (long)rowc * p.ldc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == MH_EPI_GATE_RESID) g4[jj] = f4(p.gate + (long)(rowc / p.rows_per_batch) * p.gate_ld + c);
        else if (kPos && p.gate) g4[jj] = f4(p.gate + (long)(rowc % p.rows_per_batch) * p.gate_ld + c);
        else g4[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // all values of this row block first (pinned: nothing that depends on a load may sink behind a store), then the stores
    f32x4_t vv[NI];
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
      f32x4_t v = acc[jj][i];
      v[0] += bias4[jj].x; v[1] += bias4[jj].y; v[2] += bias4[jj].z; v[3] += bias4[jj].w;
      if constexpr (EPI == MH_EPI_GEGLU) {
        if (!(jj & 1)) {                             // 16-row weight blocks alternate wi_0 / wi_1: jj the gate, jj + 1 the linear half
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast(v[r]) * acc[jj + 1][i][r];   // (bf16 / MX outputs only here: the one-exponential form is ~1e-6 relative, 4 000 x below the output rounding; tanhf's ~45 instructions and two branches per value made this epilogue 5 us of VALU per 256 x 256 tile.  The fp32 kernels keep gelu_tanh: their results decide bit-exact greedy ids)
        }
      } else if constexpr (EPI == MH_EPI_RESID) {
        v[0] += old4[jj].x; v[1] += old4[jj].y; v[2] += old4[jj].z; v[3] += old4[jj].w;
      } else if constexpr (EPI == MH_EPI_GATE_RESID) {
        v[0] = old4[jj].x + g4[jj].x * v[0]; v[1] = old4[jj].y + g4[jj].y * v[1];
        v[2] = old4[jj].z + g4[jj].z * v[2]; v[3] = old4[jj].w + g4[jj].w * v[3];
      } else if constexpr (EPI == MH_EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast(v[r]);
      } else if constexpr (EPI == MH_EPI_BIAS_GELU_ERF) {
        const float gq[4] = {g4[jj].x, g4[jj].y, g4[jj].z, g4[jj].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]) + gq[r];    // (bf16 / MX outputs: see gelu_erf_fast)
      }
      asm volatile("" : "+v"(v));
      vv[jj] = v;
    }
    if constexpr (MXO) {
      // The result as the next GEMM's MX-fp8 A operand (mx8.hip's rule on the bf16-ROUNDED values, so the bytes are those of
      // mh_quantize_mx8 over the bf16 matrix this epilogue would have written).  A 32-column block of the output row is held by
      // the 4 lanes lgc = 0..3 of this row (4 consecutive columns each) in two accumulator tiles: jj = {2b, 2b + 1} (BIAS_GELU:
      // columns wc*64 + b*32 ..) or jj = {0, 2} (GEGLU: the wave's 32 output columns are one block).
      constexpr int NBLK = EPI == MH_EPI_GEGLU ? 1 : 2;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int j0 = EPI == MH_EPI_GEGLU ? 0 : 2 * b, j1 = EPI == MH_EPI_GEGLU ? 2 : 2 * b + 1;
        float x[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x[r] = Elem<T>::to_f32(Elem<T>::from_f32(vv[j0][r]));
          x[4 + r] = Elem<T>::to_f32(Elem<T>::from_f32(vv[j1][r]));
        }
        float am = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) am = fmaxf(am, fabsf(x[r]));
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        int e = am > 0.f ? mx8_exponent(am) : -127;
        e = e < -127 ? -127 : (e > 127 ? 127 : e);
        const float inv = mx8_inv_scale(e);
        int o0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0] * inv, x[1] * inv, 0, false);
        o0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[2] * inv, x[3] * inv, o0, true);
        int o1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[4] * inv, x[5] * inv, 0, false);
        o1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[6] * inv, x[7] * inv, o1, true);
        // output column of this lane's first value in tile j0 / j1
        const int cbase = EPI == MH_EPI_GEGLU ? (n0 + wc * WN) / 2 + lgc * 4 : n0 + wc * WN + b * 32 + lgc * 4;
        const int width = EPI == MH_EPI_GEGLU ? p.N / 2 : p.N;
        if (rok && cbase < width) {
          uint8_t* qrow = p.mxq + (long)row * p.ldc;
          *reinterpret_cast<uint32_t*>(qrow + cbase) = (uint32_t)o0;
          *reinterpret_cast<uint32_t*>(qrow + cbase + 16) = (uint32_t)o1;
          if (lgc == 0) p.mxs[(long)row * p.mxs_b + mx8_scale_index(cbase)] = (uint8_t)(e + 127);
        }
      }
      continue;
    }
    // ---- bf16 outputs, WIDE form: the four lanes (lgc = 0..3) of an output row hold 4 consecutive columns each of every 16-column
    // block jj.  One v_permlane16_swap per dword on the registers of two blocks (ja, jb) leaves lanes lgc = 0 / 2 with columns
    // 0-7 / 8-15 of block ja and lanes 1 / 3 with those of block jb: ONE 16-byte store per lane and pair, 64 contiguous bytes per
    // row and instruction instead of 32.  The write phase of these kernels is paced by the memory system, not by the waves:
    // tools/micro/store_pattern_probe.hip -- 184 MB in 52 us (3.5 TB/s) with the 8-byte form, 30 us (6.1 TB/s) with this one.
    // Same values to the same addresses.
    constexpr bool kWideCapable = (EPI == MH_EPI_STORE || EPI == MH_EPI_BIAS_GELU || EPI == MH_EPI_BIAS_GELU_ERF || EPI == MH_EPI_GEGLU ||
                                   EPI == MH_EPI_KV_SCATTER || EPI == MH_EPI_QKV_VT);
    if constexpr (kWideCapable) {
      const bool wide_ok = (p.N % 16 == 0) && (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                           (EPI != MH_EPI_QKV_VT || p.n_split % 32 == 0);    // kernel-uniform
      if (wide_ok) {
        uint2 w[NI];
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) w[jj] = make_uint2(pack_bf16x2(vv[jj][0], vv[jj][1]), pack_bf16x2(vv[jj][2], vv[jj][3]));
        const int blk = lgc & 1, half = lgc >> 1;
#pragma unroll
        for (int pr = 0; pr < (EPI == MH_EPI_GEGLU ? 1 : 2); ++pr) {
          constexpr bool kG = EPI == MH_EPI_GEGLU;
          const int ja = kG ? 0 : 2 * pr, jb = kG ? 2 : 2 * pr + 1;
          if constexpr (EPI == MH_EPI_QKV_VT) {
            if (n0 + wc * WN + ja * 16 >= p.n_split) {          // (wave-uniform) the V columns: transposed scatter, as below
#pragma unroll
              for (int jj = ja; jj <= jb; ++jj) {
                const int c = ecol_base + jj * 16;
                if (!rok || c >= p.N) continue;
                const int c2 = c - p.n_split;
                const int b = row / p.kv_L, key = row - b * p.kv_L;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  reinterpret_cast<T*>(p.C2)[((long)b * p.kv_H * 64 + c2 + r) * p.kv_Lpad + key] = Elem<T>::from_f32(vv[jj][r]);
              }
              continue;
            }
          }
          const auto s0 = __builtin_amdgcn_permlane16_swap(w[ja].x, w[jb].x, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(w[ja].y, w[jb].y, false, false);
          const uint4 q = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          if constexpr (kG) {
            const int oc = (n0 + wc * WN) / 2 + blk * 16 + half * 8;
            if (rok && oc < p.N / 2) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.C) + (long)row * p.ldc + oc) = q;
          } else {
            const int c = n0 + wc * WN + (ja + blk) * 16 + half * 8;
            if (rok && c < p.N) {
              if constexpr (EPI == MH_EPI_KV_SCATTER) {
                const int dd = c & 63, h = (c >> 6) % p.kv_H, lk = (c >> 6) / p.kv_H;
                const int b = row / p.kv_L, key = row - b * p.kv_L;
                *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.C) + ((((long)lk * p.kv_B + b) * p.kv_H + h) * p.kv_L + key) * 64 + dd) = q;
              } else {
                *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.C) + (long)row * p.ldc + c) = q;
              }
            }
          }
        }
        continue;
      }
    }
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
      const int c = ecol_base + jj * 16;
      const float v[4] = {vv[jj][0], vv[jj][1], vv[jj][2], vv[jj][3]};
      if constexpr (EPI == MH_EPI_GEGLU) {
        if (jj & 1) continue;
        const int oc = (n0 + wc * WN) / 2 + (jj / 2) * 16 + lgc * 4;
        if (rok && oc < p.N / 2) st_bf16x4(p.C, (long)row * p.ldc + oc, v);
        continue;
      }
      if (!rok || c >= p.N) continue;
      if constexpr (EPI == MH_EPI_STORE || EPI == MH_EPI_BIAS_GELU || EPI == MH_EPI_BIAS_GELU_ERF) {
        st_bf16x4(p.C, (long)row * p.ldc + c, v);
      } else if constexpr (EPI == MH_EPI_STORE_F32 || EPI == MH_EPI_RESID || EPI == MH_EPI_GATE_RESID) {
        st_f32x4(p.C, (long)row * p.ldc + c, v);
      } else if constexpr (EPI == MH_EPI_KV_SCATTER) {
        const int dd = c & 63, h = (c >> 6) % p.kv_H, lk = (c >> 6) / p.kv_H;
        const int b = row / p.kv_L, key = row - b * p.kv_L;
        st_bf16x4(p.C, ((((long)lk * p.kv_B + b) * p.kv_H + h) * p.kv_L + key) * 64 + dd, v);
      } else if constexpr (EPI == MH_EPI_QKV_VT) {
        if (c < p.n_split) {
          st_bf16x4(p.C, (long)row * p.ldc + c, v);
        } else {
          const int c2 = c - p.n_split;
          const int b = row / p.kv_L, key = row - b * p.kv_L;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            reinterpret_cast<T*>(p.C2)[((long)b * p.kv_H * 64 + c2 + r) * p.kv_Lpad + key] = Elem<T>::from_f32(v[r]);
        }
      } else if constexpr (EPI == MH_EPI_QKV_CACHE) {
        if (c < p.n_split) {
          st_bf16x4(p.C, (long)row * p.ldc + c, v);
        } else {
          const int inner = p.kv_H * 64;
          const int c2 = c - p.n_split;
          const int kv = c2 / inner, cc = c2 - kv * inner;
          const int b = row / p.kv_L, ip = row - b * p.kv_L;
          const long dst = (((long)b * p.kv_H + (cc >> 6)) * p.cache_len + ip) * 64 + (cc & 63);
          if (kv == 0) {
            st_bf16x4(p.C2, dst, v);
          } else {
            st_bf16x4(p.C3, dst, v);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              reinterpret_cast<T*>(p.C4)[((long)b * p.kv_H * 64 + cc + r) * p.kv_Lpad + ip] = Elem<T>::from_f32(v[r]);
          }
        }
      }
    }
  }
}

// ---- G3: bf16, 256 x 128 tile, 8 waves as 4(M) x 2(N), THREE LDS stages filled by LDS-DMA two K steps ahead --------------
// The two-stage kernel above waits for the whole next stage at the end of every K step (vmcnt(0) + barrier): with the
// 0.2 us of MFMA work a step holds, a step costs one loaded memory round trip (~2.2 us measured on the encoder GEMMs, two
// workgroups per CU: MfmaUtil 17-28 %).  Here the DMA of step kt + 2 is issued before step kt is multiplied and the wait
// at the end of step kt only covers step kt + 1 (counted vmcnt: the 6 newest DMA instructions of this wave stay in
// flight), so two stages (96 KB) are always on their way.  One workgroup per CU (144 KB of LDS), two waves per SIMD:
// one wave's ds_reads run under the other's MFMAs.  Same XOR-swizzled 128-byte rows, same fragment reads, same
// epilogues as the 128 x 128 LDS-DMA form.
// MI = 4: 256 x 128 tile (wave tile 64 x 64); MI = 2: 128 x 128 tile (wave tile 32 x 64) for grids that would leave most CUs
// idle at 256 rows per tile (the DiT's N = 384 projections: 96 -> 192 workgroups).
template <int EPI, int MI>
__global__ __launch_bounds__(512) void gemm_glds3_kernel(GemmP p) {
  using T = bf16_t;
  constexpr int BM = 64 * MI, BN = 128, BK = 64, KM = 32, NST = 3;
  constexpr int kRowStride = 128, kStage = (BM + BN) * kRowStride;
  constexpr int WM = 16 * MI, WN = 64, NI = 4;
  constexpr int NA = MI;                     // A-tile DMA instructions per wave and stage (BM / 8 row groups over 8 waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {   // block b runs on XCD b % 8: give every XCD a contiguous range of the work list (bijective)
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int bm, bn;
  {   // groups of GM row panels (<= ~2.5 MB of A), inside a group the row panel runs fastest: W panels are fetched once per group
    const long panel_bytes = (long)BM * p.K * (long)sizeof(T);
    int GM = (int)((5L << 19) / (panel_bytes > 0 ? panel_bytes : 1));
    GM = GM < 2 ? 2 : (GM > 16 ? 16 : GM);
    const int per_group = GM * nbn;
    const int grp = bid / per_group, rem = bid - grp * per_group;
    const int gm = (nbm - grp * GM) < GM ? (nbm - grp * GM) : GM;
    bn = rem / gm;
    bm = grp * GM + (rem - bn * gm);
  }
  const int m0 = bm * BM, n0 = bn * BN;
  const int nk = (p.K + BK - 1) / BK;

  // K tile kt of both operands -> stage st: 4 (A) + 2 (B) wave instructions of 8 rows x 128 bytes.  K % 64 == 0 (dispatch
  // condition): no K tail, so a step is six loads from six per-lane pointers that advance by 128 bytes -- no scalar
  // loads and no branches inside the K loop (either would make hipcc's counter bookkeeping fall back to lgkmcnt(0) waits
  // in front of the MFMAs, serialising the fragment reads below with them).
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const char* srcp[NA + 2];
  {
    const int r8 = lane >> 3;
    const long k_off = (long)(((lane & 7) ^ r8) * 8) * sizeof(T);     // pre-swizzled source chunk (tile row & 7 == r8)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int ra_ = m0 + (i * 8 + wid) * 8 + r8; ra_ = ra_ < p.M ? ra_ : p.M - 1;
      srcp[i] = p.A + (long)ra_ * p.lda_b + k_off;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int rb_ = n0 + (i * 8 + wid) * 8 + r8; rb_ = rb_ < p.N ? rb_ : p.N - 1;
      srcp[NA + i] = p.W + (long)rb_ * p.ldw_b + k_off;
    }
  }
  auto issue = [&](int st) {          // the NEXT K tile (the pointers advance)
    char* base = smem + st * kStage;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)srcp[i], (lptr_t)(base + (i * 8 + wid) * 1024), 16, 0, 0);
      srcp[i] += BK * sizeof(T);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)srcp[NA + i], (lptr_t)(base + BM * kRowStride + (i * 8 + wid) * 1024), 16, 0, 0);
      srcp[NA + i] += BK * sizeof(T);
    }
  };

  // The product is accumulated TRANSPOSED (W fragment as the MFMA's A operand): acc[j][i][r] =
  // C[row m0 + wr*64 + i*16 + (lane & 15)][col n0 + wc*64 + j*16 + (lane >> 4)*4 + r] -- a lane owns 4 CONSECUTIVE
  // columns of one row, so the epilogue writes 8-byte (bf16) / 16-byte (fp32) pieces, 16 stores per lane instead of 64.
  f32x4_t acc[NI][MI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  issue(0);
  if (nk > 1) {
    issue(1);
    if constexpr (MI == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // stage 0 of this wave has landed, stage 1 may still fly
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // raw s_barrier: __syncthreads() carries a workgroup fence, and hipcc waits for EVERY outstanding LDS-DMA (vmcnt(0)) in
  // front of it -- the very wait this kernel exists to avoid.  The volatile asm statements keep the LDS reads below it.
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int frow = lane & 15, sw = frow & 7, lgc = lane >> 4;
  // fragments of K sub-step `ks` (32 of the 64 columns of a stage) for this wave's 64 x 64 tile.  The reads are inline
  // asm on purpose: hipcc's own counter bookkeeping puts lgkmcnt(0) in front of the first MFMA of a phase (it does not
  // count across the loop edge), which would make the 8 reads just issued for the NEXT phase finish before the MFMAs
  // of this one start.  With the reads invisible to it, the waits below are the only ones: lgkmcnt(8) = "everything but
  // the 8 newest reads has arrived" (LDS reads return in order).
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const uint32_t a_off = (uint32_t)((wr * WM + frow) * kRowStride), b_off = (uint32_t)((BM + wc * WN + frow) * kRowStride);
  auto ldfrag = [&](int st, int ks, bf16x8_t (&af)[MI], bf16x8_t (&bf)[NI]) {
    const uint32_t coff = (uint32_t)(((ks * 4 + lgc) ^ sw) * 16);
    const uint32_t pa = lds0 + st * kStage + a_off + coff, pb = lds0 + st * kStage + b_off + coff;
    asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(pa));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[1]) : "v"(pa));
    if constexpr (MI == 4) {
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[2]) : "v"(pa));
      asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[3]) : "v"(pa));
    }
    asm volatile("ds_read_b128 %0, %1" : "=v"(bf[0]) : "v"(pb));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(bf[1]) : "v"(pb));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(bf[2]) : "v"(pb));
    asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(bf[3]) : "v"(pb));
  };
  auto mma = [&](const bf16x8_t (&af)[MI], const bf16x8_t (&bf)[NI]) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[j][i], 0, 0, 0);
  };
  // Two phases per K step, each = 8 fragment reads in flight under 16 MFMAs of the fragments read one phase earlier:
  //   phase A: DMA of step kt + 2 | reads of (stage kt, ks 1) | MFMAs of (stage kt, ks 0) | lgkmcnt(0), vmcnt(6), barrier
  //   phase B: reads of (stage kt + 1, ks 0)                  | MFMAs of (stage kt, ks 1)
  // The lgkmcnt(0) in front of the barrier retires this wave's last reads of stage kt, so the DMA that the fastest wave
  // issues into that buffer two phases later (step kt + 1's phase A, stage kt + 3) cannot overtake them.
  bf16x8_t a0[MI], b0[NI], a1[MI], b1[NI];
  ldfrag(0, 0, a0, b0);
  int cur = 0, nxt2 = 2;
  // every wait of the loop is written out; a sched_barrier after each keeps hipcc from moving MFMAs (register-only, so a
  // "memory" clobber does not hold them) above the wait that makes their operands valid
#define G3_WAIT(str) do { asm volatile(str ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
  for (int kt = 0; kt + 2 < nk; ++kt) {            // steady state: a DMA every step
    issue(nxt2);
    ldfrag(cur, 1, a1, b1);
    if constexpr (MI == 4) G3_WAIT("s_waitcnt lgkmcnt(8)");               // a0 / b0 (read one phase ago) are in
    else G3_WAIT("s_waitcnt lgkmcnt(6)");
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MI == 4) G3_WAIT("s_waitcnt vmcnt(6) lgkmcnt(0)");      // a1 / b1 are in; stage kt + 1 of this wave has landed
    else G3_WAIT("s_waitcnt vmcnt(4) lgkmcnt(0)");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cur = cur == NST - 1 ? 0 : cur + 1;
    nxt2 = nxt2 == NST - 1 ? 0 : nxt2 + 1;
    ldfrag(cur, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (nk > 1) {                                    // step nk - 2: nothing left to request, step nk - 1 must have landed
    ldfrag(cur, 1, a1, b1);
    if constexpr (MI == 4) G3_WAIT("s_waitcnt lgkmcnt(8)");
    else G3_WAIT("s_waitcnt lgkmcnt(6)");
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    G3_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cur = cur == NST - 1 ? 0 : cur + 1;
    ldfrag(cur, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  ldfrag(cur, 1, a1, b1);                          // last step
  if constexpr (MI == 4) G3_WAIT("s_waitcnt lgkmcnt(8)");
  else G3_WAIT("s_waitcnt lgkmcnt(6)");
  mma(a0, b0);
  __builtin_amdgcn_sched_barrier(0);
  G3_WAIT("s_waitcnt lgkmcnt(0)");
  mma(a1, b1);
#undef G3_WAIT

  g3_epilogue<EPI, MI>(p, acc, m0, n0, wr, wc, lane);
}

// ---- G2S: bf16, 128 x 128 tile, TWO LDS stages (64 KB): two workgroups per CU ----------------------------------------------------
// For SHORT K (the batched DiT's K = 384 / 768 projections: 6-12 K steps) a tile's prologue (the first stage's round trip) and
// epilogue (stores, residual / gate reads) are as long as its K loop, and the three-stage kernels hold a whole CU (96-144 KB of
// LDS): nothing runs under them.  Two stages of (128 + 128) rows x 128 bytes are 64 KB, the wave tile 32 x 64 needs ~90 VGPRs:
// two workgroups share a CU and one's prologue / epilogue runs under the other's K loop.  The DMA of step kt + 1 is issued at the
// start of step kt (its buffer held step kt - 1, retired by every wave before that step's barrier) and waited for at the
// barrier in the middle of step kt.  Same swizzle, fragment reads, MFMA order and epilogues as gemm_glds3_kernel<EPI, 2>:
// bit-identical results.  Dispatch: K <= option gemm_2stage_max_k.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_glds2s_kernel(GemmP p) {
  using T = bf16_t;
  constexpr int MI = 2, NI = 4;
  constexpr int BM = 128, BN = 128, BK = 64;
  constexpr int kRowStride = 128, kStage = (BM + BN) * kRowStride;
  constexpr int WM = 16 * MI, WN = 64;
  constexpr int NA = 2, NB = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {   // block b runs on XCD b % 8: give every XCD a contiguous range of the work list (bijective)
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int bm, bn;
  {
    const long panel_bytes = (long)BM * p.K * (long)sizeof(T);
    int GM = (int)((5L << 19) / (panel_bytes > 0 ? panel_bytes : 1));
    GM = GM < 2 ? 2 : (GM > 16 ? 16 : GM);
    const int per_group = GM * nbn;
    const int grp = bid / per_group, rem = bid - grp * per_group;
    const int gm = (nbm - grp * GM) < GM ? (nbm - grp * GM) : GM;
    bn = rem / gm;
    bm = grp * GM + (rem - bn * gm);
  }
  const int m0 = bm * BM, n0 = bn * BN;
  const int nk = (p.K + BK - 1) / BK;

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const char* a_base = p.A + (long)m0 * p.lda_b;
  const char* w_base = p.W + (long)n0 * p.ldw_b;
  uint32_t soff[NA + NB];
  {
    const int r8 = lane >> 3;
    const uint32_t k_off = (uint32_t)(((lane & 7) ^ r8) * 16);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int ra_ = (i * 8 + wid) * 8 + r8; ra_ = m0 + ra_ < p.M ? ra_ : p.M - 1 - m0;
      soff[i] = (uint32_t)ra_ * (uint32_t)p.lda_b + k_off;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int rb_ = (i * 8 + wid) * 8 + r8; rb_ = n0 + rb_ < p.N ? rb_ : p.N - 1 - n0;
      soff[NA + i] = (uint32_t)rb_ * (uint32_t)p.ldw_b + k_off;
    }
  }
  auto issue = [&](int st) {
    char* base = smem + st * kStage;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a_base + soff[i]), (lptr_t)(base + (i * 8 + wid) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(w_base + soff[NA + i]), (lptr_t)(base + BM * kRowStride + (i * 8 + wid) * 1024), 16, 0, 0);
    a_base += BK * sizeof(T);
    w_base += BK * sizeof(T);
  };

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int frow = lane & 15, sw = frow & 7, lgc = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const uint32_t a_off = (uint32_t)((wr * WM + frow) * kRowStride), b_off = (uint32_t)((BM + wc * WN + frow) * kRowStride);
  auto ldfrag = [&](int st, int ks, bf16x8_t (&af)[MI], bf16x8_t (&bf)[NI]) {
    const uint32_t coff = (uint32_t)(((ks * 4 + lgc) ^ sw) * 16);
    const uint32_t pa = lds0 + st * kStage + a_off + coff, pb = lds0 + st * kStage + b_off + coff;
    asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(pa));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[1]) : "v"(pa));
    asm volatile("ds_read_b128 %0, %1" : "=v"(bf[0]) : "v"(pb));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(bf[1]) : "v"(pb));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(bf[2]) : "v"(pb));
    asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(bf[3]) : "v"(pb));
  };
  auto mma = [&](const bf16x8_t (&af)[MI], const bf16x8_t (&bf)[NI]) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[j][i], 0, 0, 0);
  };
#define G2_WAIT(str) do { asm volatile(str ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
  bf16x8_t a0[MI], b0[NI], a1[MI], b1[NI];
  ldfrag(0, 0, a0, b0);
  int cur = 0;
  for (int kt = 0; kt + 1 < nk; ++kt) {
    issue(cur ^ 1);
    ldfrag(cur, 1, a1, b1);
    G2_WAIT("s_waitcnt lgkmcnt(6)");               // a0 / b0 (read one phase ago) are in
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    G2_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)");      // a1 / b1 are in; this wave's pieces of step kt + 1 have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cur ^= 1;
    ldfrag(cur, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  ldfrag(cur, 1, a1, b1);                          // last step
  G2_WAIT("s_waitcnt lgkmcnt(6)");
  mma(a0, b0);
  __builtin_amdgcn_sched_barrier(0);
  G2_WAIT("s_waitcnt lgkmcnt(0)");
  mma(a1, b1);
#undef G2_WAIT
  g3_epilogue<EPI, MI>(p, acc, m0, n0, wr, wc, lane);
}

// ---- G4: bf16, 256 x 256 tile, 8 waves as 2 (M) x 4 (N) of 128 x 64, TWO LDS stages, two staggered wave groups --------------------
// gemm_glds3_kernel is bound neither by LDS bandwidth (its XOR-swizzled fragment reads are conflict-free: 4 LDS cycles each under
// the quarter-wave lane groups of MI355X_MICROARCH.md, 512 cycles per K step and CU) nor by the matrix pipe (1 024 cycles per K
// step and SIMD of 2 448 measured) but by what a wave issues BESIDES its MFMAs: per K step 6 LDS-DMA pieces (60-185 cycles of
// issue each), 16 fragment reads, two counted waits and a barrier against 32 MFMAs -- and all eight waves do those at the same
// time.  This kernel (a) doubles the wave tile to 128 x 64: 0.375 fragment reads per MFMA instead of 0.5, 8 pieces per 64 MFMAs
// instead of 6 per 32; the 128 accumulator registers per lane live in AGPRs (MFMAs written as asm with the accumulator tied in
// place, as in gemm_mx8_kernel), the two fragment sets in the 128 VGPRs beside them; (b) runs its two wave groups (waves w and
// w + 4 share a SIMD) one segment apart, so that one wave of a SIMD issues loads while the other has the matrix pipe (schedule
// at the K loop).  64 KB per stage leave room for two stages; the pieces of step k + 2 are issued right after the last read of
// step k and have a whole step to land.  Bit-identical to gemm_glds3_kernel (same products, same order, same epilogues:
// tests/test_gpu_kernels.py).  Used for grids of at least option gemm_tile256sq_min 256 x 256 tiles whose rounds of 256
// workgroups are >= 88 % full (the encoder's GEMMs at 32 chunks).  Measured (profiles/r04_gemm_256sq.txt): 8192^3 1 052 ->
// 1 157 TFLOP/s, base qkv 191 -> 170 us, wi 344 -> 305, wo 129 -> 113, cross-K/V 1 480 -> 1 391.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_glds4_kernel(GemmP p) {
  using T = bf16_t;
  constexpr int MI = 8, NI = 4;
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int kRowStride = 128, kStage = (BM + BN) * kRowStride;
  constexpr int WM = 16 * MI, WN = 64;
  constexpr int NA = 4, NB = 4;              // LDS-DMA instructions per wave and stage and operand (32 groups of 8 rows over 8 waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 2, wc = wid & 3;

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {   // block b runs on XCD b % 8: give every XCD a contiguous range of the work list (bijective)
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int bm, bn;
  {   // groups of GM row panels (<= ~2.5 MB of A), inside a group the row panel runs fastest: W panels are fetched once per group
    const long panel_bytes = (long)BM * p.K * (long)sizeof(T);
    int GM = (int)((5L << 19) / (panel_bytes > 0 ? panel_bytes : 1));
    GM = GM < 2 ? 2 : (GM > 16 ? 16 : GM);
    const int per_group = GM * nbn;
    const int grp = bid / per_group, rem = bid - grp * per_group;
    const int gm = (nbm - grp * GM) < GM ? (nbm - grp * GM) : GM;
    bn = rem / gm;
    bm = grp * GM + (rem - bn * gm);
  }
  const int m0 = bm * BM, n0 = bn * BN;
  const int nk = (p.K + BK - 1) / BK;        // K % 64 == 0 (dispatch condition)

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // sources: a block-uniform 64-bit base per operand (advanced by SALU every K step) + ONE 32-bit byte offset per lane and piece
  // -- 8 offset registers instead of 16 pointer registers, and no VALU in the K loop for them
  const char* a_base = p.A + (long)m0 * p.lda_b;
  const char* w_base = p.W + (long)n0 * p.ldw_b;
  uint32_t soff[NA + NB];
  {
    const int r8 = lane >> 3;
    const uint32_t k_off = (uint32_t)(((lane & 7) ^ r8) * 16);       // pre-swizzled source chunk (tile row & 7 == r8)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int ra_ = (i * 8 + wid) * 8 + r8; ra_ = m0 + ra_ < p.M ? ra_ : p.M - 1 - m0;
      soff[i] = (uint32_t)ra_ * (uint32_t)p.lda_b + k_off;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int rb_ = (i * 8 + wid) * 8 + r8; rb_ = n0 + rb_ < p.N ? rb_ : p.N - 1 - n0;
      soff[NA + i] = (uint32_t)rb_ * (uint32_t)p.ldw_b + k_off;
    }
  }
  auto issue = [&](int st) {          // the NEXT K tile (the bases advance)
    char* base = smem + st * kStage;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a_base + soff[i]), (lptr_t)(base + (i * 8 + wid) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(w_base + soff[NA + i]), (lptr_t)(base + BM * kRowStride + (i * 8 + wid) * 1024), 16, 0, 0);
    a_base += BK * sizeof(T);
    w_base += BK * sizeof(T);
  };

  // transposed product as in gemm_glds3_kernel: acc[j][i][r] = C[m0 + wr*128 + i*16 + (lane & 15)][n0 + wc*64 + j*16 + (lane >> 4)*4 + r]
  f32x4_t acc[NI][MI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // STAGGERED wave groups.  Waves w and w + 4 share a SIMD; group g = wid >> 2 (= wr).  A K step is four segments separated by
  // barriers -- ISSUE_A (12 fragment reads), MFMA0 (32 MFMAs), ISSUE_B (8 LDS-DMA pieces of step k + 2, 12 fragment reads), MFMA1
  // (32 MFMAs) -- and group 1 runs ONE segment behind group 0 (one extra barrier in front of its loop, one behind group 0's):
  // while one wave of a SIMD issues loads, the other has the matrix pipe to itself.
  //   slot 4k: G0 ISSUE_A(k) | G1 MFMA1(k-1)     slot 4k+1: G0 MFMA0(k) | G1 ISSUE_A(k)
  //   slot 4k+2: G0 ISSUE_B(k) | G1 MFMA0(k)     slot 4k+3: G0 MFMA1(k) | G1 ISSUE_B(k)
  // Stage k % 2 is read in ISSUE_B(k-1) (ks 0) and ISSUE_A(k) (ks 1), last by G1 in slot 4k+1, every read retired (lgkmcnt(0))
  // before that slot's barrier; the pieces of step k + 2 go into it from slot 4k+2 (G0) / 4k+3 (G1) on, are waited for
  // (vmcnt(0)) at the end of each wave's ISSUE_A(k+1) (slots 4k+4 / 4k+5) and first read in slot 4k+6.
  issue(0);
  if (nk > 1) {
    issue(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this wave's pieces of step 0 have landed
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int frow = lane & 15, sw = frow & 7, lgc = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const uint32_t a_off = (uint32_t)((wr * WM + frow) * kRowStride), b_off = (uint32_t)((BM + wc * WN + frow) * kRowStride);
#define G4_RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
  auto ldfrag = [&](int st, int ks, bf16x8_t (&af)[MI], bf16x8_t (&bf)[NI]) {
    const uint32_t coff = (uint32_t)(((ks * 4 + lgc) ^ sw) * 16);
    const uint32_t pa = lds0 + st * kStage + a_off + coff, pb = lds0 + st * kStage + b_off + coff;
    G4_RD(bf[0], pb, 0); G4_RD(bf[1], pb, 2048); G4_RD(bf[2], pb, 4096); G4_RD(bf[3], pb, 6144);
    G4_RD(af[0], pa, 0); G4_RD(af[1], pa, 2048); G4_RD(af[2], pa, 4096); G4_RD(af[3], pa, 6144);
    G4_RD(af[4], pa, 8192); G4_RD(af[5], pa, 10240); G4_RD(af[6], pa, 12288); G4_RD(af[7], pa, 14336);
  };
#undef G4_RD
  auto mma = [&](bf16x8_t (&af)[MI], bf16x8_t (&bf)[NI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(bf[j]), "v"(af[i]));
  };
#define G4_WAIT(str) do { asm volatile(str ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G4_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
  bf16x8_t a0[MI], b0[NI], a1[MI], b1[NI];
  ldfrag(0, 0, a0, b0);
  if (wr == 1) G4_BARRIER();                        // group 1: one segment behind
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // ISSUE_A
    ldfrag(cur, 1, a1, b1);
    G4_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)");       // a0 / b0 / a1 / b1 are in; this wave's pieces of step kt + 1 have landed
    G4_BARRIER();
    // MFMA0
    mma(a0, b0);
    G4_BARRIER();
    // ISSUE_B
    if (kt + 2 < nk) issue(cur);                    // (stage of step kt: every wave retired its reads of it two barriers ago or more)
    if (kt + 1 < nk) ldfrag(cur ^ 1, 0, a0, b0);
    G4_BARRIER();
    // MFMA1
    mma(a1, b1);
    G4_BARRIER();
    cur ^= 1;
  }
  if (wr == 0) G4_BARRIER();
#undef G4_BARRIER
#undef G4_WAIT
  // the MFMAs are asm: hipcc does not know that the accumulators come out of the matrix pipe.  It pads nothing in front of
  // their first read -- and it is free to move a v_accvgpr_read of an accumulator up to right behind the asm statement that last
  // wrote it, i.e. in between the final MFMAs (seen: the GEGLU epilogue read acc[0][0] there and got the value of a step ago).
  // The sched_barriers pin every read below the padding.
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  g3_epilogue<EPI, MI>(p, acc, m0, n0, wr, wc, lane);
}

// ---- S3G: the bf16 x 3 fp32 GEMM on the three-stage LDS-DMA structure, BOTH operands pre-split -----------------------
// The 64 x 64 split-3 tile above converts its fp32 A operand into [hi | lo] while staging it (registers, VALU, LDS
// writes), one stage ahead: 41-57 us per GEMM of the batched DiT.  When the PRODUCER of the activations writes them
// pre-split -- the same [32 x bf16 hi | 32 x bf16 lo] per 32 values as the weights, the same bytes as fp32 -- a K step is
// one 128-byte row per operand row, i.e. exactly what gemm_glds3_kernel moves by LDS-DMA: its "ks = 0 / 1" fragment reads
// are the hi / lo halves.  Per 32-k step and wave: 2 (MI + 4) fragment reads, 3 * 4 * MI MFMAs (lo.hi, hi.lo, hi.hi -- the
// order of the 64 x 64 kernel), reads of step kt + 1 issued before the MFMAs of step kt into a second register set.
// Epilogues: QKV_VT and GATE_RESID write fp32 (attention operands, residual stream), BIAS_GELU writes the next GEMM's A
// operand pre-split again (p.split3 & 4).
template <int EPI, int MI>
__global__ __launch_bounds__(512) void gemm_s3g_kernel(GemmP p) {
  static_assert(MI == 2, "MI = 4 spills fragment registers that inline-asm LDS reads are still filling");
  constexpr int BM = 64 * MI, BN = 128, NST = 3;
  constexpr int kRowStride = 128, kStage = (BM + BN) * kRowStride;
  constexpr int WM = 16 * MI, WN = 64, NI = 4, NA = MI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  // PERSISTENT over tiles: workgroup w takes tiles w, w + gridDim.x, ... of the XCD-aware work list (gridDim.x is a multiple of
  // 8 or the whole list, so tile t still runs on XCD t % 8).  With one workgroup per CU the DMA latency of a tile's first two
  // stages and its epilogue were dead time (9 of 17 us per tile at K = 384): the NEXT tile's first two stages are requested
  // once the K loop is over (its last barrier retired every read of the stage buffers) and the epilogue has its OWN loads in
  // registers (the load counter is in-order: a bias load issued after the DMAs would wait for them), before the stores.
  auto coords = [&](int t, int& m0_, int& n0_) {
    int bid = t;
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const long panel_bytes = (long)BM * p.K * 4L;
    int GM = (int)((5L << 19) / (panel_bytes > 0 ? panel_bytes : 1));
    GM = GM < 2 ? 2 : (GM > 16 ? 16 : GM);
    const int per_group = GM * nbn;
    const int grp = bid / per_group, rem = bid - grp * per_group;
    const int gm = (nbm - grp * GM) < GM ? (nbm - grp * GM) : GM;
    const int bn = rem / gm;
    m0_ = (grp * GM + (rem - bn * gm)) * BM;
    n0_ = bn * BN;
  };
  const int nk = p.K / 32;                       // one 128-byte block (32 values as hi | lo) per K step

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const char* srcp[NA + 2];
  auto setup = [&](int m0_, int n0_) {
    const int r8 = lane >> 3;
    const long k_off = (long)(((lane & 7) ^ r8) * 16);               // pre-swizzled 16-byte chunk of the 128-byte block
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int ra_ = m0_ + (i * 8 + wid) * 8 + r8; ra_ = ra_ < p.M ? ra_ : p.M - 1;
      srcp[i] = p.A + (long)ra_ * p.lda_b + k_off;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int rb_ = n0_ + (i * 8 + wid) * 8 + r8; rb_ = rb_ < p.N ? rb_ : p.N - 1;
      srcp[NA + i] = p.W + (long)rb_ * p.ldw_b + k_off;
    }
  };
  auto issue = [&](int st) {
    char* base = smem + st * kStage;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)srcp[i], (lptr_t)(base + (i * 8 + wid) * 1024), 16, 0, 0);
      srcp[i] += 128;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)srcp[NA + i], (lptr_t)(base + BM * kRowStride + (i * 8 + wid) * 1024), 16, 0, 0);
      srcp[NA + i] += 128;
    }
  };

  f32x4_t acc[NI][MI];                           // transposed product, as in gemm_glds3_kernel
  const int frow = lane & 15, sw = frow & 7, lgc = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const uint32_t a_off = (uint32_t)((wr * WM + frow) * kRowStride), b_off = (uint32_t)((BM + wc * WN + frow) * kRowStride);
  struct Frags { bf16x8_t ah[MI], al[MI], bh[NI], bl[NI]; };
  auto rd = [&](int st, Frags& f) {              // hi halves = chunks 0..3, lo halves = chunks 4..7 of the (swizzled) block
    const uint32_t ch = (uint32_t)((lgc ^ sw) * 16), cl = (uint32_t)(((4 + lgc) ^ sw) * 16);
    const uint32_t pa = lds0 + st * kStage + a_off, pb = lds0 + st * kStage + b_off;
    const uint32_t pah = pa + ch, pal = pa + cl, pbh = pb + ch, pbl = pb + cl;
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.ah[0]) : "v"(pah));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(f.ah[1]) : "v"(pah));
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.al[0]) : "v"(pal));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(f.al[1]) : "v"(pal));
    if constexpr (MI == 4) {
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.ah[2]) : "v"(pah));
      asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(f.ah[3]) : "v"(pah));
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.al[2]) : "v"(pal));
      asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(f.al[3]) : "v"(pal));
    }
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.bh[0]) : "v"(pbh));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(f.bh[1]) : "v"(pbh));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.bh[2]) : "v"(pbh));
    asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(f.bh[3]) : "v"(pbh));
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.bl[0]) : "v"(pbl));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(f.bl[1]) : "v"(pbl));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.bl[2]) : "v"(pbl));
    asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(f.bl[3]) : "v"(pbl));
  };
  auto mma3 = [&](const Frags& f) {              // small terms first, the dominant product last
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bh[j], f.al[i], acc[j][i], 0, 0, 0);
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bl[j], f.ah[i], acc[j][i], 0, 0, 0);
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bh[j], f.ah[i], acc[j][i], 0, 0, 0);
      }
  };
#define S3G_WAIT(str) do { asm volatile(str ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
  // step kt: [DMA of stage kt + 2] wait: stage kt + 1 landed, own LDS reads retired | barrier | reads of stage kt + 1 into
  // the OTHER register set | MFMAs on this set (read one step ago, retired by the wait above)
  Frags f0, f1;
  int t = blockIdx.x;
  int m0, n0;
  coords(t, m0, n0);
  setup(m0, n0);
  issue(0);
  if (nk > 1) issue(1);
  bool first = true;
  int nxt = 1, nxt2 = 2;
  int m0n = 0, n0n = 0;                          // the next tile's origin (set by the last K step)
  auto step = [&](Frags& cur, Frags& other, bool dma, bool more) {
    if (dma) {
      issue(nxt2);
      if constexpr (MI == 4) S3G_WAIT("s_waitcnt vmcnt(6) lgkmcnt(0)"); else S3G_WAIT("s_waitcnt vmcnt(4) lgkmcnt(0)");
    } else {
      S3G_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (more) rd(nxt, other);
    __builtin_amdgcn_sched_barrier(0);
    mma3(cur);
    __builtin_amdgcn_sched_barrier(0);
    nxt = nxt == NST - 1 ? 0 : nxt + 1;
    nxt2 = nxt2 == NST - 1 ? 0 : nxt2 + 1;
  };
  for (; t < nwg; t += (int)gridDim.x) {
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // stage 0 has landed (the first tile may keep stage 1 in flight; later tiles also wait for their predecessor's stores)
  if (first && nk > 1) {
    if constexpr (MI == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  first = false;
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  rd(0, f0);
  nxt = 1; nxt2 = 2;
  int kt = 0;
  for (; kt + 3 < nk; kt += 2) {                 // pairs of steps with a DMA each (register sets alternate)
    step(f0, f1, true, true);
    step(f1, f0, true, true);
  }
  for (; kt < nk; ++kt) {                        // the last <= 3 steps
    const bool dma = kt + 2 < nk, more = kt + 1 < nk;
    if ((kt & 1) == 0) step(f0, f1, dma, more); else step(f1, f0, dma, more);
  }
#undef S3G_WAIT

  // ---- epilogue (4 consecutive columns per lane; loads pinned before the first store of a row block) ----
  const int l15 = lane & 15;
  const int ecol_base = n0 + wc * WN + lgc * 4, erow_base = m0 + wr * WM + l15;
  auto f4 = [](const float* q) { return *reinterpret_cast<const float4*>(q); };
  float4 bias4[NI];
#pragma unroll
  for (int jj = 0; jj < NI; ++jj) {
    int c = ecol_base + jj * 16;
    c = c < p.N ? c : p.N - 4;
    bias4[jj] = p.bias ? f4(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int jj = 0; jj < NI; ++jj)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      acc[jj][i][0] += bias4[jj].x; acc[jj][i][1] += bias4[jj].y; acc[jj][i][2] += bias4[jj].z; acc[jj][i][3] += bias4[jj].w;
      asm volatile("" : "+v"(acc[jj][i]));
    }
  float4 old4[MI][NI], g4[MI][NI];
  if constexpr (EPI == MH_EPI_GATE_RESID) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = erow_base + i * 16;
      const int rowc = row < p.M ? row : p.M - 1;
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) {
        int c = ecol_base + jj * 16;
        c = c < p.N ? c : p.N - 4;
        old4[i][jj] = f4(reinterpret_cast<const float*>(p.C) + (long)rowc * p.ldc + c);
        g4[i][jj] = f4(p.gate + (long)(rowc / p.rows_per_batch) * p.gate_ld + c);
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) asm volatile("" : "+v"(old4[i][jj].x), "+v"(old4[i][jj].y), "+v"(old4[i][jj].z), "+v"(old4[i][jj].w),
                                                        "+v"(g4[i][jj].x), "+v"(g4[i][jj].y), "+v"(g4[i][jj].z), "+v"(g4[i][jj].w));
  }
  if (t + (int)gridDim.x < nwg) {                // every load this epilogue needs has arrived: the next tile's first K tiles start flying
    coords(t + (int)gridDim.x, m0n, n0n);
    setup(m0n, n0n);
    issue(0);
    if (nk > 1) issue(1);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = erow_base + i * 16;
    const bool rok = row < p.M;
    f32x4_t vv[NI];
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
      f32x4_t v = acc[jj][i];
      if constexpr (EPI == MH_EPI_GATE_RESID) {
        const float4 o4 = old4[i][jj], q4 = g4[i][jj];
        v[0] = o4.x + q4.x * v[0]; v[1] = o4.y + q4.y * v[1]; v[2] = o4.z + q4.z * v[2]; v[3] = o4.w + q4.w * v[3];
      } else if constexpr (EPI == MH_EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast(v[r]);
      }
      vv[jj] = v;
    }
    if constexpr (EPI == MH_EPI_BIAS_GELU) {
      // pre-split output, WIDE form (see g3_epilogue): the hi / lo halves of two 16-column blocks of one 32-column group are exchanged
      // across lane rows, every lane then stores 16 bytes of hi and 16 bytes of lo: 64 contiguous bytes per row, half and instruction
      if (p.N % 32 == 0 && p.ldc % 32 == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) {      // kernel-uniform
#pragma unroll
        for (int pr = 0; pr < NI / 2; ++pr) {
          uint32_t hi[2][2], lo[2][2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4_t v = vv[2 * pr + q];
            const uint32_t h01 = pack_bf16x2(v[0], v[1]), h23 = pack_bf16x2(v[2], v[3]);
            const float r0 = v[0] - __uint_as_float(h01 << 16), r1 = v[1] - __uint_as_float(h01 & 0xffff0000u);
            const float r2 = v[2] - __uint_as_float(h23 << 16), r3 = v[3] - __uint_as_float(h23 & 0xffff0000u);
            hi[q][0] = h01; hi[q][1] = h23; lo[q][0] = pack_bf16x2(r0, r1); lo[q][1] = pack_bf16x2(r2, r3);
          }
          const auto a0 = __builtin_amdgcn_permlane16_swap(hi[0][0], hi[1][0], false, false);
          const auto a1 = __builtin_amdgcn_permlane16_swap(hi[0][1], hi[1][1], false, false);
          const auto b0 = __builtin_amdgcn_permlane16_swap(lo[0][0], lo[1][0], false, false);
          const auto b1 = __builtin_amdgcn_permlane16_swap(lo[0][1], lo[1][1], false, false);
          const int lgc_ = lane >> 4;
          const int cblk = n0 + wc * WN + pr * 32;                       // first column of the 32-column group
          if (rok && cblk < p.N) {
            // lanes lgc 0 / 2: block 2 pr, columns 0-7 / 8-15; lanes 1 / 3: block 2 pr + 1 -- byte offset of column c in its group half: (c % 32) * 2
            char* dst = reinterpret_cast<char*>(p.C) + ((long)row * p.ldc + cblk) * 4 + ((lgc_ & 1) * 16 + (lgc_ >> 1) * 8) * 2;
            *reinterpret_cast<uint4*>(dst) = make_uint4(a0[0], a1[0], a0[1], a1[1]);
            *reinterpret_cast<uint4*>(dst + 64) = make_uint4(b0[0], b1[0], b0[1], b1[1]);
          }
        }
        continue;
      }
    }
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
      const int c = ecol_base + jj * 16;
      if (!rok || c >= p.N) continue;
      const f32x4_t v = vv[jj];
      if constexpr (EPI == MH_EPI_BIAS_GELU) {
        // the next GEMM's A operand, pre-split: column c of row `row` -> block c / 32, hi at (c % 32) * 2, lo 64 bytes on
        char* dst = reinterpret_cast<char*>(p.C) + ((long)row * p.ldc + (c & ~31)) * 4 + (c & 31) * 2;
        const uint32_t h01 = pack_bf16x2(v[0], v[1]), h23 = pack_bf16x2(v[2], v[3]);
        const float r0 = v[0] - __uint_as_float(h01 << 16), r1 = v[1] - __uint_as_float(h01 & 0xffff0000u);
        const float r2 = v[2] - __uint_as_float(h23 << 16), r3 = v[3] - __uint_as_float(h23 & 0xffff0000u);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));
      } else if constexpr (EPI == MH_EPI_GATE_RESID || EPI == MH_EPI_STORE_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)row * p.ldc + c) = make_float4(v[0], v[1], v[2], v[3]);
      } else if constexpr (EPI == MH_EPI_QKV_VT) {
        if (c < p.n_split) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)row * p.ldc + c) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          const int c2 = c - p.n_split;
          const int b = row / p.kv_L, key = row - b * p.kv_L;
#pragma unroll
          for (int r = 0; r < 4; ++r) reinterpret_cast<float*>(p.C2)[((long)b * p.kv_H * 64 + c2 + r) * p.kv_Lpad + key] = v[r];
        }
      }
    }
  }
  m0 = m0n; n0 = n0n;
  }   // tile loop
}

// ---- MX8: OCP e4m3 operands + one E8M0 scale per (row, 32 consecutive k) on v_mfma_scale_f32_16x16x128_f8f6f4 -------------
// BASELINE configs[4] ("fp8 MFMA").  The three-stage LDS-DMA structure of gemm_glds3_kernel with the K step re-read: a
// 128-byte row of a stage is 128 e4m3 values = ONE MFMA k (the bf16 form needs two 16x16x32 MFMAs per 64 values), so a K
// step moves the same bytes through HBM / L2 / LDS as the bf16 kernel and multiplies twice the k range in the same matrix-core
// time (32 cycles per MFMA) -- the MX rate is 2x the bf16 rate at equal operand bytes per step.
//   operand layout (measured, tools/micro/mfma_mx_probe*.hip): lane (row = l & 15, lg = l >> 4) holds bytes [lg*16, +16) and
//   [64 + lg*16, +16) of its row's 128-byte k step; the scale register's byte OP_SEL of that lane scales k block lg (32
//   consecutive k) of the row -- for both operands.  Swizzled stage rows as in the bf16 kernels: 16-byte chunk c of tile
//   row r sits at chunk c ^ (r & 7), so a fragment is two conflict-free ds_read_b128.
//   scales: [rows][16 * ceil(K / 512)] bytes, byte (kt / 4) * 16 + lg * 4 + (kt % 4) = E8M0 of k block kt * 4 + lg (kt = K step):
//   a lane fetches ONE dword per row block and four K steps and selects the step's byte with OP_SEL (0..3) -- MI + 4 dword
//   loads per wave and four steps, issued a group ahead by inline asm and covered by the counted vmcnt waits of the DMA pipeline.
// Two geometries of one kernel (NW waves as NW/2 (M) x 2 (N), wave tile 16 MI x 64; a third -- FOUR waves of 128 x 64 with AGPR
// accumulators, 0.99-1.37 x bf16 / 1418 TFLOP/s at 8192^3 against 1.07-1.46 x / 1534 -- was removed in round 5):
//   NW = 8, MI = 4: 256 x 128 tile, EIGHT waves of 64 x 64, two per SIMD (214 VGPRs, accumulators in VGPRs: a kernel that touches
//     AGPRs gets its 256 registers split 128 + 128) -- one wave's fragment reads run under the other's MFMAs.  The default:
//     1.07-1.46 x the bf16 kernel on the encoder / DiT-B shapes, 1534 TFLOP/s at 8192^3
//     (profiles/r04_mx8_gemm_bench.txt).  An earlier form of the K loop spilled asm-loaded fragments at this geometry
//     (tools/check_kernel_resources.py fails the build on that); the rolled loop with one A register set fits.
//   NW = 8, MI = 2: 128 x 128 tile for grids that would leave CUs idle at 256 rows per tile.
// Two phases per K step, split by A row blocks so that the A fragments need ONE register set:
//   phase 1: DMA of step kt + 2 (+ the next scale group every fourth step) | reads of (stage kt: A blocks MI/2 ..) | MFMAs of A
//            blocks 0 .. MI/2 - 1 with all four W blocks | wait (stage kt + 1 landed, every read of stage kt retired), barrier
//   phase 2: reads of (stage kt + 1: A blocks 0 .. MI/2 - 1, the four W blocks into the OTHER W set) | MFMAs of A blocks MI/2 ..
// Transposed product and epilogue as gemm_glds3_kernel (bf16 outputs / fp32 residual stream).
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

template <int EPI, int MI, int NW, bool MXO = false>
__global__ __launch_bounds__(NW * 64) void gemm_mx8_kernel(GemmP p) {
  constexpr int BM = 16 * MI * (NW / 2), BN = 128, NST = 3;
  constexpr int kRowStride = 128, kStage = (BM + BN) * kRowStride;
  constexpr int WM = 16 * MI, WN = 64, NI = 4, MH_ = MI / 2;
  constexpr int NA = BM / 8 / NW, NB = BN / 8 / NW;      // DMA instructions per wave and stage (8 rows x 128 bytes each)
  constexpr int NDMA = NA + NB, NSC = MI + NI;
  static_assert(MI % 2 == 0 && NSC <= 12 && 2 * MH_ + 2 * NI <= 28, "operand lists of the wait statements");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {   // block b runs on XCD b % 8: give every XCD a contiguous range of the work list (bijective)
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int bm, bn;
  {   // groups of GM row panels (<= ~2.5 MB of A), inside a group the row panel runs fastest
    const long panel_bytes = (long)BM * p.K;
    int GM = (int)((5L << 19) / (panel_bytes > 0 ? panel_bytes : 1));
    GM = GM < 2 ? 2 : (GM > 16 ? 16 : GM);
    const int per_group = GM * nbn;
    const int grp = bid / per_group, rem = bid - grp * per_group;
    const int gm = (nbm - grp * GM) < GM ? (nbm - grp * GM) : GM;
    bn = rem / gm;
    bm = grp * GM + (rem - bn * gm);
  }
  const int m0 = bm * BM, n0 = bn * BN;
  const int nk = p.K / 128;                  // K % 128 == 0 (dispatch condition)

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const char* srcp[NDMA];
  {
    const int r8 = lane >> 3;
    const long k_off = (long)(((lane & 7) ^ r8) * 16);                // pre-swizzled source chunk (tile row & 7 == r8)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int ra_ = m0 + (i * NW + wid) * 8 + r8; ra_ = ra_ < p.M ? ra_ : p.M - 1;
      srcp[i] = p.A + (long)ra_ * p.lda_b + k_off;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int rb_ = n0 + (i * NW + wid) * 8 + r8; rb_ = rb_ < p.N ? rb_ : p.N - 1;
      srcp[NA + i] = p.W + (long)rb_ * p.ldw_b + k_off;
    }
  }
  auto issue = [&](int st) {          // the NEXT K tile (the pointers advance)
    char* base = smem + st * kStage;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)srcp[i], (lptr_t)(base + (i * NW + wid) * 1024), 16, 0, 0);
      srcp[i] += 128;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)srcp[NA + i], (lptr_t)(base + BM * kRowStride + (i * NW + wid) * 1024), 16, 0, 0);
      srcp[NA + i] += 128;
    }
  };

  const int frow = lane & 15, sw = frow & 7, lgc = lane >> 4;
  // scale dwords of this lane: W row blocks j (scale of the MFMA's first operand), A row blocks i (second operand); 32-bit
  // offsets from the two (scalar) base pointers, advancing one 16-byte group per four K steps
  uint32_t sco[NSC];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    int r = n0 + wc * WN + j * 16 + frow; r = r < p.N ? r : p.N - 1;
    sco[j] = (uint32_t)(r * p.ks_b + lgc * 4);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int r = m0 + wr * WM + i * 16 + frow; r = r < p.M ? r : p.M - 1;
    sco[NI + i] = (uint32_t)(r * p.ks_b + lgc * 4);
  }
  const uint8_t* const wsb = p.w_scale;
  const uint8_t* const asb = p.a_scale;
  int scur[NSC], snxt[NSC];
  auto load_scales = [&](int (&dst)[NSC]) {   // the NEXT group of four K steps (the offsets advance); asm: counted by hand
#pragma unroll
    for (int q = 0; q < NSC; ++q) {
      if (q < NI) asm volatile("global_load_dword %0, %1, %2" : "=v"(dst[q]) : "v"(sco[q]), "s"(wsb) : "memory");
      else asm volatile("global_load_dword %0, %1, %2" : "=v"(dst[q]) : "v"(sco[q]), "s"(asb) : "memory");
      sco[q] += 16;
    }
  };

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Every fragment read, every wait and every MFMA of the K loop is a VOLATILE asm statement: hipcc keeps volatile asm
  // statements in program order, so an MFMA can never be placed above the wait its operands need (the builtin form could:
  // a pure register operation, and the asm-issued loads look complete to hipcc the moment they are issued).  What remains
  // for the compiler is register assignment: the two 16-byte halves of a fragment are joined into the MFMA's 8-register
  // operand by a shufflevector, which must coalesce to NOTHING (the ds_reads write straight into the halves of the tuple) --
  // a copy could be scheduled above the wait.  Checked in the ISA of this build (the K loop holds no move of a fragment
  // register, only the scale dwords move); the guards that run on every build are tools/check_kernel_resources.py (no scratch,
  // no spills) and the EXACT product tests of tests/test_gpu_mx8.py over every K-tail case -- a stale fragment is a wrong integer.
  i32x4_t alo_[MI], ahi_[MI], blo_[2][NI], bhi_[2][NI];
#define MX_DEP4(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
  auto wait_scales = [&](int (&v)[NSC]) {      // (no instruction: the vmcnt wait that covers them stands next to it)
    if constexpr (NSC == 12) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]));
    else if constexpr (NSC == 8) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    else asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
    static_assert(NSC == 12 || NSC == 8 || NSC == 6, "scale register lists");
  };
  // vmcnt immediates: what may stay in flight behind the stage that must have landed
#define MX_STR2(x) #x
#define MX_STR(x) MX_STR2(x)
  auto wait_vm = [&](auto n_c) {               // s_waitcnt vmcnt(N), N in {0, NDMA, NDMA + NSC}
    constexpr int N_ = decltype(n_c)::value;
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N_ == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N_ == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N_ == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N_ == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N_ < 0, "vmcnt immediate not listed");
  };
  load_scales(scur);
  issue(0);
  if (nk > 1) {
    issue(1);
    wait_vm(std::integral_constant<int, NDMA>{});      // scales and stage 0 of this wave have landed, stage 1 may still fly
  } else {
    wait_vm(std::integral_constant<int, 0>{});
  }
  __builtin_amdgcn_sched_barrier(0);
  wait_scales(scur);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const uint32_t a_off = (uint32_t)((wr * WM + frow) * kRowStride), b_off = (uint32_t)((BM + wc * WN + frow) * kRowStride);
  const uint32_t c_lo = (uint32_t)((lgc ^ sw) * 16), c_hi = (uint32_t)(((4 + lgc) ^ sw) * 16);
  // row block q of a 16-row-block column sits q * 2048 bytes further (16 rows x 128 bytes)
#define MX_RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
  auto ld_a_half = [&](int st, auto half_c) {            // A row blocks half * MI/2 .. of stage st
    constexpr int H0 = decltype(half_c)::value * MH_;
    auto& alo = alo_; auto& ahi = ahi_;        // (named in a plain expression: a generic lambda does not capture a variable that only an asm operand uses)
    const uint32_t pl = lds0 + st * kStage + a_off + c_lo, ph = lds0 + st * kStage + a_off + c_hi;
#pragma unroll
    for (int q = 0; q < MH_; ++q) {
      if constexpr (MH_ >= 1) { if (q == 0) { MX_RD(alo[H0 + 0], pl, (H0 + 0) * 2048); MX_RD(ahi[H0 + 0], ph, (H0 + 0) * 2048); } }
      if constexpr (MH_ >= 2) { if (q == 1) { MX_RD(alo[H0 + 1], pl, (H0 + 1) * 2048); MX_RD(ahi[H0 + 1], ph, (H0 + 1) * 2048); } }
      if constexpr (MH_ >= 4) { if (q == 2) { MX_RD(alo[H0 + 2], pl, (H0 + 2) * 2048); MX_RD(ahi[H0 + 2], ph, (H0 + 2) * 2048); } }
      if constexpr (MH_ >= 4) { if (q == 3) { MX_RD(alo[H0 + 3], pl, (H0 + 3) * 2048); MX_RD(ahi[H0 + 3], ph, (H0 + 3) * 2048); } }
    }
  };
  auto ld_b = [&](int st, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    auto& blo = blo_; auto& bhi = bhi_;
    const uint32_t pl = lds0 + st * kStage + b_off + c_lo, ph = lds0 + st * kStage + b_off + c_hi;
    MX_RD(blo[S][0], pl, 0); MX_RD(bhi[S][0], ph, 0);
    MX_RD(blo[S][1], pl, 2048); MX_RD(bhi[S][1], ph, 2048);
    MX_RD(blo[S][2], pl, 4096); MX_RD(bhi[S][2], ph, 4096);
    MX_RD(blo[S][3], pl, 6144); MX_RD(bhi[S][3], ph, 6144);
  };
#undef MX_RD
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  ld_a_half(0, I0{});
  ld_b(0, I0{});
  int cur = 0, nxt2 = 2;

  // One K step; SET = kt & 1 (compile time): the W register set the step multiplies (it reads the next step's W blocks into the
  // other one).  Everything else that varies along K is a block-uniform scalar branch around asm statements: whether a step
  // kt + 2 exists (DMA), whether a scale group is requested (every fourth step), whether a step follows at all.  The step's
  // scale byte is byte 0 of the scale dwords: they are shifted down one byte per step (MI + 4 VALU ops beside 4 MI MFMAs), so
  // OP_SEL stays 0 and ONE step body serves every step -- the loop is rolled (two steps per iteration for the two W sets).
  // An earlier form unrolled four steps with OP_SEL 0..3 and a five-way tail: 19 inlined step bodies whose register
  // assignments hipcc joined with copies of not-yet-landed fragments.
#pragma unroll
  for (int q = 0; q < NSC; ++q) snxt[q] = 0;
  auto step = [&](auto set_c, int kt) {
    constexpr int SET = decltype(set_c)::value;
    auto& alo = alo_; auto& ahi = ahi_; auto& blo = blo_; auto& bhi = bhi_;
    const bool more = (kt & 3) == 0 && kt + 4 < nk;
    const bool dma = kt + 2 < nk;
    const bool last = kt + 1 >= nk;
    if (more) load_scales(snxt);
    if (dma) issue(nxt2);
    ld_a_half(cur, I1{});
    // A blocks 0 .. MI/2 - 1 and this step's W blocks (read one phase ago) are in; the MI reads just issued may still fly
    if constexpr (MI == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else if constexpr (MI == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
    static_assert(MI == 8 || MI == 4 || MI == 2, "2 * (MI / 2) reads were just issued");
    __builtin_amdgcn_sched_barrier(0);
    auto& acc_r = acc; auto& scur_r = scur;      // (plain uses: see ld_a_half)
    auto mma = [&](auto half_c) {
      constexpr int H0 = decltype(half_c)::value * MH_;
      auto& acc = acc_r; auto& scur = scur_r;
#pragma unroll
      for (int i = H0; i < H0 + MH_; ++i) {
        const i32x8_t af = __builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const i32x8_t wf = __builtin_shufflevector(blo[SET][j], bhi[SET][j], 0, 1, 2, 3, 4, 5, 6, 7);
          // inline asm with the accumulator tied in place: the builtin let hipcc rotate accumulators through fresh registers
          // (D != C), which doubled their footprint and spilled asm-loaded fragments.  Accumulators: AGPRs (`+a`) for the
          // one-wave-per-SIMD geometry (512 registers per wave: 256 + 256); VGPRs for the eight-wave form (a kernel that
          // touches AGPRs gets its 256 registers split 128 + 128 by hipcc).
          if constexpr (NW == 4) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(acc[j][i]) : "v"(wf), "v"(af), "v"(scur[j]), "v"(scur[NI + i]));
          else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc[j][i]) : "v"(wf), "v"(af), "v"(scur[j]), "v"(scur[NI + i]));
        }
      }
    };
    mma(I0{});
    __builtin_amdgcn_sched_barrier(0);
    // stage kt + 1 of this wave has landed -- what may stay in flight is this step's DMA and (group starts) the scale dwords;
    // A blocks MI/2 .. are in and every read of stage kt by this wave has retired
    if (!last) {
      if (more) wait_vm(std::integral_constant<int, NDMA + NSC>{});
      else if (dma) wait_vm(std::integral_constant<int, NDMA>{});
      else wait_vm(std::integral_constant<int, 0>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!last) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      cur = cur == NST - 1 ? 0 : cur + 1;
      nxt2 = nxt2 == NST - 1 ? 0 : nxt2 + 1;
      ld_a_half(cur, I0{});
      ld_b(cur, std::integral_constant<int, SET ^ 1>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    mma(I1{});
    __builtin_amdgcn_sched_barrier(0);
    // the next step's scale byte into byte 0 (a new group every fourth step: requested three steps ago, landed behind the
    // vmcnt wait of the step after that)
    if ((kt & 3) == 3) {
#pragma unroll
      for (int q = 0; q < NSC; ++q) scur[q] = snxt[q];
    } else {
#pragma unroll
      for (int q = 0; q < NSC; ++q) scur[q] = (int)((unsigned)scur[q] >> 8);
    }
  };
  int kt = 0;
  for (; kt + 2 <= nk; kt += 2) {
    step(I0{}, kt);
    step(I1{}, kt + 1);
  }
  if (kt < nk) step(I0{}, kt);
#undef MX_DEP4
#undef MX_STR
#undef MX_STR2
  // the MFMAs are asm: hipcc does not know that the accumulators come out of the matrix pipe (8 passes behind the last
  // issue) and pads nothing in front of their first VALU / accvgpr read; the sched_barriers keep every such read below the
  // padding (without them a read may be scheduled in between the final MFMAs: see gemm_glds4_kernel)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  g3_epilogue<EPI, MI, MXO>(p, acc, m0, n0, wr, wc, lane);
}

template <int EPI, int MI, int NW, bool MXO = false>
int launch_mx8(const GemmP& p, hipStream_t s) {
  constexpr int BM = 16 * MI * (NW / 2);
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + 127) / 128;
  hipLaunchKernelGGL((gemm_mx8_kernel<EPI, MI, NW, MXO>), dim3(nbm * nbn), dim3(NW * 64), 3 * (BM + 128) * 128, s, p);
  return check_launch("gemm_mx8_kernel");
}
template <int EPI>
bool prepare_mx8() {
  bool ok = true;
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mx8_kernel<EPI, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (128 + 128) * 128) == hipSuccess;
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mx8_kernel<EPI, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 + 128) * 128) == hipSuccess;
  if constexpr (EPI == MH_EPI_GEGLU || EPI == MH_EPI_BIAS_GELU) {   // ... and their forms that write the result as an MX-fp8 operand
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mx8_kernel<EPI, 2, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (128 + 128) * 128) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mx8_kernel<EPI, 4, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 + 128) * 128) == hipSuccess;
  }
  return ok;
}
// tile-count thresholds that were run-time options until round 5 (measured: batched DiT-S bf16 171.6 ms at 192, 167.4 at 96, 168.6 at 48)
constexpr long kMx8Tile256Min = 192, kGemmTile256Min = 96;
template <int EPI>
int dispatch_mx8(const GemmP& p, hipStream_t s) {
  // fewer 256-row tiles than option mx8_tile256_min: the 128-row form doubles the workgroups
  const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
  if constexpr (EPI == MH_EPI_GEGLU || EPI == MH_EPI_BIAS_GELU) {
    if (p.mxq) {   // the result leaves as the next GEMM's MX-fp8 operand
      if (tiles256 < kMx8Tile256Min) return launch_mx8<EPI, 2, 8, true>(p, s);
      return launch_mx8<EPI, 4, 8, true>(p, s);
    }
  }
  if (tiles256 < kMx8Tile256Min) return launch_mx8<EPI, 2, 8>(p, s);
  return launch_mx8<EPI, 4, 8>(p, s);     // 256 x 128 tile as eight waves of 64 x 64 (two per SIMD); the four-wave 128 x 64 geometry measured 0.99-1.37 x the bf16 kernel against 1.07-1.46 x (profiles/r04_mx8_gemm_bench.txt) and was removed in round 5
}
int dispatch_mx8_epi(const GemmP& p, int epi, hipStream_t s) {
  switch (epi) {
    case MH_EPI_STORE: return dispatch_mx8<MH_EPI_STORE>(p, s);
    case MH_EPI_STORE_F32: return dispatch_mx8<MH_EPI_STORE_F32>(p, s);
    case MH_EPI_RESID: return dispatch_mx8<MH_EPI_RESID>(p, s);
    case MH_EPI_GEGLU: return dispatch_mx8<MH_EPI_GEGLU>(p, s);
    case MH_EPI_BIAS_GELU: return dispatch_mx8<MH_EPI_BIAS_GELU>(p, s);
    case MH_EPI_GATE_RESID: return dispatch_mx8<MH_EPI_GATE_RESID>(p, s);
    case MH_EPI_KV_SCATTER: return dispatch_mx8<MH_EPI_KV_SCATTER>(p, s);
    case MH_EPI_QKV_VT: return dispatch_mx8<MH_EPI_QKV_VT>(p, s);
  }
  set_error("mh_gemm: epilogue %d is not built for MH_MX8 operands", epi);
  return MH_ERR_ARG;
}

template <int EPI, int MI>
int launch_s3g(const GemmP& p, hipStream_t s) {
  constexpr int BM = 64 * MI;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + 127) / 128;
  const int tiles = nbm * nbn;
  hipLaunchKernelGGL((gemm_s3g_kernel<EPI, MI>), dim3(tiles < 256 ? tiles : 256), dim3(512), 3 * (BM + 128) * 128, s, p);
  return check_launch("gemm_s3g_kernel");
}

template <int EPI, int MI>
int launch_glds3(const GemmP& p, hipStream_t s) {
  constexpr int BM = 64 * MI;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + 127) / 128;
  hipLaunchKernelGGL((gemm_glds3_kernel<EPI, MI>), dim3(nbm * nbn), dim3(512), 3 * (BM + 128) * 128, s, p);
  return check_launch("gemm_glds3_kernel");
}

template <int EPI>
int launch_glds2s(const GemmP& p, hipStream_t s) {
  const int nbm = (p.M + 127) / 128, nbn = (p.N + 127) / 128;
  hipLaunchKernelGGL((gemm_glds2s_kernel<EPI>), dim3(nbm * nbn), dim3(512), 2 * (128 + 128) * 128, s, p);
  return check_launch("gemm_glds2s_kernel");
}

template <int EPI>
int launch_glds4(const GemmP& p, hipStream_t s) {
  const int nbm = (p.M + 255) / 256, nbn = (p.N + 255) / 256;
  hipLaunchKernelGGL((gemm_glds4_kernel<EPI>), dim3(nbm * nbn), dim3(512), 2 * (256 + 256) * 128, s, p);
  return check_launch("gemm_glds4_kernel");
}

template <typename T, int BM, int BN, int EPI, bool S3 = false, bool GL = false>
int launch_gemm(const GemmP& p, hipStream_t s) {
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  size_t smem = 2 * (size_t)(BM + BN) * (RowBytes<BM>::v + 16) + BM * 8 + 2048;   // + LayerNorm statistics
  hipLaunchKernelGGL((gemm_tn_kernel<T, BM, BN, EPI, S3, GL>), dim3(nbm * nbn), dim3(256), smem, s, p);
  return check_launch("gemm_tn_kernel");
}

// > 64 KiB of dynamic LDS needs an explicit opt-in per kernel; done once for every instantiation,
// outside any stream capture (gemm_prepare()).
template <typename T, int BM, int BN, int EPI>
bool prepare_one() {
  const size_t smem = 2 * (size_t)(BM + BN) * (RowBytes<BM>::v + 16) + BM * 8 + 2048;   // + LayerNorm statistics
  bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<T, BM, BN, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess;
  if constexpr (sizeof(T) == 2 && BM == 128 && BN == 128)
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<T, BM, BN, EPI, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess;
  return ok;
}
template <typename T, int EPI>
bool prepare_epi() {
  bool ok = prepare_one<T, 128, 128, EPI>() && prepare_one<T, 64, 64, EPI>();
  if constexpr (sizeof(T) == 2)
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds3_kernel<EPI, 4>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 + 128) * 128) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds3_kernel<EPI, 2>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (128 + 128) * 128) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds4_kernel<EPI>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds2s_kernel<EPI>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 128) * 128) == hipSuccess;
  if constexpr (EPI != MH_EPI_GEGLU) ok = ok && prepare_one<T, 32, 32, EPI>() && prepare_one<T, 16, 16, EPI>();   // GEGLU pairs two 16-col blocks per wave
  return ok;
}
template <int EPI>
bool prepare_s3g() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_s3g_kernel<EPI, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             3 * (128 + 128) * 128) == hipSuccess;
}

template <typename T>
bool prepare_type() {
  return prepare_epi<T, MH_EPI_STORE>() && prepare_epi<T, MH_EPI_STORE_F32>() && prepare_epi<T, MH_EPI_RESID>() &&
         prepare_epi<T, MH_EPI_GEGLU>() && prepare_epi<T, MH_EPI_BIAS_GELU>() && prepare_epi<T, MH_EPI_GATE_RESID>() &&
         prepare_epi<T, MH_EPI_KV_SCATTER>() && prepare_epi<T, MH_EPI_QKV_VT>() && prepare_epi<T, MH_EPI_QKV_CACHE>() &&
         prepare_epi<T, MH_EPI_BIAS_GELU_ERF>();
}

// below this many 32x32 tiles the 16x16 split-K tile is used (option gemm_splitk_tiles, 0 = never)
long splitk_threshold() { return option(OPT_GEMM_SPLITK_TILES); }

template <typename T, int EPI>
int dispatch_tile(const GemmP& p, hipStream_t s) {
  if constexpr (std::is_same<T, float>::value && (EPI == MH_EPI_STORE_F32 || EPI == MH_EPI_QKV_VT ||
                                                  EPI == MH_EPI_GATE_RESID || EPI == MH_EPI_BIAS_GELU)) {
    // A pre-split as well: the three-stage LDS-DMA form.  128-row tiles only: the 256-row form needs two fragment sets of
    // 128 VGPRs beside 64 accumulators and hipcc spills fragments -- registers written by the inline-asm LDS reads, which it
    // believes valid and stores to scratch BEFORE the data has arrived (wrong results, found by the DiT-B 1024-point golden).
    if (p.split3 & 2) return launch_s3g<EPI, 2>(p, s);
    if (p.split3) return launch_gemm<T, 64, 64, EPI, true>(p, s);
  }
  // tile by grid size: the chip has 256 CUs; a K step of a wave costs MI*NI MFMAs, so small problems want
  // many small tiles (DiT: M = 256 rows) and big ones the 128x128 tile (encoder: M = 40k rows)
  const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const long tiles64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
  // exact-fp32 MFMA is 16x slower than bf16: an fp32 GEMM is compute-bound long before the big tile pays (measured on
  // the batched DiT, M = 8192: 64x64 tiles 667 ms vs 128x128 827 ms per 100 steps) -> 8x the bf16 threshold
  const long min128 = option(OPT_GEMM_TILE128_MIN) * (std::is_same<T, float>::value ? 8 : 1);
  if (tiles128 >= min128) {
    if constexpr (sizeof(T) == 2) {   // plain bf16 operands: the LDS-DMA forms (option gemm_glds = 0: register staging)
      const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
      const bool vec_ok = p.N % 4 == 0 && p.ldc % 4 == 0 && (p.gate == nullptr || p.gate_ld % 4 == 0) &&
                          (EPI != MH_EPI_GEGLU || p.N % 8 == 0) && p.K % 64 == 0;
      if (option(OPT_GEMM_GLDS) >= 3 && p.K <= option(OPT_GEMM_2STAGE_MAX_K) && !p.stats_out && vec_ok)
        return launch_glds2s<EPI>(p, s);      // short K: two workgroups per CU, one's prologue / epilogue under the other's K loop
      if (option(OPT_GEMM_GLDS) >= 2 && tiles256 >= kGemmTile256Min && !p.stats_out && vec_ok) {
        // fewer 256-row tiles than half the CUs: the 128-row form of the same kernel doubles the workgroups (batched DiT-S bf16,
        // N = 384: 96 -> 192 workgroups, 153 -> 137 ms per 100 steps; at 192 tiles -- DiT-B, N = 768 -- it loses, 294 -> 308)
        if (tiles256 < 128 && option(OPT_GEMM_GLDS) >= 3) return launch_glds3<EPI, 2>(p, s);
        // enough 256 x 256 tiles to fill the chip: the two-stage kernel with 128 x 64 wave tiles
        // ... when its last round of workgroups is not mostly idle: one workgroup per CU, so 628 tiles (osuT5-large's N = 1024
        // projections) are three rounds at 82 % -- measured 276 us against 244 for the 256 x 128 tile, while 471 tiles (92 %) gain
        const long tiles256sq = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        const long min256sq = option(OPT_GEMM_TILE256SQ_MIN);
        const long rounds = (tiles256sq + 255) / 256;
        const bool full_rounds = min256sq == 1 || tiles256sq * 100 >= rounds * 256 * 88;     // (option value 1: always, for tests)
        if (min256sq > 0 && tiles256sq >= min256sq && full_rounds && option(OPT_GEMM_GLDS) >= 3) return launch_glds4<EPI>(p, s);
        return launch_glds3<EPI, 4>(p, s);
      }
      if (option(OPT_GEMM_GLDS) != 0) return launch_gemm<T, 128, 128, EPI, false, true>(p, s);
    }
    return launch_gemm<T, 128, 128, EPI>(p, s);
  }
  if constexpr (EPI == MH_EPI_GEGLU) {
    return launch_gemm<T, 64, 64, EPI>(p, s);
  } else {
    if (tiles64 >= 192) return launch_gemm<T, 64, 64, EPI>(p, s);
    const long tiles32 = (long)((p.M + 31) / 32) * ((p.N + 31) / 32);
    if (p.ascending_k || tiles32 >= splitk_threshold()) return launch_gemm<T, 32, 32, EPI>(p, s);
    return launch_gemm<T, 16, 16, EPI>(p, s);
  }
}

template <typename T>
int dispatch_epi(const GemmP& p, int epi, hipStream_t s) {
  switch (epi) {
    case MH_EPI_STORE: return dispatch_tile<T, MH_EPI_STORE>(p, s);
    case MH_EPI_STORE_F32: return dispatch_tile<T, MH_EPI_STORE_F32>(p, s);
    case MH_EPI_RESID: return dispatch_tile<T, MH_EPI_RESID>(p, s);
    case MH_EPI_GEGLU: return dispatch_tile<T, MH_EPI_GEGLU>(p, s);
    case MH_EPI_BIAS_GELU: return dispatch_tile<T, MH_EPI_BIAS_GELU>(p, s);
    case MH_EPI_GATE_RESID: return dispatch_tile<T, MH_EPI_GATE_RESID>(p, s);
    case MH_EPI_KV_SCATTER: return dispatch_tile<T, MH_EPI_KV_SCATTER>(p, s);
    case MH_EPI_QKV_VT: return dispatch_tile<T, MH_EPI_QKV_VT>(p, s);
    case MH_EPI_QKV_CACHE: return dispatch_tile<T, MH_EPI_QKV_CACHE>(p, s);
    case MH_EPI_BIAS_GELU_ERF: return dispatch_tile<T, MH_EPI_BIAS_GELU_ERF>(p, s);
  }
  set_error("mh_gemm: unknown epilogue %d", epi);
  return MH_ERR_ARG;
}

}  // namespace

int gemm_prepare() {
  static bool done = false;
  if (done) return MH_OK;
  if (!(prepare_type<bf16_t>() && prepare_type<float>() && prepare_s3g<MH_EPI_QKV_VT>() && prepare_s3g<MH_EPI_GATE_RESID>() &&
        prepare_s3g<MH_EPI_BIAS_GELU>() && prepare_s3g<MH_EPI_STORE_F32>() && prepare_mx8<MH_EPI_STORE>() &&
        prepare_mx8<MH_EPI_STORE_F32>() && prepare_mx8<MH_EPI_RESID>() && prepare_mx8<MH_EPI_GEGLU>() && prepare_mx8<MH_EPI_BIAS_GELU>() &&
        prepare_mx8<MH_EPI_GATE_RESID>() && prepare_mx8<MH_EPI_KV_SCATTER>() && prepare_mx8<MH_EPI_QKV_VT>())) {
    set_error("gemm_prepare: hipFuncSetAttribute failed: %s", hipGetErrorString(hipGetLastError()));
    return MH_ERR_LAUNCH;
  }
  done = true;
  return MH_OK;
}

int gemm(const MhGemm& g, hipStream_t s, bool ascending_k) {
  MH_REQUIRE(g.A && g.W && (g.C || (g.dtype == MH_MX8 && g.mx_out)), "mh_gemm: null operand");
  MH_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "mh_gemm: bad shape M=%d N=%d K=%d", g.M, g.N, g.K);
  MH_REQUIRE(g.dtype == MH_F32 || g.dtype == MH_BF16 || g.dtype == MH_MX8, "mh_gemm: bad dtype %d", g.dtype);
  if (g.dtype == MH_MX8) {
    // e4m3 elements (1 byte) + E8M0 scales: K steps of 128, vector epilogue; outputs are bf16 / fp32 as the epilogue says
    MH_REQUIRE(g.a_scale && g.w_scale, "mh_gemm: MH_MX8 needs a_scale and w_scale");
    MH_REQUIRE(g.K % 128 == 0 && g.lda % 16 == 0 && g.ldw % 16 == 0 && g.lda >= g.K && g.ldw >= g.K,
               "mh_gemm: MH_MX8 needs K %% 128 == 0 and lda, ldw multiples of 16 (K=%d lda=%d ldw=%d)", g.K, g.lda, g.ldw);
    MH_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0 && ((uintptr_t)g.a_scale % 4) == 0 && ((uintptr_t)g.w_scale % 4) == 0,
               "mh_gemm: MH_MX8 operands must be 16-byte aligned, scales 4-byte aligned");
    MH_REQUIRE(g.N % 4 == 0 && g.ldc % 4 == 0 && (g.gate == nullptr || g.gate_ld % 4 == 0) && (g.epilogue != MH_EPI_GEGLU || g.N % 32 == 0) &&
                   !g.stats_out && !g.ln_stats && !g.w_split3,
               "mh_gemm: MH_MX8 needs N, ldc (gate_ld) multiples of 4 and takes no LayerNorm fusion / split3");
    if (g.epilogue == MH_EPI_GATE_RESID) MH_REQUIRE(g.gate && g.rows_per_batch > 0, "mh_gemm: GATE_RESID needs gate and rows_per_batch");
    if (g.epilogue == MH_EPI_KV_SCATTER)
      MH_REQUIRE(g.kv_B > 0 && g.kv_H > 0 && g.kv_L > 0 && g.M == g.kv_B * g.kv_L && g.N % (g.kv_H * 64) == 0, "mh_gemm: bad KV scatter geometry");
    if (g.epilogue == MH_EPI_QKV_VT)
      MH_REQUIRE(g.C2 && g.kv_H > 0 && g.kv_L > 0 && g.kv_Lpad >= g.kv_L && g.n_split > 0 && g.N - g.n_split == g.kv_H * 64 && g.M % g.kv_L == 0,
                 "mh_gemm: bad QKV_VT geometry");
    { int rc = gemm_prepare(); if (rc != MH_OK) return rc; }
    GemmP q{};
    q.A = (const char*)g.A; q.lda_b = g.lda; q.W = (const char*)g.W; q.ldw_b = g.ldw; q.C = g.C; q.ldc = g.ldc;
    q.M = g.M; q.N = g.N; q.K = g.K; q.bias = g.bias; q.gate = g.gate; q.gate_ld = g.gate_ld; q.rows_per_batch = g.rows_per_batch;
    q.kv_B = g.kv_B; q.kv_H = g.kv_H; q.kv_L = g.kv_L; q.C2 = g.C2; q.n_split = g.n_split; q.kv_Lpad = g.kv_Lpad;
    q.a_scale = g.a_scale; q.w_scale = g.w_scale; q.ks_b = mx8_scale_row_bytes(g.K);
    if (g.mx_out) {
      const int width = g.epilogue == MH_EPI_GEGLU ? g.N / 2 : g.N;
      MH_REQUIRE(g.epilogue == MH_EPI_GEGLU || g.epilogue == MH_EPI_BIAS_GELU, "mh_gemm: mx_out goes with MH_EPI_GEGLU / MH_EPI_BIAS_GELU");
      MH_REQUIRE(g.mx_out_scales && width % 128 == 0 && g.ldc >= width && ((uintptr_t)g.mx_out % 4) == 0,
                 "mh_gemm: mx_out needs mx_out_scales, an output width that is a multiple of 128 and ldc >= width (width=%d ldc=%d)", width, g.ldc);
      q.mxq = g.mx_out; q.mxs = g.mx_out_scales; q.mxs_b = mx8_scale_row_bytes(width);
    }
    return dispatch_mx8_epi(q, g.epilogue, s);
  }
  const int es = g.dtype == MH_BF16 ? 2 : 4;
  const int vec = 16 / es;
  MH_REQUIRE(g.K % vec == 0 && g.lda % vec == 0 && g.ldw % vec == 0,
             "mh_gemm: K=%d lda=%d ldw=%d must be multiples of %d", g.K, g.lda, g.ldw, vec);
  MH_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0, "mh_gemm: operands must be 16-byte aligned");
  MH_REQUIRE(g.lda >= g.K && g.ldw >= g.K, "mh_gemm: leading dimension smaller than K");
  if (g.epilogue == MH_EPI_GEGLU) MH_REQUIRE(g.N % 32 == 0, "mh_gemm: GEGLU needs N %% 32 == 0 (N=%d)", g.N);
  if (g.epilogue == MH_EPI_GATE_RESID)
    MH_REQUIRE(g.gate && g.rows_per_batch > 0, "mh_gemm: GATE_RESID needs gate and rows_per_batch");
  if (g.epilogue == MH_EPI_KV_SCATTER)
    MH_REQUIRE(g.kv_B > 0 && g.kv_H > 0 && g.kv_L > 0 && g.M == g.kv_B * g.kv_L && g.N % (g.kv_H * 64) == 0,
               "mh_gemm: bad KV scatter geometry");
  if (g.epilogue == MH_EPI_QKV_VT)
    MH_REQUIRE(g.C2 && g.kv_H > 0 && g.kv_L > 0 && g.kv_Lpad >= g.kv_L && g.n_split > 0 &&
                   g.N - g.n_split == g.kv_H * 64 && g.M % g.kv_L == 0,
               "mh_gemm: bad QKV_VT geometry");
  if (g.epilogue == MH_EPI_QKV_CACHE)
    MH_REQUIRE(g.C2 && g.C3 && g.C4 && g.kv_H > 0 && g.kv_L > 0 && g.kv_Lpad >= g.kv_L && g.cache_len >= g.kv_L &&
                   g.n_split == g.kv_H * 64 && g.N == 3 * g.kv_H * 64 && g.M % g.kv_L == 0,
               "mh_gemm: bad QKV_CACHE geometry");
  { int rc = gemm_prepare(); if (rc != MH_OK) return rc; }
  GemmP p;
  p.ascending_k = ascending_k;
  p.split3 = g.w_split3;
  p.C3 = g.C3; p.C4 = g.C4; p.cache_len = g.cache_len;
  p.C2 = g.C2; p.n_split = g.n_split; p.kv_Lpad = g.kv_Lpad;
  p.A = (const char*)g.A; p.lda_b = (long)g.lda * es;
  p.W = (const char*)g.W; p.ldw_b = (long)g.ldw * es;
  p.C = g.C; p.ldc = g.ldc;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.bias = g.bias; p.gate = g.gate; p.gate_ld = g.gate_ld; p.rows_per_batch = g.rows_per_batch;
  p.kv_B = g.kv_B; p.kv_H = g.kv_H; p.kv_L = g.kv_L;
  p.stats_out = g.stats_out; p.ln_stats = g.ln_stats; p.ln_strips = g.ln_strips; p.ln_shift = g.ln_shift;
  p.ln_scale = g.ln_scale; p.ln_ld = g.ln_ld; p.ln_eps = g.ln_eps;
  if (g.stats_out)
    MH_REQUIRE((g.epilogue == MH_EPI_STORE_F32 || g.epilogue == MH_EPI_GATE_RESID) && g.N % 16 == 0,
               "mh_gemm: stats_out needs a fp32-output epilogue (STORE_F32 / GATE_RESID) and N %% 16 == 0");
  if (g.w_split3 & 2)
    MH_REQUIRE((g.w_split3 & 1) && g.lda % 32 == 0 && g.N % 4 == 0 && g.ldc % 4 == 0 && !g.stats_out && !g.ln_stats &&
                   (g.gate == nullptr || g.gate_ld % 4 == 0) && ((g.w_split3 & 4) == 0 || (g.epilogue == MH_EPI_BIAS_GELU && g.ldc % 32 == 0)),
               "mh_gemm: w_split3 & 2 (A pre-split) needs w_split3 & 1, lda %% 32 == 0, N and ldc multiples of 4, no LayerNorm fusion");
  if ((g.w_split3 & 2) == 0) MH_REQUIRE((g.w_split3 & 4) == 0, "mh_gemm: w_split3 & 4 (pre-split output) only with w_split3 & 2");
  if ((g.w_split3 & 2) && g.epilogue == MH_EPI_BIAS_GELU) MH_REQUIRE(g.w_split3 & 4, "mh_gemm: the A-pre-split BIAS_GELU form writes its output pre-split (w_split3 & 4)");
  if (g.w_split3)
    MH_REQUIRE(g.dtype == MH_F32 && g.K % 32 == 0 && g.ldw % 32 == 0 &&
                   (g.epilogue == MH_EPI_STORE_F32 || g.epilogue == MH_EPI_QKV_VT || g.epilogue == MH_EPI_GATE_RESID ||
                    g.epilogue == MH_EPI_BIAS_GELU),
               "mh_gemm: w_split3 is an fp32 path (K, ldw multiples of 32; STORE_F32 / QKV_VT / GATE_RESID / BIAS_GELU)");
  if (g.ln_stats)
    MH_REQUIRE(g.dtype == MH_F32 && g.ln_shift && g.ln_scale && g.ln_strips > 0 && g.rows_per_batch > 0 && g.ln_ld >= g.K,
               "mh_gemm: the fused LayerNorm-modulate prologue is fp32 only and needs shift / scale / strips / rows_per_batch");
  // LayerNorm fused into the bf16 x 3 kernel's A load was an option of the batched DiT until round 5 (measured slower than the
  // stand-alone pass: 303.7 vs 292.3 ms) with a history: an early build returned non-repeatable rows on grids of > 1000
  // workgroups, the cause was never isolated in the ISA, and a later unrelated edit of this file brought the failure back.  A path
  // nothing uses and nobody can vouch for is refused, not shipped.
  MH_REQUIRE(!(g.ln_stats && g.w_split3), "mh_gemm: the fused LayerNorm-modulate prologue is not available with w_split3 (run mh_ln_modulate first)");
  if (g.dtype == MH_BF16) return dispatch_epi<bf16_t>(p, g.epilogue, s);
  return dispatch_epi<float>(p, g.epilogue, s);
}

}  // namespace mh

extern "C" int mh_gemm(const MhGemm* g, void* stream) {
  if (!g) { mh::set_error("mh_gemm: null descriptor"); return MH_ERR_ARG; }
  return mh::gemm(*g, (hipStream_t)stream, false);
}
