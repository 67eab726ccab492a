// Fused (flash-style) multi-head attention for head_dim = 64 on the gfx950 matrix cores.
//   T5 encoder self-attention : softmax(Q K^T + rel_bias[h][k - q]) V, scale 1 (T5 folds 1/sqrt(d) into init)
//                               HF T5Attention.forward; restated custom_transformers/t5.py:170-250
//   T5 decoder prompt prefill : causal self-attention over the (left-padded) prompt with the unidirectional
//                               bias table and the key-padding mask; cross-attention of all prompt positions
//                               over the encoder keys (no bias) -- the batched form of what the per-token
//                               decode kernels do one position at a time
//   DiT self-attention        : softmax(Q K^T / 8 + band mask) V (nn.MultiheadAttention with the banded
//                               bool mask of diffusion_pipeline.py:146-148; models.py:111-116,145-151)
// Layout: block = 4 waves, 64 query rows (16 per wave) of one (batch, head); K tile [64 keys][64] and the
// pre-transposed V tile [64 d][64 keys] staged in LDS with 16-byte accesses; S and O live in MFMA
// accumulators, softmax reductions are 16-lane shuffles, P goes through a wave-private LDS patch to
// become the A operand of the PV product.  Q, K, V^T, the output and the masks are addressed through
// explicit (row, batch, head) strides so that the same kernel reads the packed QKV GEMM output, the
// [B][H][L][64] K/V caches of the decoder and their transposed copies.
#include <type_traits>

#include "internal.hpp"

namespace mh {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void flash_attn_kernel(AttnArgs p) {
  constexpr int ES = (int)sizeof(T);
  constexpr int KM = Atom<T>::KM, KCH = Atom<T>::KCH;
  constexpr int KS = 64 / KM;               // MFMA k-steps over a 64-wide contraction
  constexpr int RS = 64 * ES + 16;          // padded LDS row stride (bytes)
  constexpr int CPR = 64 * ES / 16;         // 16-byte chunks per tile row
  constexpr int CPT = 64 * CPR / 256;       // chunks per thread per tile

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vts = smem + 64 * RS;
  char* Ps = smem + 128 * RS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int Lq = p.Lq, Lk = p.Lk;
  const int qb0 = qt * 64;
  const int q0w = qb0 + wid * 16;
  const int l15 = lane & 15, lg = lane >> 4;

  // Q fragments straight from global memory (A operand: row = l15, k chunk = lg)
  typename Atom<T>::frag_t qf[KS];
  {
    int qr = q0w + l15;
    qr = qr < Lq ? qr : Lq - 1;
    const char* qp = (const char*)p.q + (long)b * p.q_bs + (long)qr * p.q_rs + (long)(h * 64 + lg * KCH) * ES;
    if constexpr (sizeof(T) == 4) {
      // fp32 (round 5): k slot lg of MFMA step (k16, i) <-> contraction index k16*16 + lg*4 + i -- for BOTH operands of a product, so
      // the sum is the same set of terms in another order, and every fragment of four steps is ONE 16-byte access (global here,
      // ds_read_b128 below) instead of four 4-byte ones: 36 LDS reads per 128 MFMAs of a key tile instead of 144
      const char* qv = (const char*)p.q + (long)b * p.q_bs + (long)qr * p.q_rs + (long)(h * 64 + lg * 4) * ES;
#pragma unroll
      for (int k16 = 0; k16 < 4; ++k16) {
        const float4 t = *reinterpret_cast<const float4*>(qv + k16 * 64);
        reinterpret_cast<float*>(qf)[k16 * 4 + 0] = t.x; reinterpret_cast<float*>(qf)[k16 * 4 + 1] = t.y;
        reinterpret_cast<float*>(qf)[k16 * 4 + 2] = t.z; reinterpret_cast<float*>(qf)[k16 * 4 + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[ks] = Atom<T>::load(reinterpret_cast<const T*>(qp + ks * KM * ES));
    }
  }

  f32x4_t o[4];
  float m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -1e30f; l[r] = 0.f; }

  // key tiles this query block can see
  int t_lo = 0, t_hi = (Lk - 1) / 64;
  // band > 0: -(band - 1) <= k - q <= band (the DiT's banded mask); band < 0: |k - q| <= -band (the symmetric window of
  // the Whisper family's local layers: flash-attention `window_size = (w, w)`)
  const int rel_lo = p.band > 0 ? -(p.band - 1) : (p.band < 0 ? p.band : -(1 << 30));
  const int rel_hi = p.band > 0 ? p.band : (p.band < 0 ? -p.band : (1 << 30));
  if (p.band != 0) {
    int klo = qb0 + rel_lo;
    klo = klo < 0 ? 0 : klo;
    int khi = qb0 + 63 + rel_hi;
    khi = khi > Lk - 1 ? Lk - 1 : khi;
    t_lo = klo / 64;
    t_hi = khi / 64;
  }
  if (p.causal) {
    int khi = p.q_pos0 + qb0 + 63;
    khi = khi > Lk - 1 ? Lk - 1 : khi;
    t_hi = khi / 64;
  }
  // open_from > 0 (the DiT pipeline's pad_sequence, diffusion_pipeline.py:186-193): positions >= open_from are padding that
  // the reference leaves attendable -- key k is visible from query q iff in-band OR k >= open_from OR q >= open_from
  const int open_from = (p.band != 0 && p.open_from > 0) ? p.open_from : (1 << 30);
  int t_last = t_hi, t_open = 1 << 30;
  if (open_from < Lk) {
    t_last = (Lk - 1) / 64;
    if (qb0 + 63 >= open_from) { t_lo = 0; t_hi = t_last; }     // a pad query in the block: every key tile
    else t_open = open_from / 64;                                 // band tiles, then the tiles holding pad keys
  }

  const char* kbase = (const char*)p.k + (long)b * p.k_bs + (long)h * p.k_hs;
  const char* vbase = (const char*)p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
  const float* bh = p.bias ? p.bias + (long)h * p.bias_hs + p.bias_center : nullptr;
  const uint8_t* mrow = p.key_mask ? p.key_mask + (long)b * p.mask_ld : nullptr;
  char* Pw = Ps + wid * 16 * RS;

  for (int kt = t_lo; kt <= t_last; ++kt) {
    if (kt > t_hi && kt < t_open) continue;      // (block-uniform) between the band and the pad columns
    const int kv0 = kt * 64;
    // ---- stage K tile [key][d] and V^T tile [d][key] ----
    // (requesting the next tile into registers while this one is multiplied was measured and lost: the encoder's
    // attention 9.8 -> ~12 ms per batch; so was a 128-query workgroup -- two 16-query fragments per wave, half the LDS
    // reads and barriers per multiply, but 180 + 32 registers: mel + encoder 22.4 -> 24.1 ms per batch.  The co-resident
    // workgroups already overlap each other's latencies; what this kernel lacks is MFMA work per wave instruction.)
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPR, ch = idx % CPR;
      int kr = kv0 + row;
      kr = kr < Lk ? kr : Lk - 1;
      uint4 kv = *reinterpret_cast<const uint4*>(kbase + (long)kr * p.k_rs + ch * 16);
      uint4 vv = *reinterpret_cast<const uint4*>(vbase + ((long)row * p.Lkpad + kv0) * ES + ch * 16);
      *reinterpret_cast<uint4*>(Ks + row * RS + ch * 16) = kv;
      *reinterpret_cast<uint4*>(Vts + row * RS + ch * 16) = vv;
    }
    __syncthreads();

    // ---- S = Q K^T (16 queries x 64 keys per wave) ----
    f32x4_t s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 kf = *reinterpret_cast<const float4*>(Ks + (j * 16 + l15) * RS + (k16 * 16 + lg * 4) * 4);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(reinterpret_cast<const float*>(qf)[k16 * 4 + 0], kf.x, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(reinterpret_cast<const float*>(qf)[k16 * 4 + 1], kf.y, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(reinterpret_cast<const float*>(qf)[k16 * 4 + 2], kf.z, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(reinterpret_cast<const float*>(qf)[k16 * 4 + 3], kf.w, s[j], 0, 0, 0);
        }
    } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typename Atom<T>::frag_t kf =
            Atom<T>::load(reinterpret_cast<const T*>(Ks + (j * 16 + l15) * RS + (ks * KM + lg * KCH) * ES));
        s[j] = Atom<T>::mma(qf[ks], kf, s[j]);
      }
    }
    }

    // ---- scale, bias, mask, online softmax ----
    float mx[4], rs[4], alpha[4];
    // bias / key-mask values: wave-uniform branches, unconditional loads from clamped indices (a predicated
    // load per element would serialise into one L2 round trip each)
    float bv[4][4];
    int kmv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kmv[j] = 1;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[r][j] = 0.f;
    if (bh) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qg = q0w + lg * 4 + r;
        const int qpos = p.q_pos0 + (qg < Lq ? qg : Lq - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kg = kv0 + j * 16 + l15;
          int idx = p.bias_sign * ((kg < Lk ? kg : Lk - 1) - qpos);
          idx = idx < p.bias_min ? p.bias_min : (idx > p.bias_max ? p.bias_max : idx);
          bv[r][j] = bh[idx];
        }
      }
    }
    if (mrow) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kg = kv0 + j * 16 + l15;
        const int mval = mrow[kg < p.mask_len ? kg : p.mask_len - 1];
        kmv[j] = (kg < p.mask_len) ? mval : 1;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qg = q0w + lg * 4 + r;
      const int qpos = p.q_pos0 + (qg < Lq ? qg : Lq - 1);
      float mxr = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kg = kv0 + j * 16 + l15;
        float v = s[j][r] * p.scale + bv[r][j];
        const int rel = kg - qpos;
        bool ok = (kg < Lk) && (kmv[j] != 0);
        if (p.band != 0) ok = ok && (((rel >= rel_lo) && (rel <= rel_hi)) || kg >= open_from || qpos >= open_from);
        if (p.causal) ok = ok && (rel <= 0);
        v = ok ? v : -INFINITY;
        s[j][r] = v;
        mxr = fmaxf(mxr, v);
      }
#pragma unroll
      for (int o2 = 8; o2 > 0; o2 >>= 1) mxr = fmaxf(mxr, __shfl_xor(mxr, o2, 64));
      mx[r] = mxr;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mn = fmaxf(m[r], mx[r]);
      alpha[r] = fexp<T>(m[r] - mn);
      m[r] = mn;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = fexp<T>(s[j][r] - mn);
        s[j][r] = pv;
        sum += pv;
      }
#pragma unroll
      for (int o2 = 8; o2 > 0; o2 >>= 1) sum += __shfl_xor(sum, o2, 64);
      rs[r] = sum;
      l[r] = l[r] * alpha[r] + rs[r];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j][r] *= alpha[r];
    }
    // ---- P (C layout) -> wave-private LDS patch (A layout source) ----
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<T*>(Pw + (lg * 4 + r) * RS + (j * 16 + l15) * ES) = Elem<T>::from_f32(s[j][r]);
    // (no workgroup barrier: the patch is private to this wave and the LDS operations of one wave execute in order; the
    // wave-scope fence only stops the COMPILER from moving the cross-lane reads below above these stores -- no instruction)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- O += P V ----
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int k16 = 0; k16 < 4; ++k16) {
        const float4 pf = *reinterpret_cast<const float4*>(Pw + l15 * RS + (k16 * 16 + lg * 4) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 vf = *reinterpret_cast<const float4*>(Vts + (j * 16 + l15) * RS + (k16 * 16 + lg * 4) * 4);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.x, vf.x, o[j], 0, 0, 0);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.y, vf.y, o[j], 0, 0, 0);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.z, vf.z, o[j], 0, 0, 0);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.w, vf.w, o[j], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      typename Atom<T>::frag_t pf =
          Atom<T>::load(reinterpret_cast<const T*>(Pw + l15 * RS + (ks * KM + lg * KCH) * ES));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typename Atom<T>::frag_t vf =
            Atom<T>::load(reinterpret_cast<const T*>(Vts + (j * 16 + l15) * RS + (ks * KM + lg * KCH) * ES));
        o[j] = Atom<T>::mma(pf, vf, o[j]);
      }
    }
    }
    __syncthreads();
  }

  // ---- normalise and store (a fully masked query row -- a left-pad position -- yields zeros) ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qg = q0w + lg * 4 + r;
    if (qg >= Lq) continue;
    const float inv = l[r] > 0.f ? 1.0f / l[r] : 0.f;
    if constexpr (sizeof(T) == 4) {
      if (p.out_split3) {   // the out-projection's A operand, pre-split for gemm_s3g_kernel: [32 x bf16 hi | 32 x bf16 lo] per 32 values
        char* orow = (char*)p.out + (long)b * p.out_bs + (long)qg * p.out_rs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = h * 64 + j * 16 + l15;
          const float v = o[j][r] * inv;
          const bf16_t hi = f32_to_bf16(v);
          const bf16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
          bf16_t* blk = reinterpret_cast<bf16_t*>(orow + (long)(c >> 5) * 128);
          blk[c & 31] = hi;
          blk[32 + (c & 31)] = lo;
        }
        continue;
      }
    }
    T* op = reinterpret_cast<T*>((char*)p.out + (long)b * p.out_bs + (long)qg * p.out_rs) + h * 64 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) op[j * 16] = Elem<T>::from_f32(o[j][r] * inv);
  }
}

// ---- flash2: bf16, 128 queries per workgroup (32 per wave), S computed TRANSPOSED ----------------------------------------
// The kernel above multiplies 16 queries per wave: every 16 MFMAs read 16 K / V fragments from LDS and push P through a
// wave-private LDS patch (2 KB written with 2-byte stores, 2 KB read) -- LDS and VALU, not the matrix cores, bound it
// (MfmaUtil 7-9 % on the encoder).  Here
//   S^T = K Q^T   (K fragment = the MFMA's A operand, Q fragment = B): lane (l15, lg) holds, for QUERY l15 of a 16-query
//                 block, the scores of KEYS kb*16 + lg*4 + r -- 16 keys per lane and query block: the row maximum / sum are
//                 in-lane reductions plus 2 shuffles (xor 16, 32) instead of 4, and the sum is only reduced once at the end;
//   O^T = V^T P^T (V^T fragment = A, P = B): those very registers, packed to bf16, ARE the B fragment of the PV product --
//                 an MFMA does not care in which order its 32 k-slots enumerate the keys as long as both operands agree,
//                 so slot lg*8 + j stands for key (2s + j/4)*16 + lg*4 + j%4 and the V^T fragment is two 8-byte reads at
//                 those keys.  No P patch, no barrier, and the result lane (l15 = query) owns the same query as the
//                 softmax statistics: the rescale needs no shuffle.
// Two query blocks per wave share every K and V^T fragment: per 64-key tile 32 MFMAs on 16 KB of fragment reads (was 16 on
// 20 KB).  Scores are kept in the log2 domain (one FMA applies scale and bias), a tile that needs no masking skips it.
// Round 5: (i) ONE-dimensional grid with an XCD-aware id -> (query tile, head, batch) map: the dispatcher places workgroup
// id i on XCD i % 8, so XCD x takes the CONSECUTIVE virtual ids [x * per_xcd, (x + 1) * per_xcd) -- all query tiles of a
// (batch, head) pair run on one XCD, next to each other in time, and its L2 fetches that pair's K / V^T once (the 3-D grid
// spread them over all eight: FETCH_SIZE 2.8-5.8 x the 184 MB of Q / K / V per launch, profiles/r04_pmc_hbm_traffic.txt);
// (ii) the staging is split (guide T14): the NEXT tile's global loads are issued while this tile is computed, into 16
// registers, and written to the OTHER LDS buffer -- one barrier per tile instead of two and no exposed global round trip
// (the single-buffered loop waited ~3 us per tile for its own loads with only two workgroups per CU to hide them).  All
// loads of the loop are unconditional with clamped indices (a branch around a load makes hipcc drain vmcnt at the join),
// and a tile's bias loads are issued BEFORE the prefetch of the tile after next: vmcnt counts in order, the bias must not
// wait behind the prefetch.  Same products, same order per tile: bit-identical to the previous form.
// Reductions over the 4 lane groups (lanes l15, l15 + 16, + 32, + 48) on the VALU: gfx950's v_permlane16_swap / v_permlane32_swap
// exchange 16-lane rows / 32-lane halves between two registers, so {x, x} -> {[r0 r0 r2 r2], [r1 r1 r3 r3]} and one max / add
// is the xor-16 butterfly step -- no ds_bpermute round trip through the LDS crossbar and, with it, no `s_waitcnt lgkmcnt(0)` that
// also waits for every fragment read in flight (__shfl_xor compiles to ds_bpermute_b32 here).
__device__ inline float max_over_lane_groups(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ inline float sum_over_lane_groups(float v) {   // (r0 + r1) + (r2 + r3) in every lane: the order of `v += xor 16; v += xor 32`
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// tools/flash2_bench.py builds this kernel with parts switched OFF to price them (bit mask: 1 bias loads, 2 exponentials,
// 4 the global -> LDS staging, 8 the cross-lane reductions, 16 QK^T MFMAs, 32 PV MFMAs); always 0 in the library
#ifndef MH_F2_PROBE
#define MH_F2_PROBE 0
#endif
// SIMPLE: no band, no causal mask, no key mask (the encoders) -- known at compile time, so the general tile form is only the
// `key < Lk` test of the last tile and the clamped bias gather, and the LEAN form can afford the S^T / softmax overlap (a second
// S^T register set; with the band / causal / key-mask arithmetic compiled in, the allocator spills inside the tile loop).
template <int QB, bool BIAS, bool SIMPLE>
__global__ __launch_bounds__(256, QB == 1 ? 3 : 2) void flash2_bf16_kernel(AttnArgs p_, int nqt, int H, int total) {
  AttnArgs p = p_;
  if constexpr (SIMPLE) { p.band = 0; p.causal = 0; p.key_mask = nullptr; p.open_from = 0; }
  using T = bf16_t;
  constexpr int RS = 64 * 2 + 16;           // padded LDS row stride (bytes)
  constexpr int TILE = 64 * RS;             // one K or V^T tile
  constexpr int WQ = QB * 16;               // queries per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K tile | V^T tile]
  const int per_xcd = gridDim.x >> 3;
  const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (vid >= total) return;
  const int qt = vid % nqt, bh_ = vid / nqt;
  const int h = bh_ % H, b = bh_ / H;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int Lq = p.Lq, Lk = p.Lk;
  const int qb0 = qt * 4 * WQ;
  const int q0w = qb0 + wid * WQ;
  const int l15 = lane & 15, lg = lane >> 4;
  constexpr float kLog2e = 1.4426950408889634f;

  // Q fragments (B operand: row = query l15, k chunk = lg) straight from global memory
  bf16x8_t qf[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qr = q0w + qb * 16 + l15;
    qr = qr < Lq ? qr : Lq - 1;
    const char* qp = (const char*)p.q + (long)b * p.q_bs + (long)qr * p.q_rs + (long)(h * 64 + lg * 8) * 2;
    qf[qb][0] = *reinterpret_cast<const bf16x8_t*>(qp);
    qf[qb][1] = *reinterpret_cast<const bf16x8_t*>(qp + 64);
  }
  f32x4_t o[QB][4];                          // O^T: o[qb][db][r] = O[query qb*16 + l15][d = db*16 + lg*4 + r]
  float m[QB], lsum[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m[qb] = -INFINITY; lsum[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < 4; ++db) o[qb][db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // key tiles this query block can see (same rules as flash_attn_kernel)
  int t_lo = 0, t_hi = (Lk - 1) / 64;
  const int rel_lo = p.band > 0 ? -(p.band - 1) : (p.band < 0 ? p.band : -(1 << 30));
  const int rel_hi = p.band > 0 ? p.band : (p.band < 0 ? -p.band : (1 << 30));
  if (p.band != 0) {
    int klo = qb0 + rel_lo;
    klo = klo < 0 ? 0 : klo;
    int khi = qb0 + 4 * WQ - 1 + rel_hi;
    khi = khi > Lk - 1 ? Lk - 1 : khi;
    t_lo = klo / 64;
    t_hi = khi / 64;
  }
  if (p.causal) {
    int khi = p.q_pos0 + qb0 + 4 * WQ - 1;
    khi = khi > Lk - 1 ? Lk - 1 : khi;
    t_hi = khi / 64;
  }
  const int open_from = (p.band != 0 && p.open_from > 0) ? p.open_from : (1 << 30);
  int t_last = t_hi, t_open = 1 << 30;
  if (open_from < Lk) {
    t_last = (Lk - 1) / 64;
    if (qb0 + 4 * WQ - 1 >= open_from) { t_lo = 0; t_hi = t_last; }
    else t_open = open_from / 64;
  }
  // the visited tiles in order: t_lo .. t_hi, then (pad columns of an open band) t_open .. t_last   (block-uniform)
  auto next_tile = [&](int kt) -> int {
    int n = kt + 1;
    if (n > t_hi && n < t_open) n = t_open;
    return n <= t_last ? n : -1;
  };
  int kt = t_lo;
  if (kt > t_hi && kt < t_open) kt = t_open;

  const char* kbase = (const char*)p.k + (long)b * p.k_bs + (long)h * p.k_hs;
  const char* vbase = (const char*)p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
  const float* bh = BIAS ? p.bias + (long)h * p.bias_hs + p.bias_center : nullptr;
  const uint8_t* mrow = p.key_mask ? p.key_mask + (long)b * p.mask_ld : nullptr;
  const float c2 = p.scale * kLog2e;

  // staging: thread -> (row = idx >> 3, 16-byte chunk = idx & 7) of the K tile [key][d] and the V^T tile [d][key], idx = tid, tid + 256
  const int srow = tid >> 3, sch = (tid & 7) * 16;
  const long vrow0 = (long)srow * p.Lkpad * 2 + sch, vrow1 = (long)(srow + 32) * p.Lkpad * 2 + sch;
  const int soff0 = srow * RS + sch, soff1 = (srow + 32) * RS + sch;
  uint4 kr0, kr1, vr0, vr1;
  // K tiles are staged TWO tiles ahead of the tile whose softmax runs, V^T tiles ONE ahead (a step computes S of tile i + 1 and
  // P V of tile i): K ring slot = tile ordinal & 1, V^T ring slot likewise -- smem = [K slot 0 | K slot 1 | V^T slot 0 | V^T slot 1]
#define MH_F2_ISSUE_K(tile)                                                                        \
  do {                                                                                             \
    const int kv0_ = (tile) * 64;                                                                  \
    int ka_ = kv0_ + srow, kb_ = kv0_ + srow + 32;                                                 \
    ka_ = ka_ < Lk ? ka_ : Lk - 1;                                                                 \
    kb_ = kb_ < Lk ? kb_ : Lk - 1;                                                                 \
    kr0 = *reinterpret_cast<const uint4*>(kbase + (long)ka_ * p.k_rs + sch);                       \
    kr1 = *reinterpret_cast<const uint4*>(kbase + (long)kb_ * p.k_rs + sch);                       \
  } while (0)
#define MH_F2_ISSUE_V(tile)                                                                        \
  do {                                                                                             \
    const int kv0_ = (tile) * 64;                                                                  \
    vr0 = *reinterpret_cast<const uint4*>(vbase + vrow0 + (long)kv0_ * 2);                         \
    vr1 = *reinterpret_cast<const uint4*>(vbase + vrow1 + (long)kv0_ * 2);                         \
  } while (0)
#define MH_F2_WRITE_K(slot)                                                                        \
  do {                                                                                             \
    char* kd_ = smem + (slot) * TILE;                                                              \
    *reinterpret_cast<uint4*>(kd_ + soff0) = kr0;                                                  \
    *reinterpret_cast<uint4*>(kd_ + soff1) = kr1;                                                  \
  } while (0)
#define MH_F2_WRITE_V(slot)                                                                        \
  do {                                                                                             \
    char* vd_ = smem + (2 + (slot)) * TILE;                                                        \
    *reinterpret_cast<uint4*>(vd_ + soff0) = vr0;                                                  \
    *reinterpret_cast<uint4*>(vd_ + soff1) = vr1;                                                  \
  } while (0)
  // S^T = K Q^T of one tile: dst[kb][qb][r] = score(key kb*16 + lg*4 + r of the tile, query q0w + qb*16 + l15)
#define MH_F2_QK(dst, Ks_)                                                                         \
  do {                                                                                             \
    _Pragma("unroll") for (int kb = 0; kb < 4; ++kb)                                               \
      _Pragma("unroll") for (int qb = 0; qb < QB; ++qb) dst[kb][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                               \
      _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) {                                           \
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>((Ks_) + (kb * 16 + l15) * RS + (ks * 32 + lg * 8) * 2); \
        _Pragma("unroll") for (int qb = 0; qb < QB; ++qb) {                                        \
          if (MH_F2_PROBE & 16) dst[kb][qb][0] += __builtin_bit_cast(f32x4_t, kf)[qb] * 1e-30f;    \
          else dst[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], dst[kb][qb], 0, 0, 0); \
        }                                                                                          \
      }                                                                                            \
  } while (0)

  if (kt <= t_last) {                       // (block-uniform; an empty range leaves zeros)
    // tiles t0 (= kt), t1, t2, t3 of the visited sequence; a missing one is replaced by a valid tile whose data is never used
    int t1 = next_tile(kt);
    int t2 = t1 >= 0 ? next_tile(t1) : -1;
    {   // K / V^T of the first tile and K of the second in ONE round trip (short sequences -- the DiT's 128 points -- are all prologue)
      MH_F2_ISSUE_K(t1 >= 0 ? t1 : kt);
      const uint4 k1a = kr0, k1b = kr1;
      MH_F2_ISSUE_K(kt); MH_F2_ISSUE_V(kt);
      MH_F2_WRITE_K(0); MH_F2_WRITE_V(0);
      kr0 = k1a; kr1 = k1b;
      MH_F2_WRITE_K(1);
    }
    MH_F2_ISSUE_K(t2 >= 0 ? t2 : kt); MH_F2_ISSUE_V(t1 >= 0 ? t1 : kt);      // in flight across the barrier: written by step 0
    __syncthreads();
    f32x4_t stA[4][QB], stB[4][QB];
    MH_F2_QK(stA, smem);
    __syncthreads();                        // step 0 overwrites K slot 0: every wave has read it
    int ord = 0;                            // ordinal of tile kt in the visited sequence
    const int qlo = p.q_pos0 + q0w, qhi = qlo + WQ - 1;                         // positions of this wave's queries
    // One step = softmax of tile kt (its S^T arrives in `st`, computed by the previous step) + S^T of tile t1 into `sn` + P V of
    // tile kt.  The MFMAs of the next tile's S^T do not depend on this tile's softmax: both sit in one basic block so that the
    // matrix pipe works in the shadow of the VALU stream (the kernel is VALU-issue-bound: tools/flash2_bench.py probe builds --
    // with MFMAs compiled out it took 250 of 372 us, i.e. the un-pipelined form ran its MFMAs beside nothing).
    // Two compiled forms chosen per wave and tile (wave-uniform: a scalar branch).  LEAN: no mask of any kind and (with a bias)
    // the whole wave on the vector-load path -- straight-line code; the general form evaluates masks and bias clamps per element.
    auto step = [&](auto lean_c, f32x4_t (&st)[4][QB], f32x4_t (&sn)[4][QB], int t3) {
      constexpr bool LEAN = decltype(lean_c)::value;
      const char* Kn = smem + ((ord + 1) & 1) * TILE;             // K of tile t1
      const char* Vts = smem + (2 + (ord & 1)) * TILE;            // V^T of tile kt
      const int kv0 = kt * 64;
      bool need_mask = false;
      if constexpr (!LEAN) {
        need_mask = (kv0 + 63 >= Lk) || mrow != nullptr;
        if (p.band != 0) need_mask = need_mask || (kv0 - qhi < rel_lo) || (kv0 + 63 - qlo > rel_hi);
        if (p.causal) need_mask = need_mask || (kv0 + 63 > qlo);
      }
      // Relative bias, vector path (bias_sign = +1, tile away from the table's ends): rel = key - query of element (kb, qb, r) is
      // rel0 + 16 (kb - qb) + r with rel0 = kv0 + lg*4 - (q_pos0 + q0w + l15): the 4 r values are CONSECUTIVE table entries --
      // 4 + QB - 1 unaligned 16-byte loads per lane instead of 16 QB scalar gathers with their index arithmetic.  Issued BEFORE
      // the staging loads below: vmcnt counts in order, the bias must not wait behind the prefetch.
      const int rel0 = kv0 + lg * 4 - (qlo + l15);
      float4 bq[BIAS && LEAN ? 4 + QB - 1 : 1];
      if constexpr (BIAS && LEAN) {
#pragma unroll
        for (int g = 0; g < 4 + QB - 1; ++g) {
          if (MH_F2_PROBE & 1) bq[g] = make_float4(c2, c2, c2, c2);
          else bq[g] = *reinterpret_cast<const float4*>(bh + rel0 + 16 * (g - (QB - 1)));   // group g <-> kb - qb = g - (QB - 1)
        }
      }
      // ---- staged registers (K of t2, V^T of t1: requested one step ago) into their ring slots; K of t3 / V^T of t2 requested ----
      if (!(MH_F2_PROBE & 4)) {
        MH_F2_WRITE_K(ord & 1);
        MH_F2_WRITE_V((ord + 1) & 1);
        MH_F2_ISSUE_K(t3 >= 0 ? t3 : kt);
        MH_F2_ISSUE_V(t2 >= 0 ? t2 : kt);
      }
      // ---- S^T of the next tile (UNCONDITIONAL: in the last step it multiplies a stale K slot into a register set nobody reads -- a
      // skip would keep the old contents of `sn` alive through the whole step: 32 registers, and the loop spills).  LEAN form of a SIMPLE problem: here, in
      // front of the softmax it overlaps with; otherwise behind P V (the per-element mask arithmetic needs the registers) ----
      if constexpr (LEAN && SIMPLE) MH_F2_QK(sn, Kn);

      // ---- log2-domain scores: s2 = s * scale * log2(e) + bias * log2(e); masks only where this tile needs them ----
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        if constexpr (LEAN) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            if constexpr (BIAS) {
              const float4 t = bq[kb - qb + QB - 1];
              st[kb][qb][0] = st[kb][qb][0] * c2 + t.x * kLog2e; st[kb][qb][1] = st[kb][qb][1] * c2 + t.y * kLog2e;
              st[kb][qb][2] = st[kb][qb][2] * c2 + t.z * kLog2e; st[kb][qb][3] = st[kb][qb][3] * c2 + t.w * kLog2e;
            } else {
              st[kb][qb][0] *= c2; st[kb][qb][1] *= c2; st[kb][qb][2] *= c2; st[kb][qb][3] *= c2;
            }
          }
        } else {
          // (lane coordinates through an opaque copy: hipcc would otherwise hoist every per-element index of this rarely taken
          // form out of the tile loop and keep them alive across the lean form, which then spills)
          int l15x = l15, lgx = lg;
          asm volatile("" : "+v"(l15x), "+v"(lgx));
          const int qg = q0w + qb * 16 + l15x;
          const int qpos = p.q_pos0 + (qg < Lq ? qg : Lq - 1);
          if constexpr (BIAS) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int kg = kv0 + kb * 16 + lgx * 4 + r;
                int idx = p.bias_sign * ((kg < Lk ? kg : Lk - 1) - qpos);
                idx = idx < p.bias_min ? p.bias_min : (idx > p.bias_max ? p.bias_max : idx);
                st[kb][qb][r] = st[kb][qb][r] * c2 + bh[idx] * kLog2e;
                if (r == 3) __builtin_amdgcn_sched_barrier(0);   // 4 gathers in flight, not 32: this rarely taken form must not set the kernel's register count
              }
          } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) st[kb][qb][r] *= c2;
          }
          if (need_mask) {                                                          // wave-uniform
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int kg = kv0 + kb * 16 + lgx * 4 + r;
                const int rel = kg - qpos;
                bool ok = kg < Lk;
                if (mrow) ok = ok && (kg >= p.mask_len || mrow[kg < p.mask_len ? kg : p.mask_len - 1] != 0);
                if (p.band != 0) ok = ok && (((rel >= rel_lo) && (rel <= rel_hi)) || kg >= open_from || qpos >= open_from);
                if (p.causal) ok = ok && (rel <= 0);
                st[kb][qb][r] = ok ? st[kb][qb][r] : -INFINITY;
                if (r == 3) __builtin_amdgcn_sched_barrier(0);
              }
          }
        }
        // ---- online softmax of query (qb, l15): 16 keys in this lane, the other 48 in lanes l15 + 16 / 32 / 48 ----
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kb][qb][r]);
        if (!(MH_F2_PROBE & 8)) mx = max_over_lane_groups(mx);
        const float mn = fmaxf(m[qb], mx);
        const float mu = mn == -INFINITY ? 0.f : mn;          // a fully masked row so far: keep exp2(-inf - 0) = 0, not NaN
        const float alpha = __builtin_amdgcn_exp2f(m[qb] - mu);     // raw v_exp_f32 (exp2f() adds a denormal path: 11 more instructions each)
        m[qb] = mn;
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = (MH_F2_PROBE & 2) ? (st[kb][qb][r] - mu) : __builtin_amdgcn_exp2f(st[kb][qb][r] - mu);
            st[kb][qb][r] = pv;
            sum += pv;
          }
        lsum[qb] = lsum[qb] * alpha + sum;                    // this lane's keys only: reduced over lg once, at the end
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          o[qb][db][0] *= alpha; o[qb][db][1] *= alpha; o[qb][db][2] *= alpha; o[qb][db][3] *= alpha;
        }
      }

      // the compiler would emit the 16 S^T MFMAs of the next tile as one block in front of the softmax: spread them through it,
      // one MFMA per ~14 VALU instructions (16 cycles of matrix pipe under ~56+ cycles of VALU issue)
      if constexpr (LEAN && SIMPLE) {
#pragma unroll
        for (int i = 0; i < 8 * QB; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);   // VALU
        }
      }
      // ---- O^T += V^T P^T: k-slot lg*8 + j of step s <-> key (2s + j/4)*16 + lg*4 + j%4 (both operands) ----
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8_t pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const uint32_t w0 = pack_bf16x2(st[2 * s2][qb][0], st[2 * s2][qb][1]), w1 = pack_bf16x2(st[2 * s2][qb][2], st[2 * s2][qb][3]);
          const uint32_t w2 = pack_bf16x2(st[2 * s2 + 1][qb][0], st[2 * s2 + 1][qb][1]), w3 = pack_bf16x2(st[2 * s2 + 1][qb][2], st[2 * s2 + 1][qb][3]);
          const uint4 pk = make_uint4(w0, w1, w2, w3);
          pf[qb] = __builtin_bit_cast(bf16x8_t, pk);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const char* vr = Vts + (db * 16 + l15) * RS + (2 * s2 * 16 + lg * 4) * 2;
          const uint2 v0 = *reinterpret_cast<const uint2*>(vr);
          const uint2 v1 = *reinterpret_cast<const uint2*>(vr + 32);
          const uint4 vk = make_uint4(v0.x, v0.y, v1.x, v1.y);
          const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vk);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            if (MH_F2_PROBE & 32) o[qb][db][0] += __builtin_bit_cast(f32x4_t, vf)[qb] * __builtin_bit_cast(f32x4_t, pf[qb])[db];
            else o[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb], o[qb][db], 0, 0, 0);
          }
        }
      }
      if constexpr (!(LEAN && SIMPLE)) MH_F2_QK(sn, Kn);   // (a skip in the last step would keep `sn` alive: 178 instead of 164 registers = two workgroups per CU instead of three, and the DiT's 768-workgroup launches need three: 14.1 vs 16.5 us)
    };
    // LEAN iff no element of the tile is masked for any query of the wave and every lane's bias reads stay inside the table
    auto is_lean = [&]() -> bool {
      const int kv0 = kt * 64;
      bool lean = (kv0 + 63 < Lk) && mrow == nullptr;
      if (p.band != 0) lean = lean && !((kv0 - qhi < rel_lo) || (kv0 + 63 - qlo > rel_hi));
      if (p.causal) lean = lean && !(kv0 + 63 > qlo);
      if (BIAS) lean = lean && p.bias_sign == 1 && (kv0 - (qlo + 15) - 16 * (QB - 1) >= p.bias_min) && (kv0 + 12 - qlo + 51 <= p.bias_max);
      return __builtin_amdgcn_readfirstlane((int)lean) != 0;
    };
    // two steps per trip with the roles of the two S^T register sets swapped (a runtime-indexed set would live in scratch)
    while (true) {
      int t3 = t2 >= 0 ? next_tile(t2) : -1;
      if constexpr (SIMPLE) {
        if (is_lean()) step(std::true_type{}, stA, stB, t3);
        else step(std::false_type{}, stA, stB, t3);
      } else {
        step(std::false_type{}, stA, stB, t3);     // band / causal / key-mask problems are short (the DiT's windows, a prompt): ONE compiled
      }                                            // form -- their time is instruction fetch and prologue, not the tile arithmetic
      if (t1 < 0) break;
      __syncthreads();
      kt = t1; t1 = t2; t2 = t3; ++ord;
      t3 = t2 >= 0 ? next_tile(t2) : -1;
      if constexpr (SIMPLE) {
        if (is_lean()) step(std::true_type{}, stB, stA, t3);
        else step(std::false_type{}, stB, stA, t3);
      } else {
        step(std::false_type{}, stB, stA, t3);
      }
      if (t1 < 0) break;
      __syncthreads();
      kt = t1; t1 = t2; t2 = t3; ++ord;
    }
  }
#undef MH_F2_ISSUE_K
#undef MH_F2_ISSUE_V
#undef MH_F2_WRITE_K
#undef MH_F2_WRITE_V
#undef MH_F2_QK

  // ---- normalise and store: 4 consecutive d per lane (a fully masked query row -- a left-pad position -- yields zeros) ----
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l = sum_over_lane_groups(lsum[qb]);
    const int qg = q0w + qb * 16 + l15;
    const float inv = l > 0.f ? 1.0f / l : 0.f;         // (no early exit for qg >= Lq: every lane takes part in the lane swaps below)
    // 16 bytes per lane: one v_permlane16_swap per dword on the registers of two d blocks leaves lanes lg = 0 / 2 with d 0-7 / 8-15
    // of block 2 pr and lanes 1 / 3 with those of block 2 pr + 1 -- 64 contiguous bytes per query row and store instruction instead
    // of 32 (the memory system takes 16-byte-per-lane stores at 1.6-1.7 x the rate: tools/micro/store_pattern_probe.hip)
    if constexpr (!SIMPLE) {      // (the band / mask forms sit at 168 registers = three workgroups per CU: 8-byte stores, no extra registers)
      if (qg < Lq) {
        T* op8 = reinterpret_cast<T*>((char*)p.out + (long)b * p.out_bs + (long)qg * p.out_rs) + h * 64 + lg * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db)
          *reinterpret_cast<uint2*>(op8 + db * 16) = make_uint2(pack_bf16x2(o[qb][db][0] * inv, o[qb][db][1] * inv),
                                                                pack_bf16x2(o[qb][db][2] * inv, o[qb][db][3] * inv));
      }
      continue;
    }
    uint2 w[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
      w[db] = make_uint2(pack_bf16x2(o[qb][db][0] * inv, o[qb][db][1] * inv), pack_bf16x2(o[qb][db][2] * inv, o[qb][db][3] * inv));
    T* op = reinterpret_cast<T*>((char*)p.out + (long)b * p.out_bs + (long)(qg < Lq ? qg : Lq - 1) * p.out_rs) + h * 64;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const auto s0 = __builtin_amdgcn_permlane16_swap(w[2 * pr].x, w[2 * pr + 1].x, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(w[2 * pr].y, w[2 * pr + 1].y, false, false);
      if (qg < Lq) *reinterpret_cast<uint4*>(op + (2 * pr + (lg & 1)) * 16 + (lg >> 1) * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
  }
}

// V [B][H][Lk][64] (row stride 64) -> V^T [B][H][64][Lkpad]; pad columns are left untouched (zeroed once by the caller)
template <typename T>
__global__ __launch_bounds__(256) void transpose_v_kernel(const T* __restrict__ v, long v_bs, long v_hs, int Lk, T* __restrict__ vt,
                                                         int Lkpad, int H) {
  __shared__ T tile[64][66];
  const int k0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const T* src = v + b * v_bs + h * v_hs;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int kr = i >> 6, d = i & 63;
    tile[kr][d] = (k0 + kr < Lk) ? src[(long)(k0 + kr) * 64 + d] : T(0);
  }
  __syncthreads();
  T* dst = vt + ((long)(b * H + h) * 64) * Lkpad;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int d = i >> 6, kr = i & 63;
    if (k0 + kr < Lk) dst[(long)d * Lkpad + k0 + kr] = tile[kr][d];
  }
}

}  // namespace

int attention_general(const AttnArgs& a, int B, int H, int dtype, hipStream_t s) {
  MH_REQUIRE(a.q && a.k && a.vt && a.out, "attention: null operand");
  MH_REQUIRE(B > 0 && a.Lq > 0 && a.Lk > 0 && H > 0, "attention: bad shape");
  MH_REQUIRE(a.Lkpad % 64 == 0 && a.Lkpad >= a.Lk, "attention: Lkpad=%d must be a multiple of 64 and >= Lk=%d", a.Lkpad, a.Lk);
  MH_REQUIRE(a.q_rs % 16 == 0 && a.k_rs % 16 == 0 && a.q_bs % 16 == 0 && a.k_bs % 16 == 0 && a.k_hs % 16 == 0 &&
                 a.vt_bs % 16 == 0 && a.vt_hs % 16 == 0,
             "attention: strides must be multiples of 16 bytes");
  const int es = dtype == MH_BF16 ? 2 : 4;
  if (dtype == MH_BF16 && a.out_rs % 16 == 0 && a.out_bs % 16 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {
    // bf16: transposed-S kernel, 128 queries per workgroup (the 64-query kernel below only for outputs it cannot store 8 bytes at a time)
    const bool simple = a.band == 0 && !a.causal && a.key_mask == nullptr;
#ifndef MH_F2_QB_SIMPLE
#define MH_F2_QB_SIMPLE 2      // query blocks (of 16) per wave of the encoder form (A/B builds: 1 = 64 queries per workgroup, three workgroups per CU)
#endif
    constexpr int QBS = MH_F2_QB_SIMPLE;
    const int nqt = ceil_div(a.Lq, simple ? 64 * QBS : 128);
    const long total = (long)nqt * H * B;
    MH_REQUIRE(total < (1L << 30), "attention: too many workgroups");
    const int per_xcd = (int)((total + 7) / 8);
    const size_t smem = (size_t)4 * 64 * (64 * 2 + 16);        // two buffers of (K tile | V^T tile)
    if (a.bias && simple)
      hipLaunchKernelGGL((flash2_bf16_kernel<QBS, true, true>), dim3(8 * per_xcd), dim3(256), smem, s, a, nqt, H, (int)total);
    else if (a.bias)
      hipLaunchKernelGGL((flash2_bf16_kernel<2, true, false>), dim3(8 * per_xcd), dim3(256), smem, s, a, nqt, H, (int)total);
    else if (simple)
      hipLaunchKernelGGL((flash2_bf16_kernel<QBS, false, true>), dim3(8 * per_xcd), dim3(256), smem, s, a, nqt, H, (int)total);
    else
      hipLaunchKernelGGL((flash2_bf16_kernel<2, false, false>), dim3(8 * per_xcd), dim3(256), smem, s, a, nqt, H, (int)total);
    return check_launch("flash2_bf16_kernel");
  }
  dim3 grid(ceil_div(a.Lq, 64), H, B), block(256);
  const size_t smem = (size_t)(128 + 64) * (64 * es + 16);
  if (dtype == MH_BF16)
    hipLaunchKernelGGL(flash_attn_kernel<bf16_t>, grid, block, smem, s, a);
  else
    hipLaunchKernelGGL(flash_attn_kernel<float>, grid, block, smem, s, a);
  return check_launch("flash_attn_kernel");
}

namespace {

// Short-sequence fp32 attention (the DiT at T <= 64 KBW keys): the flash kernel above gives a 128-point window 24
// workgroups whose waves each walk every key tile; here a workgroup owns 16 query rows of one (batch, head) and its
// 4 waves split the KEYS (KBW 16-key blocks each, operands straight from L2 into MFMA fragments, no LDS staging),
// then merge their (max, sum, O) through LDS in wave order.  8x the workgroups, a quarter of the per-wave MFMA chain.
struct SmallAttnP {
  const float* q; const float* k; const float* vt; float* out;
  long ld_qk, vt_hs, vt_bs; int ld_out;
  int L, Lpad, H, band, open_from, out_split3;
  float scale;
};

template <int KBW>
__global__ __launch_bounds__(256) void attn_small_f32_kernel(SmallAttnP p) {
  constexpr int PW = KBW * 16 + 4;                 // P patch row stride (floats)
  __shared__ __attribute__((aligned(16))) float Ps[4][16][PW];
  __shared__ float m_s[4][16], l_s[4][16];
  __shared__ __attribute__((aligned(16))) float O_s[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
  const int L = p.L;
  const int kbase = wid * KBW * 16;                // this wave's first key
  // Round 5: 16-byte loads.  An MFMA does not care in which order its k slots enumerate the contraction index as long as both
  // operands agree, so lane group lg takes the 16 CONSECUTIVE head dims lg*16 .. +15 of its q / k row (4 float4 instead of 16 dwords
  // at a stride of 4) and, for P V, the KBW*4 CONSECUTIVE keys kbase + lg*KBW*4 .. of its V^T row (KBW float4 instead of KBW*4
  // dwords): 5 KBW + 4 load instructions per lane instead of 20 KBW + 16 -- the kernel is a latency kernel whose time is the
  // CU's load path (every strided dword instruction touches 16 lines for 256 bytes).
  const float* qrow = p.q + ((long)b * L + (q0 + l15 < L ? q0 + l15 : L - 1)) * p.ld_qk + h * 64 + lg * 16;
  // operands: every load is independent of every other -- all requested before the first MFMA
  float4 qf[4], kf[KBW][4], vf[4][KBW];
#pragma unroll
  for (int j = 0; j < 4; ++j) qf[j] = *reinterpret_cast<const float4*>(qrow + j * 4);
#pragma unroll
  for (int kb = 0; kb < KBW; ++kb) {
    const int key = kbase + kb * 16 + l15;
    const float* krow = p.k + ((long)b * L + (key < L ? key : L - 1)) * p.ld_qk + h * 64 + lg * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) kf[kb][j] = *reinterpret_cast<const float4*>(krow + j * 4);
  }
  const float* vtb = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int j = 0; j < KBW; ++j) {    // keys kbase + lg*KBW*4 + j*4 .. +3, clamped into the padded row (Lpad % 64 == 0): columns >= L meet P = 0
      int k4 = kbase + lg * KBW * 4 + j * 4;
      k4 = k4 <= p.Lpad - 4 ? k4 : p.Lpad - 4;
      vf[db][j] = *reinterpret_cast<const float4*>(vtb + (long)(db * 16 + l15) * p.Lpad + k4);
    }
  // S = scale * Q K^T with the band mask; C layout: col = l15 (key in block), row = lg*4 + r (query)
  const int rel_lo = p.band > 0 ? -(p.band - 1) : (p.band < 0 ? p.band : -(1 << 30));          // band 0 = open, < 0 = |k - q| <= -band
  const int rel_hi = p.band > 0 ? p.band : (p.band < 0 ? -p.band : (1 << 30));
  const int open_from = (p.band != 0 && p.open_from > 0) ? p.open_from : (1 << 30);   // pad_sequence: see flash_attn_kernel
  f32x4_t sc[KBW];
  float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int kb = 0; kb < KBW; ++kb) {
    f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j].x, kf[kb][j].x, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j].y, kf[kb][j].y, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j].z, kf[kb][j].z, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j].w, kf[kb][j].w, a, 0, 0, 0);
    }
    const int key = kbase + kb * 16 + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + lg * 4 + r;
      const int rel = key - qq;
      const bool ok = (key < L) && (((rel >= rel_lo) && (rel <= rel_hi)) || key >= open_from || qq >= open_from);
      a[r] = ok ? a[r] * p.scale : -INFINITY;
      m[r] = fmaxf(m[r], a[r]);
    }
    sc[kb] = a;
  }
  float l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m[r] = fmaxf(m[r], __shfl_xor(m[r], o, 64));
    float part = 0.f;
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) {
      const float pv = (m[r] == -INFINITY) ? 0.f : expf(sc[kb][r] - m[r]);
      sc[kb][r] = pv;
      part += pv;
    }
    l[r] = group_sum<16>(part);
  }
  // P: C layout -> A layout through the wave's LDS patch
#pragma unroll
  for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) Ps[wid][lg * 4 + r][kb * 16 + l15] = sc[kb][r];
  __syncthreads();
  f32x4_t o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < KBW; ++j) {      // k slot lg of step (j, i) <-> key lg*KBW*4 + j*4 + i of this wave's range, for P and V alike
    const float4 pf = *reinterpret_cast<const float4*>(&Ps[wid][l15][lg * KBW * 4 + j * 4]);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.x, vf[db][j].x, o[db], 0, 0, 0);
      o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.y, vf[db][j].y, o[db], 0, 0, 0);
      o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.z, vf[db][j].z, o[db], 0, 0, 0);
      o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.w, vf[db][j].w, o[db], 0, 0, 0);
    }
  }
  if (l15 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_s[wid][lg * 4 + r] = m[r]; l_s[wid][lg * 4 + r] = l[r]; }
  }
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 4; ++r) O_s[wid][lg * 4 + r][db * 16 + l15] = o[db][r];
  __syncthreads();
  // merge the 4 key ranges in wave order: thread -> (row = tid / 16, 4 consecutive dims)
  const int row = tid >> 4, d0 = (tid & 15) * 4;
  float M = fmaxf(fmaxf(m_s[0][row], m_s[1][row]), fmaxf(m_s[2][row], m_s[3][row]));
  float Ls = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float f = (m_s[w][row] == -INFINITY) ? 0.f : expf(m_s[w][row] - M);
    Ls += l_s[w][row] * f;
    const float4 ov = *reinterpret_cast<const float4*>(&O_s[w][row][d0]);
    acc.x += ov.x * f; acc.y += ov.y * f; acc.z += ov.z * f; acc.w += ov.w * f;
  }
  if (q0 + row < L) {
    const float inv = Ls > 0.f ? 1.f / Ls : 0.f;
    float* orow = p.out + ((long)b * L + q0 + row) * p.ld_out;
    if (p.out_split3) {   // pre-split for gemm_s3g_kernel (see flash_attn_kernel)
      const int c = h * 64 + d0;
      const float v0 = acc.x * inv, v1 = acc.y * inv, v2 = acc.z * inv, v3 = acc.w * inv;
      const uint32_t h01 = pack_bf16x2(v0, v1), h23 = pack_bf16x2(v2, v3);
      const float r0 = v0 - __uint_as_float(h01 << 16), r1 = v1 - __uint_as_float(h01 & 0xffff0000u);
      const float r2 = v2 - __uint_as_float(h23 << 16), r3 = v3 - __uint_as_float(h23 & 0xffff0000u);
      char* dst = reinterpret_cast<char*>(orow) + (long)(c >> 5) * 128 + (c & 31) * 2;
      *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
      *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));
    } else {
      *reinterpret_cast<float4*>(orow + h * 64 + d0) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
  }
}

// The key-split kernel is a LATENCY design (one CFG pair: 96 workgroups of 16 queries instead of 24 of 64); with many
// chunks in one denoiser batch its per-lane operand loads (80 dword loads per lane, every workgroup re-reading its
// head's K / V from L2) lose to the LDS-staged flash kernel: option attn_small_max_wgs (default 1024 workgroups).
constexpr long kAttnSmallMaxWgs = 1024;   // (an option until round 5; measured crossover of the two kernels on the batched DiT)
bool small_attn_ok(int B, int H, int L) { return (long)B * H * ceil_div(L, 16) <= kAttnSmallMaxWgs; }

}  // namespace

int attention(const void* qk, int ld_qk, int k_col0, const void* vt, int Lpad, const float* bias, void* out,
              int ld_out, int B, int L, int H, float scale, int band, int dtype, hipStream_t s, int open_from, int out_split3) {
  MH_REQUIRE(!out_split3 || (dtype == MH_F32 && ld_out % 32 == 0), "attention: out_split3 is an fp32 path (ld_out %% 32 == 0)");
  MH_REQUIRE(qk && vt && out, "mh_attention: null operand");
  const int es = dtype == MH_BF16 ? 2 : 4;
  MH_REQUIRE((ld_qk * es) % 16 == 0 && (k_col0 * es) % 16 == 0, "mh_attention: rows must be 16-byte aligned");
  if (dtype == MH_F32 && !bias && L <= 256 && ld_out % 4 == 0 && small_attn_ok(B, H, L)) {
    SmallAttnP sp{};
    sp.q = (const float*)qk; sp.k = (const float*)qk + k_col0; sp.vt = (const float*)vt; sp.out = (float*)out;
    sp.ld_qk = ld_qk; sp.vt_hs = 64L * Lpad; sp.vt_bs = (long)H * 64 * Lpad; sp.ld_out = ld_out;
    sp.L = L; sp.Lpad = Lpad; sp.H = H; sp.band = band; sp.open_from = open_from; sp.out_split3 = out_split3; sp.scale = scale;
    dim3 grid(ceil_div(L, 16), H, B), block(256);
    if (L <= 128) hipLaunchKernelGGL(attn_small_f32_kernel<2>, grid, block, 0, s, sp);
    else hipLaunchKernelGGL(attn_small_f32_kernel<4>, grid, block, 0, s, sp);
    return check_launch("attn_small_f32_kernel");
  }
  AttnArgs a{};
  a.q = qk; a.q_rs = (long)ld_qk * es; a.q_bs = (long)L * ld_qk * es;
  a.k = (const char*)qk + (long)k_col0 * es; a.k_rs = (long)ld_qk * es; a.k_bs = (long)L * ld_qk * es; a.k_hs = 64L * es;
  a.vt = vt; a.Lkpad = Lpad; a.vt_hs = 64L * Lpad * es; a.vt_bs = (long)H * 64 * Lpad * es;
  a.bias = bias; a.bias_hs = 2L * L - 1; a.bias_center = L - 1; a.bias_sign = 1; a.bias_min = -(L - 1); a.bias_max = L - 1;
  a.key_mask = nullptr; a.mask_ld = 0; a.mask_len = 0;
  a.out = out; a.out_rs = (long)ld_out * es; a.out_bs = (long)L * ld_out * es;
  a.Lq = L; a.Lk = L; a.scale = scale; a.band = band; a.open_from = open_from; a.out_split3 = out_split3; a.causal = 0; a.q_pos0 = 0;
  return attention_general(a, B, H, dtype, s);
}

int transpose_v(const void* v, long v_bs_el, long v_hs_el, int Lk, void* vt, int Lkpad, int B, int H, int dtype, hipStream_t s) {
  dim3 grid(ceil_div(Lk, 64), H, B), block(256);
  if (dtype == MH_BF16)
    hipLaunchKernelGGL(transpose_v_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)v, v_bs_el, v_hs_el, Lk, (bf16_t*)vt, Lkpad, H);
  else
    hipLaunchKernelGGL(transpose_v_kernel<float>, grid, block, 0, s, (const float*)v, v_bs_el, v_hs_el, Lk, (float*)vt, Lkpad, H);
  return check_launch("transpose_v_kernel");
}

}  // namespace mh

extern "C" int mh_attention(const void* qk, int ld_qk, int k_col0, const void* vt, int Lpad, const float* bias,
                            void* out, int ld_out, int B, int L, int H, float scale, int band, int dtype,
                            void* stream) {
  return mh::attention(qk, ld_qk, k_col0, vt, Lpad, bias, out, ld_out, B, L, H, scale, band, dtype,
                       (hipStream_t)stream);
}
