// K7 / K8 / K9  osu_diffusion DiT denoiser (adaLN-Zero blocks) + DDPM p_sample update, fp32.
//   DiT.forward / forward_with_cfg   osu_diffusion/utils/models.py:281-317
//   FirstLayer / TimestepEmbedder / LabelEmbedder / DiTBlock / FinalLayer   models.py:20-55,103-210
//   timestep_embedding / position_sequence_embedding   utils/positional_embedding.py:29-49,66-77
//   GaussianDiffusion.p_mean_variance / p_sample      utils/diffusion/gaussian_diffusion.py:273-369,420-467
// Dense layers run on the exact-f32 MFMA atom (gemm.hip), attention on the banded flash kernel
// (attention.hip); this file holds the small glue kernels and the orchestration.
#include <stdlib.h>

#include "internal.hpp"

namespace mh {
namespace {

#define MH_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != MH_OK) return _rc; \
  } while (0)

// E[n*T + tp][:] = [ emb(512*x[n',0,tp]) (F) | emb(512*x[n',1,tp]) (F) | c[n,:,tp] (ctx) | 0 pad ],
// emb(v) = [cos(v f_i) i<F/2 | sin(v f_i) i<F/2]; n' = n mod (N/2) (forward_with_cfg duplicates the
// first half, models.py:306-307).
__global__ __launch_bounds__(256) void dit_embed_kernel(const float* __restrict__ x, const float* __restrict__ c,
                                                       const float* __restrict__ freqs, int N, int T, int F, int ctx,
                                                       int kpad, float* __restrict__ E) {
  const int row = blockIdx.x;  // n*T + tp
  const int n = row / T, tp = row - n * T;
  const int half_n = N / 2 > 0 ? N / 2 : 1;
  const int ns = n % half_n;
  const int hf = F / 2;
  float* er = E + (long)row * kpad;
  for (int j = threadIdx.x; j < kpad; j += 256) {
    float v = 0.f;
    if (j < 2 * F) {
      const int ch = j / F, i = j - ch * F;
      const float xv = x[((long)ns * 2 + ch) * T + tp] * 512.0f;
      const float a = xv * freqs[i < hf ? i : i - hf];
      v = i < hf ? cosf(a) : sinf(a);
    } else if (j < 2 * F + ctx) {
      v = c[((long)n * ctx + (j - 2 * F)) * T + tp];
    }
    er[j] = v;
  }
}

// timestep_embedding(t, F): [cos(t f_i) | sin(t f_i)];  row r uses t_vals[r / div] (div = N when the rows are
// (step, n) pairs of a whole sampling loop, 1 for a single forward)
__global__ void dit_tfreq_kernel(const int32_t* __restrict__ t_vals, int div, const float* __restrict__ freqs, int F,
                                 float* __restrict__ out) {
  const int r = blockIdx.x;
  const float tv = (float)t_vals[r / div];
  const int hf = F / 2;
  for (int i = threadIdx.x; i < F; i += blockDim.x) {
    const float a = tv * freqs[i < hf ? i : i - hf];
    out[(long)r * F + i] = i < hf ? cosf(a) : sinf(a);
  }
}

// dst[r][:] += src[r % N][:]   (label embedding broadcast over the steps)
__global__ void add_rows_bcast_kernel(float* __restrict__ dst, const float* __restrict__ src, int rows, int N, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * D) return;
  const int r = i / D, d = i - r * D;
  dst[i] += src[(long)(r % N) * D + d];
}

// conditioning of the current step: cur[:] = all[*sel][:]
__global__ void select_step_kernel(const float* __restrict__ all, const int* __restrict__ sel, long per_step,
                                   float* __restrict__ cur) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < per_step) cur[i] = all[(long)(*sel) * per_step + i];
}

// out[n][j] (+)= act_out?( sum_k act_in?(in[n][k]) * W[j][k] + b[j] );  one wave per (n, j)
template <bool SILU_IN, bool ACCUM>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ in, int ldi,
                                                          const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ b, float* __restrict__ out, int ldo,
                                                          int N, int J, int K, int silu_out) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= N * J) return;
  const int n = o / J, j = o - n * J;
  const float* ip = in + (long)n * ldi;
  const float* wp = W + (long)j * ldw;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    float v = ip[k];
    if (SILU_IN) v = v / (1.0f + expf(-v));
    s += v * wp[k];
  }
  s = wave_sum(s);
  if (lane == 0) {
    s += b ? b[j] : 0.f;
    if (silu_out) s = s / (1.0f + expf(-s));
    float* op = out + (long)n * ldo + j;
    *op = ACCUM ? *op + s : s;
  }
}

// final Linear(D -> 4) on the modulated activations, (N, T, 4) -> (N, 4, T), then CFG combine of the eps
// channels (models.py:312-317).  One wave per (sequence position, CFG pair).
__global__ __launch_bounds__(256) void dit_final_kernel(const float* __restrict__ xm, const float* __restrict__ W,
                                                       int ldw, const float* __restrict__ b, int N, int T, int D,
                                                       float cfg_scale, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int tp = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tp >= T) return;
  const int hn = N / 2;
  {
    const int n = blockIdx.y;   // one (position, CFG pair) per wave
    float v[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float* xr = xm + ((long)(n + s2 * hn) * T + tp) * D;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float s = 0.f;
        for (int k = lane; k < D; k += 64) s += xr[k] * W[(long)o * ldw + k];
        v[s2][o] = wave_sum(s) + b[o];
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const float cond = v[0][o], uncond = v[1][o];
        const float he = uncond + cfg_scale * (cond - uncond);
        out[((long)n * 4 + o) * T + tp] = he;
        out[((long)(n + hn) * 4 + o) * T + tp] = he;
      }
#pragma unroll
      for (int o = 2; o < 4; ++o) {
        out[((long)n * 4 + o) * T + tp] = v[0][o];
        out[((long)(n + hn) * 4 + o) * T + tp] = v[1][o];
      }
    }
  }
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* __restrict__ model_out,
                                                       const float* __restrict__ x, const float* __restrict__ noise,
                                                       const float* __restrict__ coef, const int* __restrict__ sel,
                                                       long noise_stride, const uint8_t* __restrict__ imask,
                                                       const float* __restrict__ iref,
                                                       const float* __restrict__ x0_override, int raw_pred, int N,
                                                       int T, float* __restrict__ x_out, float* __restrict__ pred) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * 2 * T) return;
  const int si = sel ? *sel : 0;
  const float* cf = coef + (long)si * 7;
  const float* nz = noise + (long)si * noise_stride;
  const int n = idx / (2 * T), rem = idx - n * 2 * T, ch = rem / T, tp = rem - ch * T;
  const float eps = model_out[((long)n * 4 + ch) * T + tp];
  const float var = model_out[((long)n * 4 + 2 + ch) * T + tp];
  const float xt = x[idx];
  const float min_log = cf[0], max_log = cf[1];
  const float frac = (var + 1.0f) / 2.0f;
  const float logvar = frac * max_log + (1.0f - frac) * min_log;
  float x0 = x0_override ? x0_override[idx] : cf[2] * xt - cf[3] * eps;
  if (raw_pred) {  // only report the eps -> x0 prediction (caller applies its own denoised_fn)
    if (pred) pred[idx] = x0;
    return;
  }
  if (imask) x0 = imask[idx] ? x0 : iref[idx];
  x0 = fminf(fmaxf(x0, -2.0f), 2.0f);
  const float mean = cf[4] * x0 + cf[5] * xt;
  const float smp = mean + (cf[6] * expf(0.5f * logvar)) * nz[idx];
  x_out[idx] = smp;
  if (pred) pred[idx] = x0;
}
#pragma clang fp contract(fast)

// y = silu(x), elementwise (input of every adaLN projection: nn.Sequential(SiLU, Linear), models.py:124-127)
__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float v = x[i]; y[i] = v / (1.0f + expf(-v)); }
}

__global__ void loop_dec_kernel(int* sel) {
  if (threadIdx.x == 0) *sel = *sel - 1;
}
__global__ void loop_set_kernel(int* sel, int v) {
  if (threadIdx.x == 0) *sel = v;
}

struct DiTBuf {
  float *E, *xs, *xm, *qk, *vt, *attn, *hid, *yemb1, *yemb, *cond_cur;
  float* stats;   // [D/16][N*T][2] row sums / sums of squares of xs, written by the GEMM that produced xs
  int* sel;
  int Tpad;
};

// conditioning vectors of ONE step, layout [n][depth][6D] followed by [n][2D] (final layer)
inline long cond_floats(const MhDiTConfig* c, int N) { return (long)N * (c->depth * 6 + 2) * c->hidden; }

int64_t dit_ws_layout(const MhDiTConfig* c, int N, int T, void* base, int64_t size, DiTBuf* b) {
  Arena ar(base, size);
  const int64_t NT = (int64_t)N * T;
  const int D = c->hidden, Tpad = round_up(T, 64);
  DiTBuf t;
  t.Tpad = Tpad;
  t.E = (float*)ar.take(NT * c->first_k_pad * 4);
  t.xs = (float*)ar.take(NT * D * 4);
  t.xm = (float*)ar.take(NT * D * 4);
  t.qk = (float*)ar.take(NT * 2 * D * 4);
  t.vt = (float*)ar.take((int64_t)N * D * Tpad * 4);
  t.attn = (float*)ar.take(NT * D * 4);
  t.hid = (float*)ar.take(NT * 4 * D * 4);
  t.yemb1 = (float*)ar.take((int64_t)N * D * 4);
  t.yemb = (float*)ar.take((int64_t)N * D * 4);
  t.cond_cur = (float*)ar.take(cond_floats(c, N) * 4);
  t.stats = (float*)ar.take((int64_t)ceil_div(D, 16) * NT * 2 * 4);
  t.sel = (int*)ar.take(256);
  if (b) *b = t;
  return ar.off;
}

// scratch of the conditioning pass for `rows` (= steps * N) rows: tfreq, temb1, bvec
int64_t cond_scratch_bytes(const MhDiTConfig* c, int rows) {
  return align256((int64_t)rows * c->t_freq_dim * 4) + 2 * align256((int64_t)rows * c->hidden * 4);
}

template <bool SILU_IN, bool ACCUM>
int small_linear(const float* in, int ldi, const float* W, int ldw, const float* b, float* out, int ldo, int N, int J,
                 int K, int silu_out, hipStream_t s) {
  hipLaunchKernelGGL((small_linear_kernel<SILU_IN, ACCUM>), dim3(ceil_div(N * J, 4)), dim3(256), 0, s, in, ldi, W, ldw,
                     b, out, ldo, N, J, K, silu_out);
  return check_launch("small_linear_kernel");
}

// ---- the block GEMMs of ONE chunk (M = N T <= a few hundred rows) as latency kernels ("skinny" form, round 5) -----------------
// The one-chunk DiT step is 48 GEMMs of 0.06-0.23 GFLOP behind one another (4 per block), and the LDS-tiled kernels take 10-13 us
// each for them: tools/dit_gemm_probe.py -- 7.6-8.7 us even with every operand hot in L2 and the plain epilogue, because their K
// loop is 2-6 dependent stages of (global load -> registers -> LDS -> barrier -> a serial chain of 32 exact-f32 MFMAs).  This form
// is the decode GEMV's (decode_kernels.hpp): a workgroup owns ONE 16-row x 16-column tile, its NWV waves split K, and EVERY operand
// of the tile -- its 16 activation rows and 16 weight rows over the whole K, the epilogue's old values, the modulation vectors --
// is requested before the first wait: ONE memory round trip, then K / (16 NWV) x 4 MFMAs per wave, a cross-wave sum through LDS in
// wave order (deterministic) and the epilogue.  The LayerNorm of a consuming GEMM (qkv, fc1) needs no statistics buffers: the
// workgroup holds its 16 rows over the whole K = hidden, so mean / variance come from the registers (lane -> the 4 lane groups ->
// the waves through LDS), then (x - mu) rstd (1 + scale) + shift is applied to the A fragments in place (models.py:18 `modulate`).
// fp32 only (the reference never casts the DiT), exact-f32 MFMA atom, K a multiple of 16; the k-block -> wave assignment and the
// summation order depend on K and NWV only, never on the number of rows or chunks.
enum { DSK_PRO_PLAIN = 0, DSK_PRO_LNMOD = 1 };
enum { DSK_EPI_QKV = 0, DSK_EPI_GATE = 1, DSK_EPI_GELU = 2 };
struct DitSkinnyP {
  const float* A; int lda;           // [M, lda] activations (PLAIN) or the residual stream (LNMOD)
  const float* W; int ldw;           // [N, ldw]
  const float* bias;                 // [N]
  int M, N, K, rows_per_batch;
  const float* shift; const float* scale; int mod_ld; float eps;      // LNMOD: per batch entry [mod_ld] rows
  float* out; int ldo;               // QKV: q | k columns [M, ldo]; GELU: [M, ldo]
  float* vt; int n_split, H, Tpad;   // QKV: columns >= n_split go to V^T [batch][H*64][Tpad]
  float* xs; int ldx; const float* gate; int gate_ld;                 // GATE: xs[m][n] += gate[batch][n] * (acc + bias[n])
};

// NF 16-column fragments per workgroup (round 6): the workgroup's activation rows are loaded ONCE for NF x 16 output columns.  A
// 16 x 16 tile re-reads every activation row N / 16 times and every weight row M / 16 times through L2 -- at DiT-B (hidden 768, one
// chunk = 256 rows) 113-302 MB of L2 -> CU traffic per GEMM, which is what bounded the form there (rocprofv3: fc2 35.9 us).  Each
// output element is still the same products in the same order (k-block -> wave assignment, in-wave order, wave-order sum): NF
// changes no bit of the result.
template <int NWV, int CH, int MF, int PRO, int EPI, int NF = 1>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(NWV / 4, 4)))   // registers are free here: hipcc must not trade a load in flight for one
void dit_skinny_kernel(DitSkinnyP p) {
  // MF 16-row fragments per workgroup (MF = 2: the weight fragments are shared by 32 rows -- half the workgroups, so that the wide
  // projections (N = 3 D, 4 D: 1152 / 1536 tiles of 16 x 16 at DiT-S) fit the chip in ONE round of resident workgroups; the host
  // picks it only when rows_per_batch % 32 == 0, i.e. a row block never straddles two batch entries)
  __shared__ f32x4_t red[NWV * MF * NF * 64];
  __shared__ float st1[PRO == DSK_PRO_LNMOD ? NWV : 1][MF * 16], st2[PRO == DSK_PRO_LNMOD ? NWV : 1][MF * 16];
  // LNMOD, MF = 2 (the workgroup's rows share ONE batch entry: host condition): the modulation vectors go through LDS -- one
  // 16-byte load per thread instead of 2 CH float4 per lane held across the statistics (48 registers: 184 -> ~120, i.e. four
  // resident workgroups per CU instead of two, and the 576 / 768 workgroups of the qkv / fc1 projections run in ONE round)
  constexpr bool kModLds = PRO == DSK_PRO_LNMOD && MF == 2;
  __shared__ __attribute__((aligned(16))) float modv[kModLds ? 2 * 16 * NWV * CH : 4];     // [scale | shift] over K <= 16 NWV CH
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 16 * NF, m0 = blockIdx.y * 16 * MF;
  const int nkb = p.K / 16;
  const float* Wp[NF];                                               // this lane's weight row of column fragment nf (B fragment row = l15 = output column)
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int wrow = (n0 + nf * 16 + l15) < p.N ? (n0 + nf * 16 + l15) : p.N - 1;
    Wp[nf] = p.W + (long)wrow * p.ldw + lg * 4;
  }
  const float* Ap[MF];
  int batch_a[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) {
    const int ar = (m0 + f * 16 + l15) < p.M ? (m0 + f * 16 + l15) : p.M - 1;   // this lane's activation row of fragment f
    Ap[f] = p.A + (long)ar * p.lda + lg * 4;
    batch_a[f] = ar / p.rows_per_batch;
  }
  // epilogue elements of this lane: unit = wid + u NWV -> fragment ef = unit >> 2, accumulator register r = unit & 3:
  // row m0 + ef*16 + lg*4 + r, column n0 + l15.  What the epilogue needs besides the product is requested with the operands.
  // (unit = (row fragment ef, column fragment nf, accumulator register r): ef = unit / (4 NF), nf = (unit >> 2) % NF, r = unit & 3)
  constexpr int UPW = (MF * NF * 4 + NWV - 1) / NWV;
  float e_bias[UPW], e_old[UPW], e_gate[UPW];
#pragma unroll
  for (int u = 0; u < UPW; ++u) {
    e_old[u] = 0.f; e_gate[u] = 0.f; e_bias[u] = 0.f;
    const int unit = wid + u * NWV;
    if (unit < MF * NF * 4) {
      const int ec0 = n0 + ((unit >> 2) % NF) * 16 + l15;
      const int ecol = ec0 < p.N ? ec0 : p.N - 1;
      e_bias[u] = p.bias[ecol];
      if (EPI == DSK_EPI_GATE) {
        int er = m0 + (unit / (4 * NF)) * 16 + lg * 4 + (unit & 3);
        er = er < p.M ? er : p.M - 1;
        e_old[u] = p.xs[(long)er * p.ldx + ecol];
        e_gate[u] = p.gate[(long)(er / p.rows_per_batch) * p.gate_ld + ecol];
      }
    }
  }
  f32x4_t acc[MF][NF];
#pragma unroll
  for (int f = 0; f < MF; ++f)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[f][nf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float4 mod_raw = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (kModLds) {   // thread t < K/4: scale[4t .. 4t+3]; K/4 <= t < K/2: shift (K/2 <= NWV * 64 * ... : checked below)
    const int t4 = tid * 4;
    const float* src = (t4 < p.K ? p.scale + t4 : p.shift + (t4 - p.K < p.K ? t4 - p.K : 0)) + (long)batch_a[0] * p.mod_ld;
    mod_raw = *reinterpret_cast<const float4*>(src);
  }
  int kb0 = wid;
  do {   // ONE pass unless K > 16 NWV CH (LNMOD: checked on the host); every wave runs at least one (the barrier of the statistics)
    float4 av[MF][CH], wv[NF][CH], sc[(PRO == DSK_PRO_LNMOD && !kModLds) ? CH : 1], sh[(PRO == DSK_PRO_LNMOD && !kModLds) ? CH : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kb = kb0 + NWV * c;
      const int kel = (kb < nkb ? kb : nkb - 1) * 16;          // clamped address, value masked below (no predicated load)
#pragma unroll
      for (int f = 0; f < MF; ++f) av[f][c] = *reinterpret_cast<const float4*>(Ap[f] + kel);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) wv[nf][c] = *reinterpret_cast<const float4*>(Wp[nf] + kel);
      if constexpr (PRO == DSK_PRO_LNMOD && !kModLds) {
        sc[c] = *reinterpret_cast<const float4*>(p.scale + (long)batch_a[0] * p.mod_ld + kel + lg * 4);
        sh[c] = *reinterpret_cast<const float4*>(p.shift + (long)batch_a[0] * p.mod_ld + kel + lg * 4);
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // every load of the pass is issued ABOVE this line (hipcc otherwise sinks each load to its MFMA)
    if (PRO == DSK_PRO_LNMOD) {
      // LayerNorm statistics of row l15 from the registers: this lane's 4 CH values -> the 4 lane groups -> the NWV waves, TWICE
      // (round 6, ADVICE r5: the mean first, then the sum of squares of the CENTRED values -- F.layer_norm's arithmetic; the
      // single-pass E[x^2] - mu^2 of round 5 loses every digit on rows whose mean dwarfs their spread)
      float mu_f[MF];
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float m = (kb0 + NWV * c < nkb) ? 1.f : 0.f;
          const float4 x = av[f][c];
          s1 += m * ((x.x + x.y) + (x.z + x.w));
        }
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        if (lg == 0) st1[wid][f * 16 + l15] = s1;
      }
      if constexpr (kModLds) {
        if (tid * 4 < 2 * p.K) *reinterpret_cast<float4*>(modv + tid * 4) = mod_raw;      // [0, K): scale, [K, 2K): shift
      }
      __syncthreads();
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        float t1 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) t1 += st1[w][f * 16 + l15];
        mu_f[f] = t1 / (float)p.K;
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float m = (kb0 + NWV * c < nkb) ? 1.f : 0.f;
          const float4 x = av[f][c];
          const float a0 = x.x - mu_f[f], a1 = x.y - mu_f[f], a2 = x.z - mu_f[f], a3 = x.w - mu_f[f];
          s2 += m * ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3));
        }
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if (lg == 0) st2[wid][f * 16 + l15] = s2;
      }
      __syncthreads();
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        float t2 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) t2 += st2[w][f * 16 + l15];
        const float mu = mu_f[f];
        const float rs = rsqrtf(t2 / (float)p.K + p.eps);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          float4 x = av[f][c], qs, qh;
          if constexpr (kModLds) {
            const int kb = kb0 + NWV * c;
            const int kel = (kb < nkb ? kb : nkb - 1) * 16 + lg * 4;
            qs = *reinterpret_cast<const float4*>(modv + kel);
            qh = *reinterpret_cast<const float4*>(modv + p.K + kel);
          } else {
            qs = sc[c]; qh = sh[c];
          }
          x.x = (x.x - mu) * rs * (1.f + qs.x) + qh.x;
          x.y = (x.y - mu) * rs * (1.f + qs.y) + qh.y;
          x.z = (x.z - mu) * rs * (1.f + qs.z) + qh.z;
          x.w = (x.w - mu) * rs * (1.f + qs.w) + qh.w;
          av[f][c] = x;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float keep = (kb0 + NWV * c < nkb) ? 1.f : 0.f;    // k-blocks beyond K contribute zeros
      // element i of every lane's 4-float vector forms one 16x16x4 product (a permutation of k, the same for both operands)
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const float ax = av[f][c].x * keep, ay = av[f][c].y * keep, az = av[f][c].z * keep, aw = av[f][c].w * keep;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          acc[f][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, wv[nf][c].x, acc[f][nf], 0, 0, 0);
          acc[f][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, wv[nf][c].y, acc[f][nf], 0, 0, 0);
          acc[f][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, wv[nf][c].z, acc[f][nf], 0, 0, 0);
          acc[f][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, wv[nf][c].w, acc[f][nf], 0, 0, 0);
        }
      }
    }
    kb0 += NWV * CH;
  } while (kb0 < nkb);
#pragma unroll
  for (int f = 0; f < MF; ++f)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) red[((wid * MF + f) * NF + nf) * 64 + lane] = acc[f][nf];
  __syncthreads();
  const float* redf = reinterpret_cast<const float*>(red);
#pragma unroll
  for (int u = 0; u < UPW; ++u) {
    const int unit = wid + u * NWV;
    if (unit >= MF * NF * 4) break;
    const int ef = unit / (4 * NF), enf = (unit >> 2) % NF, r = unit & 3;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) v += redf[(((w * MF + ef) * NF + enf) * 64 + lane) * 4 + r];       // wave order: deterministic
    v += e_bias[u];
    const int erow = m0 + ef * 16 + lg * 4 + r;
    const int ecol = n0 + enf * 16 + l15;
    if (erow >= p.M || ecol >= p.N) continue;
    const int batch_e = erow / p.rows_per_batch;
    if (EPI == DSK_EPI_GATE) {
      p.xs[(long)erow * p.ldx + ecol] = e_old[u] + e_gate[u] * v;
    } else if (EPI == DSK_EPI_GELU) {
      p.out[(long)erow * p.ldo + ecol] = gelu_tanh(v);
    } else {
      if (ecol < p.n_split) {
        p.out[(long)erow * p.ldo + ecol] = v;
      } else {
        const int c2 = ecol - p.n_split, key = erow - batch_e * p.rows_per_batch;
        p.vt[((long)batch_e * p.H * 64 + c2) * p.Tpad + key] = v;
      }
    }
  }
}

template <int PRO, int EPI>
int dit_skinny(const DitSkinnyP& p, hipStream_t s) {
  MH_REQUIRE(p.K % 16 == 0 && p.K >= 16 && p.lda % 4 == 0 && p.ldw % 4 == 0, "dit skinny GEMM: K %% 16 == 0, 16-byte aligned rows");
  const int nkb = p.K / 16;
  // 32 rows per workgroup where 16 would need more than ~3 resident workgroups per CU (the register budget of the LayerNorm form)
  const bool mf2 = p.rows_per_batch % 32 == 0 && (long)ceil_div(p.N, 16) * ceil_div(p.M, 16) > 768;
  dim3 grid(ceil_div(p.N, 16), ceil_div(p.M, mf2 ? 32 : 16));
  if (PRO == DSK_PRO_LNMOD) {
    MH_REQUIRE(nkb <= 8 * 6 && p.mod_ld % 4 == 0, "dit skinny GEMM: the LayerNorm prologue holds the whole row in registers (K <= 768)");
    if (nkb <= 4 * 6) {
      if (mf2) hipLaunchKernelGGL((dit_skinny_kernel<4, 6, 2, PRO, EPI>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((dit_skinny_kernel<4, 6, 1, PRO, EPI>), grid, dim3(256), 0, s, p);
    } else {
      if (mf2) hipLaunchKernelGGL((dit_skinny_kernel<8, 6, 2, PRO, EPI>), grid, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((dit_skinny_kernel<8, 6, 1, PRO, EPI>), grid, dim3(512), 0, s, p);
    }
  } else if constexpr (PRO == DSK_PRO_PLAIN) {
    // wide form (round 6, hidden > 512): 32 rows x NF x 16 columns per workgroup -- the L2 -> CU traffic of a 16 x 16 tile is what
    // bounds these GEMMs at DiT-B (see the kernel's header); bit-identical to the narrow form
    // (measured at DiT-B, one chunk, ms per 100 steps: NF = 1 138.6, NF = 2 129.4, NF = 4 140.5 -- four column fragments leave
    // 288-384 workgroups for 256 CUs; at DiT-S dims (K = 384) the narrow forms stay faster: 59.6 vs 82.4)
    const bool wide = p.rows_per_batch % 32 == 0 && p.N % 32 == 0 && p.K >= 768;
    if (wide && nkb <= 8 * 6) {
      dim3 gw(p.N / 32, ceil_div(p.M, 32));
      hipLaunchKernelGGL((dit_skinny_kernel<8, 6, 2, PRO, EPI, 2>), gw, dim3(512), 0, s, p);
    } else if (wide && nkb > 16 * 6) {
      dim3 gw(p.N / 32, ceil_div(p.M, 32));
      hipLaunchKernelGGL((dit_skinny_kernel<8, 12, 2, PRO, EPI, 2>), gw, dim3(512), 0, s, p);
    } else if (nkb <= 4 * 6) {
      if (mf2) hipLaunchKernelGGL((dit_skinny_kernel<4, 6, 2, PRO, EPI>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((dit_skinny_kernel<4, 6, 1, PRO, EPI>), grid, dim3(256), 0, s, p);
    } else if (nkb <= 8 * 6) {
      if (mf2) hipLaunchKernelGGL((dit_skinny_kernel<8, 6, 2, PRO, EPI>), grid, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((dit_skinny_kernel<8, 6, 1, PRO, EPI>), grid, dim3(512), 0, s, p);
    } else if (nkb <= 16 * 6 && p.rows_per_batch % 32 == 0) {
      // long K (fc2: K = 4 D), 32 rows x 16 columns per workgroup, SIXTEEN waves split K: 6 k-blocks per wave in one round trip
      dim3 g2(ceil_div(p.N, 16), ceil_div(p.M, 32));
      hipLaunchKernelGGL((dit_skinny_kernel<16, 6, 2, PRO, EPI>), g2, dim3(1024), 0, s, p);
    } else {
      if (mf2) hipLaunchKernelGGL((dit_skinny_kernel<8, 12, 2, PRO, EPI>), grid, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((dit_skinny_kernel<8, 12, 1, PRO, EPI>), grid, dim3(512), 0, s, p);
    }
  }
  return check_launch("dit_skinny_kernel");
}

int check_dit(const MhDiTConfig* c, int N, int T) {
  MH_REQUIRE(c, "dit: null config");
  MH_REQUIRE(c->hidden == c->n_heads * 64, "dit: hidden must be n_heads*64");
  MH_REQUIRE(c->depth <= MH_MAX_LAYERS && c->in_channels == 2, "dit: unsupported depth / in_channels");
  MH_REQUIRE(c->first_k_pad % 32 == 0 && c->first_k_pad >= 2 * c->freq_dim + c->context_size, "dit: bad first_k_pad");
  MH_REQUIRE(c->class_pad >= c->class_size, "dit: bad class_pad");
  MH_REQUIRE(N >= 2 && N % 2 == 0 && T > 0, "dit: N must be even (CFG batch) and T > 0");
  MH_REQUIRE(c->operand_dtype == MH_F32 || c->operand_dtype == MH_BF16 || c->operand_dtype == MH_MX8, "dit: operand_dtype must be MH_F32, MH_BF16 or MH_MX8");
  MH_REQUIRE(c->operand_dtype == MH_F32 || c->hidden % 8 == 0, "dit: bf16 operands need hidden %% 8 == 0");
  MH_REQUIRE(c->operand_dtype != MH_MX8 || c->hidden % 128 == 0, "dit: MX-fp8 operands need hidden %% 128 == 0");
  return MH_OK;
}

// adaLN conditioning depends only on (t, y), never on x: b = t_embedder(t) + y_embedder(y), then every block's
// SiLU -> Linear(D, 6D) and the final layer's Linear(D, 2D) (models.py:20-55,124-127,140-147,166-173).  Computed for
// `steps` timesteps at once (rows = steps * N): a sampling loop hoists ALL of it out of the per-step graph.
// out layout: [step][n][depth][6D] ... then [step][n][2D] interleaved per step: step block = cond_floats().
int dit_conditioning(const MhDiTConfig* c, const MhDiTWeights* w, const int32_t* t_vals, int div, const float* y, int N,
                     int steps, float* cond_out, const DiTBuf& b, void* scratch, hipStream_t s) {
  const int D = c->hidden, rows = steps * N;
  Arena ar(scratch, cond_scratch_bytes(c, rows));
  float* tfreq = (float*)ar.take((int64_t)rows * c->t_freq_dim * 4);
  float* temb1 = (float*)ar.take((int64_t)rows * D * 4);
  float* bvec = (float*)ar.take((int64_t)rows * D * 4);
  hipLaunchKernelGGL(dit_tfreq_kernel, dim3(rows), dim3(256), 0, s, t_vals, div, w->t_freqs, c->t_freq_dim, tfreq);
  MH_TRY(check_launch("dit_tfreq_kernel"));
  MH_TRY((small_linear<false, false>(tfreq, c->t_freq_dim, w->t_w0, c->t_freq_dim, w->t_b0, temb1, D, rows, D,
                                     c->t_freq_dim, 1, s)));
  MH_TRY((small_linear<false, false>(temb1, D, w->t_w1, D, w->t_b1, bvec, D, rows, D, D, 0, s)));
  MH_TRY((small_linear<false, false>(y, c->class_size, w->y_w0, c->class_pad, w->y_b0, b.yemb1, D, N, D, c->class_size,
                                     1, s)));
  MH_TRY((small_linear<false, false>(b.yemb1, D, w->y_w1, D, w->y_b1, b.yemb, D, N, D, D, 0, s)));
  hipLaunchKernelGGL(add_rows_bcast_kernel, dim3(ceil_div(rows * D, 256)), dim3(256), 0, s, bvec, b.yemb, rows, N, D);
  MH_TRY(check_launch("add_rows_bcast_kernel"));
  // row r = step*N + n of the output is [depth][6D] | [2D]; a step block is N consecutive rows
  const int ld_row = c->depth * 6 * D + 2 * D;   // per (step, n) row: [depth][6D] | [2D]
  // SiLU once, then the depth + 1 adaLN projections on the MFMA GEMM (a sampling loop hoists ALL steps: rows = steps * N,
  // 6400 x 2304 x 384 per block for 32 chunks -- one wave per output element took 5 ms per block there)
  float* sb = temb1;   // (free again: bvec is complete)
  hipLaunchKernelGGL(silu_kernel, dim3((unsigned)ceil_div(rows * D, 256)), dim3(256), 0, s, bvec, sb, (long)rows * D);
  MH_TRY(check_launch("silu_kernel"));
  MhGemm g;
  for (int l = 0; l <= c->depth; ++l) {
    const bool fin = l == c->depth;
    g = MhGemm{};
    g.A = sb; g.lda = D; g.W = fin ? w->fin_ada_w : w->ada_w[l]; g.ldw = D; g.C = cond_out + (long)l * 6 * D; g.ldc = ld_row;
    g.M = rows; g.N = fin ? 2 * D : 6 * D; g.K = D; g.bias = fin ? w->fin_ada_b : w->ada_b[l]; g.dtype = MH_F32;
    g.epilogue = MH_EPI_STORE_F32;
    MH_TRY(gemm(g, s, /*ascending_k=*/true));   // same bits for 1 step or 100, 1 chunk or 32
  }
  return MH_OK;
}

// x-dependent part of the denoiser; the conditioning of this step is in b.cond_cur ([n][depth*6D + 2D])
int dit_body(const MhDiTConfig* c, const MhDiTWeights* w, const float* x, const float* cc, float cfg_scale, int band,
             int open_from, int N, int T, float* out, const DiTBuf& b, hipStream_t s) {
  const int D = c->hidden, H = c->n_heads, NT = N * T;
  const int ld_row = c->depth * 6 * D + 2 * D;
  hipLaunchKernelGGL(dit_embed_kernel, dim3(NT), dim3(256), 0, s, x, cc, w->pos_freqs, N, T, c->freq_dim,
                     c->context_size, c->first_k_pad, b.E);
  MH_TRY(check_launch("dit_embed_kernel"));
  // LayerNorm + adaLN modulate never runs as its own pass inside the blocks: every GEMM that writes the residual
  // stream xs also emits per-row (sum, sum of squares) per 16-column strip, and the GEMM that consumes
  // modulate(LN(xs)) normalises its A operand on the way into LDS (mh_gemm: stats_out / ln_stats).
  const bool fuse_ln = (D % 16 == 0);
  const int strips = D / 16;
  // big denoiser batches (many chunks stacked): the five large projections run as bf16 x 3 on the bf16 matrix cores
  // (MhGemm.w_split3); a single chunk keeps the exact-fp32 kernels.  The split path takes its LayerNorm + modulate from
  // the stand-alone pass: fusing it into the split kernel's A load (an option until round 5) is slower
  // (32 chunks x 100 steps: 303.7 vs 292.3 ms -- the modulate arithmetic sits in front of every split store).  History:
  // an earlier build of that fused form returned non-repeatable rows (6, 7 mod 8) on grids of > 1000 workgroups
  // (tools/s3_ln_probe.py); the present code shape is repeatable and exact against the un-fused form
  // (tests/test_gpu_kernels.py::test_gemm_bf16x3_with_fused_layernorm_is_repeatable); the cause was not isolated in the ISA.
  const long s3min = option(OPT_DIT_SPLIT3_MIN_ROWS);
  const bool s3 = s3min > 0 && NT >= s3min && w->first_w3 && D % 32 == 0 && c->first_k_pad % 32 == 0;
  auto weights = [&](const float* exact, const void* split, MhGemm& gg) {
    const bool use = s3 && split != nullptr;
    gg.W = use ? split : (const void*)exact;
    gg.w_split3 = use ? 1 : 0;
  };
  MhGemm g = MhGemm{};
  g.A = b.E; g.lda = c->first_k_pad; g.ldw = c->first_k_pad; g.C = b.xs; g.ldc = D; g.M = NT; g.N = D;
  g.K = c->first_k_pad; g.bias = w->first_b; g.dtype = MH_F32; g.epilogue = MH_EPI_STORE_F32;
  weights(w->first_w, w->first_w3, g);
  g.stats_out = fuse_ln ? b.stats : nullptr;
  MH_TRY(gemm(g, s));
  const bool lowp = c->operand_dtype == MH_BF16;
  if (lowp)
    for (int l = 0; l < c->depth; ++l)
      MH_REQUIRE(w->qkv_wb[l] && w->out_wb[l] && w->fc1_wb[l] && w->fc2_wb[l], "dit: operand_dtype = MH_BF16 needs the bf16 weight copies (layer %d)", l);
  for (int l = 0; l < c->depth && lowp; ++l) {
    // bf16 operands (BASELINE configs[4]): the same block, every GEMM operand rounded to bf16 where it is produced; the
    // activation buffers are the fp32 ones reinterpreted (half used).  128 x 128 LDS-DMA tiles for the wide outputs.
    const float* mod = b.cond_cur + (long)l * 6 * D;
    MH_TRY(ln_modulate(b.xs, D, mod + 0 * D, mod + 1 * D, ld_row, T, b.xm, D, NT, D, 1e-6f, MH_BF16, s));
    g = MhGemm{};
    g.A = b.xm; g.lda = D; g.W = w->qkv_wb[l]; g.ldw = D; g.C = b.qk; g.ldc = 2 * D; g.M = NT; g.N = 3 * D; g.K = D;
    g.bias = w->qkv_b[l]; g.dtype = MH_BF16; g.epilogue = MH_EPI_QKV_VT; g.C2 = b.vt; g.n_split = 2 * D; g.kv_B = N;
    g.kv_H = H; g.kv_L = T; g.kv_Lpad = b.Tpad;
    MH_TRY(gemm(g, s));
    MH_TRY(attention(b.qk, 2 * D, D, b.vt, b.Tpad, nullptr, b.attn, D, N, T, H, 0.125f, band, MH_BF16, s, open_from));
    g = MhGemm{};
    g.A = b.attn; g.lda = D; g.W = w->out_wb[l]; g.ldw = D; g.C = b.xs; g.ldc = D; g.M = NT; g.N = D; g.K = D;
    g.bias = w->out_b[l]; g.gate = mod + 2 * D; g.gate_ld = ld_row; g.rows_per_batch = T; g.dtype = MH_BF16;
    g.epilogue = MH_EPI_GATE_RESID;
    MH_TRY(gemm(g, s));
    MH_TRY(ln_modulate(b.xs, D, mod + 3 * D, mod + 4 * D, ld_row, T, b.xm, D, NT, D, 1e-6f, MH_BF16, s));
    g = MhGemm{};
    g.A = b.xm; g.lda = D; g.W = w->fc1_wb[l]; g.ldw = D; g.C = b.hid; g.ldc = 4 * D; g.M = NT; g.N = 4 * D; g.K = D;
    g.bias = w->fc1_b[l]; g.dtype = MH_BF16; g.epilogue = MH_EPI_BIAS_GELU;
    MH_TRY(gemm(g, s));
    g = MhGemm{};
    g.A = b.hid; g.lda = 4 * D; g.W = w->fc2_wb[l]; g.ldw = 4 * D; g.C = b.xs; g.ldc = D; g.M = NT; g.N = D; g.K = 4 * D;
    g.bias = w->fc2_b[l]; g.gate = mod + 5 * D; g.gate_ld = ld_row; g.rows_per_batch = T; g.dtype = MH_BF16;
    g.epilogue = MH_EPI_GATE_RESID;
    MH_TRY(gemm(g, s));
  }
  // MX-fp8 operands (BASELINE configs[4] "fp8 MFMA"): the block GEMMs on v_mfma_scale_f32_16x16x128_f8f6f4; LayerNorm-modulate
  // writes its output as the MX operand directly, the attention output and the GELU hidden (bf16, as in the bf16-operand mode)
  // are quantised by a pass of their own.  The MX copies live in the unused upper halves of the fp32-sized activation buffers.
  const bool lowp8 = c->operand_dtype == MH_MX8;
  for (int l = 0; l < c->depth && lowp8; ++l) {
    MH_REQUIRE(w->qkv_wm[l] && w->qkv_wms[l] && w->out_wm[l] && w->out_wms[l] && w->fc1_wm[l] && w->fc1_wms[l] && w->fc2_wm[l] && w->fc2_wms[l],
               "dit: operand_dtype = MH_MX8 needs the MX-fp8 weight copies (layer %d)", l);
    const float* mod = b.cond_cur + (long)l * 6 * D;
    uint8_t* xq = reinterpret_cast<uint8_t*>(b.xm);                    uint8_t* xqs = xq + (long)NT * D;            // [NT][D] | scales
    uint8_t* aq = reinterpret_cast<uint8_t*>(b.attn) + (long)NT * D * 2;  uint8_t* aqs = aq + (long)NT * D;         // behind the bf16 attention output
    uint8_t* hq = reinterpret_cast<uint8_t*>(b.hid) + (long)NT * 4 * D * 2; uint8_t* hqs = hq + (long)NT * 4 * D;  // behind the bf16 hidden
    MH_TRY(ln_modulate_mx8(b.xs, D, mod + 0 * D, mod + 1 * D, ld_row, T, NT, D, 1e-6f, xq, D, xqs, s));
    g = MhGemm{};
    g.A = xq; g.lda = D; g.a_scale = xqs; g.W = w->qkv_wm[l]; g.ldw = D; g.w_scale = w->qkv_wms[l]; g.C = b.qk; g.ldc = 2 * D; g.M = NT; g.N = 3 * D; g.K = D;
    g.bias = w->qkv_b[l]; g.dtype = MH_MX8; g.epilogue = MH_EPI_QKV_VT; g.C2 = b.vt; g.n_split = 2 * D; g.kv_B = N;
    g.kv_H = H; g.kv_L = T; g.kv_Lpad = b.Tpad;
    MH_TRY(gemm(g, s));
    MH_TRY(attention(b.qk, 2 * D, D, b.vt, b.Tpad, nullptr, b.attn, D, N, T, H, 0.125f, band, MH_BF16, s, open_from));
    MH_TRY(quantize_mx8(b.attn, D, NT, D, MH_BF16, aq, D, aqs, s));
    g = MhGemm{};
    g.A = aq; g.lda = D; g.a_scale = aqs; g.W = w->out_wm[l]; g.ldw = D; g.w_scale = w->out_wms[l]; g.C = b.xs; g.ldc = D; g.M = NT; g.N = D; g.K = D;
    g.bias = w->out_b[l]; g.gate = mod + 2 * D; g.gate_ld = ld_row; g.rows_per_batch = T; g.dtype = MH_MX8;
    g.epilogue = MH_EPI_GATE_RESID;
    MH_TRY(gemm(g, s));
    MH_TRY(ln_modulate_mx8(b.xs, D, mod + 3 * D, mod + 4 * D, ld_row, T, NT, D, 1e-6f, xq, D, xqs, s));
    g = MhGemm{};
    g.A = xq; g.lda = D; g.a_scale = xqs; g.W = w->fc1_wm[l]; g.ldw = D; g.w_scale = w->fc1_wms[l]; g.C = b.hid; g.ldc = 4 * D; g.M = NT; g.N = 4 * D; g.K = D;
    g.bias = w->fc1_b[l]; g.dtype = MH_MX8; g.epilogue = MH_EPI_BIAS_GELU;
    const bool mx_fused = (4 * D) % 128 == 0;   // the GELU hidden leaves fc1 as fc2's MX-fp8 operand
    if (mx_fused) {
      if ((4 * D) % 512) MH_REQUIRE(hipMemsetAsync(hqs, 0, (size_t)NT * mx8_scale_row_bytes(4 * D), s) == hipSuccess, "dit: scale reset failed");
      g.mx_out = hq; g.mx_out_scales = hqs; g.ldc = 4 * D;
    }
    MH_TRY(gemm(g, s));
    if (!mx_fused) MH_TRY(quantize_mx8(b.hid, 4 * D, NT, 4 * D, MH_BF16, hq, 4 * D, hqs, s));
    g = MhGemm{};
    g.A = hq; g.lda = 4 * D; g.a_scale = hqs; g.W = w->fc2_wm[l]; g.ldw = 4 * D; g.w_scale = w->fc2_wms[l]; g.C = b.xs; g.ldc = D; g.M = NT; g.N = D; g.K = 4 * D;
    g.bias = w->fc2_b[l]; g.gate = mod + 5 * D; g.gate_ld = ld_row; g.rows_per_batch = T; g.dtype = MH_MX8;
    g.epilogue = MH_EPI_GATE_RESID;
    MH_TRY(gemm(g, s));
  }
  // fp32 semantics, big batches: the bf16 x 3 GEMMs with BOTH operands pre-split (gemm_s3g_kernel: three-stage LDS-DMA,
  // no conversion work inside the GEMM) -- LayerNorm-modulate, the attention and the GELU epilogue write their outputs as
  // [32 x bf16 hi | 32 x bf16 lo] per 32 values straight away (the same bytes as fp32, the same buffers).
  const bool s3g = !lowp && !lowp8 && s3 && D % 32 == 0;
  for (int l = 0; l < c->depth && s3g; ++l) {
    const float* mod = b.cond_cur + (long)l * 6 * D;
    MH_TRY(ln_modulate(b.xs, D, mod + 0 * D, mod + 1 * D, ld_row, T, b.xm, D, NT, D, 1e-6f, MH_LN_SPLIT3, s));
    g = MhGemm{};
    g.A = b.xm; g.lda = D; g.W = w->qkv_w3[l]; g.ldw = D; g.C = b.qk; g.ldc = 2 * D; g.M = NT; g.N = 3 * D; g.K = D;
    g.bias = w->qkv_b[l]; g.dtype = MH_F32; g.epilogue = MH_EPI_QKV_VT; g.C2 = b.vt; g.n_split = 2 * D; g.kv_B = N;
    g.kv_H = H; g.kv_L = T; g.kv_Lpad = b.Tpad; g.w_split3 = 3;
    MH_TRY(gemm(g, s));
    MH_TRY(attention(b.qk, 2 * D, D, b.vt, b.Tpad, nullptr, b.attn, D, N, T, H, 0.125f, band, MH_F32, s, open_from, /*out_split3=*/1));
    g = MhGemm{};
    g.A = b.attn; g.lda = D; g.W = w->out_w3[l]; g.ldw = D; g.C = b.xs; g.ldc = D; g.M = NT; g.N = D; g.K = D;
    g.bias = w->out_b[l]; g.gate = mod + 2 * D; g.gate_ld = ld_row; g.rows_per_batch = T; g.dtype = MH_F32;
    g.epilogue = MH_EPI_GATE_RESID; g.w_split3 = 3;
    MH_TRY(gemm(g, s));
    MH_TRY(ln_modulate(b.xs, D, mod + 3 * D, mod + 4 * D, ld_row, T, b.xm, D, NT, D, 1e-6f, MH_LN_SPLIT3, s));
    g = MhGemm{};
    g.A = b.xm; g.lda = D; g.W = w->fc1_w3[l]; g.ldw = D; g.C = b.hid; g.ldc = 4 * D; g.M = NT; g.N = 4 * D; g.K = D;
    g.bias = w->fc1_b[l]; g.dtype = MH_F32; g.epilogue = MH_EPI_BIAS_GELU; g.w_split3 = 7;
    MH_TRY(gemm(g, s));
    g = MhGemm{};
    g.A = b.hid; g.lda = 4 * D; g.W = w->fc2_w3[l]; g.ldw = 4 * D; g.C = b.xs; g.ldc = D; g.M = NT; g.N = D; g.K = 4 * D;
    g.bias = w->fc2_b[l]; g.gate = mod + 5 * D; g.gate_ld = ld_row; g.rows_per_batch = T; g.dtype = MH_F32;
    g.epilogue = MH_EPI_GATE_RESID; g.w_split3 = 3;
    MH_TRY(gemm(g, s));
  }
  // one chunk (a few hundred rows): the four block GEMMs as one-round-trip latency kernels (dit_skinny_kernel above)
  // (hidden <= 512: DiT-XS / -S.  At DiT-B's 768 the LayerNorm form needs 180 registers in 8-wave workgroups -- one per CU for 1152-2304
  // tiles: hidden > 512 takes the pre-pass form below)
  const long sk_rows = option(OPT_DIT_SKINNY_MAX_ROWS);
  const bool skinny = !lowp && !lowp8 && !s3 && NT <= sk_rows && D % 16 == 0 && D <= 512;
  for (int l = 0; l < c->depth && skinny; ++l) {
    const float* mod = b.cond_cur + (long)l * 6 * D;
    DitSkinnyP q{};
    q.A = b.xs; q.lda = D; q.W = w->qkv_w[l]; q.ldw = D; q.bias = w->qkv_b[l]; q.M = NT; q.N = 3 * D; q.K = D; q.rows_per_batch = T;
    q.shift = mod + 0 * D; q.scale = mod + 1 * D; q.mod_ld = ld_row; q.eps = 1e-6f;
    q.out = b.qk; q.ldo = 2 * D; q.vt = b.vt; q.n_split = 2 * D; q.H = H; q.Tpad = b.Tpad;
    MH_TRY((dit_skinny<DSK_PRO_LNMOD, DSK_EPI_QKV>(q, s)));
    MH_TRY(attention(b.qk, 2 * D, D, b.vt, b.Tpad, nullptr, b.attn, D, N, T, H, 0.125f, band, MH_F32, s, open_from));
    q = DitSkinnyP{};
    q.A = b.attn; q.lda = D; q.W = w->out_w[l]; q.ldw = D; q.bias = w->out_b[l]; q.M = NT; q.N = D; q.K = D; q.rows_per_batch = T;
    q.xs = b.xs; q.ldx = D; q.gate = mod + 2 * D; q.gate_ld = ld_row;
    MH_TRY((dit_skinny<DSK_PRO_PLAIN, DSK_EPI_GATE>(q, s)));
    q = DitSkinnyP{};
    q.A = b.xs; q.lda = D; q.W = w->fc1_w[l]; q.ldw = D; q.bias = w->fc1_b[l]; q.M = NT; q.N = 4 * D; q.K = D; q.rows_per_batch = T;
    q.shift = mod + 3 * D; q.scale = mod + 4 * D; q.mod_ld = ld_row; q.eps = 1e-6f; q.out = b.hid; q.ldo = 4 * D;
    MH_TRY((dit_skinny<DSK_PRO_LNMOD, DSK_EPI_GELU>(q, s)));
    // fc2 (K = 4 D): 32 rows x 16 columns per workgroup, sixteen waves split K (6 k-blocks each: still one round trip); a 16-row tile
    // on eight waves measured 13.4 us and the LDS-tiled split-K kernel 12.3 us against ~10 us for this form (62.1 -> 60.3 ms per 100 steps)
    q = DitSkinnyP{};
    q.A = b.hid; q.lda = 4 * D; q.W = w->fc2_w[l]; q.ldw = 4 * D; q.bias = w->fc2_b[l]; q.M = NT; q.N = D; q.K = 4 * D; q.rows_per_batch = T;
    q.xs = b.xs; q.ldx = D; q.gate = mod + 5 * D; q.gate_ld = ld_row;
    MH_TRY((dit_skinny<DSK_PRO_PLAIN, DSK_EPI_GATE>(q, s)));
  }
  // one chunk at hidden > 512 (DiT-B: the released diffusion checkpoint, configs/diffusion/v1.yaml:11; round 6): the LayerNorm-
  // modulate as its own tiny pass (5 us for 256 x 768), then all four block GEMMs in the PLAIN skinny form -- 32 rows x 32 columns
  // per workgroup (dit_skinny_kernel NF = 2).  The LayerNorm-in-registers form needs ~180 VGPRs at K = 768 (one 8-wave workgroup
  // per CU; 182.8 ms per 100 steps) and the LDS-tiled GEMMs with the fused LayerNorm prologue take 178.9; this form 129.0
  // (profiles/r06_dit_b_one_chunk.txt).  Same arithmetic per output element as the narrow skinny form.
  const bool skinny_pre = !lowp && !lowp8 && !s3 && !skinny && NT <= sk_rows && D > 512 && D % 16 == 0;
  for (int l = 0; l < c->depth && skinny_pre; ++l) {
    const float* mod = b.cond_cur + (long)l * 6 * D;
    MH_TRY(ln_modulate(b.xs, D, mod + 0 * D, mod + 1 * D, ld_row, T, b.xm, D, NT, D, 1e-6f, MH_F32, s));
    DitSkinnyP q{};
    q.A = b.xm; q.lda = D; q.W = w->qkv_w[l]; q.ldw = D; q.bias = w->qkv_b[l]; q.M = NT; q.N = 3 * D; q.K = D; q.rows_per_batch = T;
    q.out = b.qk; q.ldo = 2 * D; q.vt = b.vt; q.n_split = 2 * D; q.H = H; q.Tpad = b.Tpad;
    MH_TRY((dit_skinny<DSK_PRO_PLAIN, DSK_EPI_QKV>(q, s)));
    MH_TRY(attention(b.qk, 2 * D, D, b.vt, b.Tpad, nullptr, b.attn, D, N, T, H, 0.125f, band, MH_F32, s, open_from));
    q = DitSkinnyP{};
    q.A = b.attn; q.lda = D; q.W = w->out_w[l]; q.ldw = D; q.bias = w->out_b[l]; q.M = NT; q.N = D; q.K = D; q.rows_per_batch = T;
    q.xs = b.xs; q.ldx = D; q.gate = mod + 2 * D; q.gate_ld = ld_row;
    MH_TRY((dit_skinny<DSK_PRO_PLAIN, DSK_EPI_GATE>(q, s)));
    MH_TRY(ln_modulate(b.xs, D, mod + 3 * D, mod + 4 * D, ld_row, T, b.xm, D, NT, D, 1e-6f, MH_F32, s));
    q = DitSkinnyP{};
    q.A = b.xm; q.lda = D; q.W = w->fc1_w[l]; q.ldw = D; q.bias = w->fc1_b[l]; q.M = NT; q.N = 4 * D; q.K = D; q.rows_per_batch = T;
    q.out = b.hid; q.ldo = 4 * D;
    MH_TRY((dit_skinny<DSK_PRO_PLAIN, DSK_EPI_GELU>(q, s)));
    q = DitSkinnyP{};
    q.A = b.hid; q.lda = 4 * D; q.W = w->fc2_w[l]; q.ldw = 4 * D; q.bias = w->fc2_b[l]; q.M = NT; q.N = D; q.K = 4 * D; q.rows_per_batch = T;
    q.xs = b.xs; q.ldx = D; q.gate = mod + 5 * D; q.gate_ld = ld_row;
    MH_TRY((dit_skinny<DSK_PRO_PLAIN, DSK_EPI_GATE>(q, s)));
  }
  for (int l = 0; l < c->depth && !lowp && !lowp8 && !s3g && !skinny && !skinny_pre; ++l) {
    const float* mod = b.cond_cur + (long)l * 6 * D;
    // attention branch
    g = MhGemm{};
    g.A = b.xs; g.lda = D; g.ldw = D; g.C = b.qk; g.ldc = 2 * D; g.M = NT; g.N = 3 * D; g.K = D;
    g.bias = w->qkv_b[l]; g.dtype = MH_F32; g.epilogue = MH_EPI_QKV_VT; g.C2 = b.vt; g.n_split = 2 * D; g.kv_B = N;
    g.kv_H = H; g.kv_L = T; g.kv_Lpad = b.Tpad;
    weights(w->qkv_w[l], w->qkv_w3[l], g);
    if (fuse_ln && !g.w_split3) {
      g.ln_stats = b.stats; g.ln_strips = strips; g.ln_shift = mod + 0 * D; g.ln_scale = mod + 1 * D; g.ln_ld = ld_row;
      g.ln_eps = 1e-6f; g.rows_per_batch = T;
    } else {
      MH_TRY(ln_modulate(b.xs, D, mod + 0 * D, mod + 1 * D, ld_row, T, b.xm, D, NT, D, 1e-6f, MH_F32, s));
      g.A = b.xm;
    }
    MH_TRY(gemm(g, s));
    MH_TRY(attention(b.qk, 2 * D, D, b.vt, b.Tpad, nullptr, b.attn, D, N, T, H, 0.125f, band, MH_F32, s, open_from));
    g = MhGemm{};
    g.A = b.attn; g.lda = D; g.ldw = D; g.C = b.xs; g.ldc = D; g.M = NT; g.N = D; g.K = D;
    g.bias = w->out_b[l]; g.gate = mod + 2 * D; g.gate_ld = ld_row; g.rows_per_batch = T; g.dtype = MH_F32;
    g.epilogue = MH_EPI_GATE_RESID; g.stats_out = fuse_ln ? b.stats : nullptr;
    weights(w->out_w[l], w->out_w3[l], g);
    MH_TRY(gemm(g, s));
    // MLP branch
    g = MhGemm{};
    g.A = b.xs; g.lda = D; g.ldw = D; g.C = b.hid; g.ldc = 4 * D; g.M = NT; g.N = 4 * D; g.K = D;
    g.bias = w->fc1_b[l]; g.dtype = MH_F32; g.epilogue = MH_EPI_BIAS_GELU;
    weights(w->fc1_w[l], w->fc1_w3[l], g);
    if (fuse_ln && !g.w_split3) {
      g.ln_stats = b.stats; g.ln_strips = strips; g.ln_shift = mod + 3 * D; g.ln_scale = mod + 4 * D; g.ln_ld = ld_row;
      g.ln_eps = 1e-6f; g.rows_per_batch = T;
    } else {
      MH_TRY(ln_modulate(b.xs, D, mod + 3 * D, mod + 4 * D, ld_row, T, b.xm, D, NT, D, 1e-6f, MH_F32, s));
      g.A = b.xm;
    }
    MH_TRY(gemm(g, s));
    g = MhGemm{};
    g.A = b.hid; g.lda = 4 * D; g.ldw = 4 * D; g.C = b.xs; g.ldc = D; g.M = NT; g.N = D; g.K = 4 * D;
    g.bias = w->fc2_b[l]; g.gate = mod + 5 * D; g.gate_ld = ld_row; g.rows_per_batch = T; g.dtype = MH_F32;
    g.epilogue = MH_EPI_GATE_RESID; g.stats_out = fuse_ln ? b.stats : nullptr;
    weights(w->fc2_w[l], w->fc2_w3[l], g);
    MH_TRY(gemm(g, s));
  }
  const float* modf = b.cond_cur + (long)c->depth * 6 * D;
  MH_TRY(ln_modulate(b.xs, D, modf, modf + D, ld_row, T, b.xm, D, NT, D, 1e-6f, MH_F32, s));
  hipLaunchKernelGGL(dit_final_kernel, dim3(ceil_div(T, 4), N / 2), dim3(256), 0, s, b.xm, w->fin_w, D, w->fin_b, N, T, D,
                     cfg_scale, out);
  return check_launch("dit_final_kernel");
}

int ddpm_step(const float* model_out, const float* x, const float* noise, const float* coef, const int* sel,
              long noise_stride, const uint8_t* imask, const float* iref, const float* x0_override, int raw_pred, int N,
              int T, float* x_out, float* pred, hipStream_t s) {
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(ceil_div(N * 2 * T, 256)), dim3(256), 0, s, model_out, x, noise, coef, sel,
                     noise_stride, imask, iref, x0_override, raw_pred, N, T, x_out, pred);
  return check_launch("ddpm_step_kernel");
}

}  // namespace
}  // namespace mh

using namespace mh;

extern "C" int64_t mh_dit_workspace_bytes(const MhDiTConfig* c, int N, int T) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  if (!c || N <= 0 || T <= 0) return -1;
  return dit_ws_layout(c, N, T, nullptr, 0, nullptr) + align256((int64_t)N * 4 * T * 4) + cond_scratch_bytes(c, N);
}

extern "C" int64_t mh_ddpm_loop_workspace_bytes(const MhDiTConfig* c, int N, int T, int n_steps) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  if (!c || N <= 0 || T <= 0 || n_steps <= 0) return -1;
  return dit_ws_layout(c, N, T, nullptr, 0, nullptr) + align256((int64_t)N * 4 * T * 4) +
         cond_scratch_bytes(c, n_steps * N) + align256(cond_floats(c, N) * 4 * (int64_t)n_steps) +
         align256((int64_t)N * 2 * T * 4);   // x0 between the two halves of a step with sliders
}

extern "C" int mh_dit_forward_cfg(const MhDiTConfig* c, const MhDiTWeights* w, const float* x, const int32_t* t,
                                  const float* cc, const float* y, float cfg_scale, int band, int open_from, int N, int T,
                                  float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_dit(c, N, T));
  MH_REQUIRE(w && x && t && cc && y && out && workspace, "mh_dit_forward_cfg: null argument");
  MH_REQUIRE(workspace_bytes >= mh_dit_workspace_bytes(c, N, T), "mh_dit_forward_cfg: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  DiTBuf b;
  const int64_t used = dit_ws_layout(c, N, T, workspace, workspace_bytes, &b);
  void* scratch = (char*)workspace + used + align256((int64_t)N * 4 * T * 4);
  if (hipMemsetAsync(b.vt, 0, (size_t)N * c->hidden * b.Tpad * 4, s) != hipSuccess) return check_launch("memset vt");
  MH_TRY(dit_conditioning(c, w, t, 1, y, N, 1, b.cond_cur, b, scratch, s));
  MH_REQUIRE(open_from >= 0 && open_from <= T, "mh_dit_forward_cfg: open_from outside [0, T]");
  return dit_body(c, w, x, cc, cfg_scale, band, open_from, N, T, out, b, s);
}

extern "C" int mh_ddpm_step(const float* model_out, const float* x, const float* noise, const float* coef,
                            const uint8_t* inpaint_mask, const float* inpaint_ref, const float* x0_override,
                            int raw_pred, int N, int T, float* x_out, float* pred_xstart, void* stream) {
  MH_REQUIRE(model_out && x && noise && coef && x_out && N > 0 && T > 0, "mh_ddpm_step: bad argument");
  MH_REQUIRE((inpaint_mask == nullptr) == (inpaint_ref == nullptr), "mh_ddpm_step: inpaint mask/ref must come together");
  MH_REQUIRE(!raw_pred || pred_xstart, "mh_ddpm_step: raw_pred needs pred_xstart");
  return ddpm_step(model_out, x, noise, coef, nullptr, 0, inpaint_mask, inpaint_ref, x0_override, raw_pred, N, T, x_out,
                   pred_xstart, (hipStream_t)stream);
}

extern "C" int mh_ddpm_sample_loop(const MhDiTConfig* c, const MhDiTWeights* w, float* x_io, const float* cc,
                                   const float* y, float cfg_scale, int band, int open_from, int N, int T, int n_steps,
                                   const int32_t* t_map, const float* coefs, const float* noise,
                                   const uint8_t* inpaint_mask, const float* inpaint_ref, const MhSliderSet* sliders,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_dit(c, N, T));
  MH_REQUIRE(w && x_io && cc && y && t_map && coefs && noise && workspace && n_steps > 0,
             "mh_ddpm_sample_loop: null argument");
  if (sliders) MH_TRY(check_slider_set(sliders, N));
  MH_REQUIRE(open_from >= 0 && open_from <= T, "mh_ddpm_sample_loop: open_from outside [0, T]");
  MH_REQUIRE(stream != nullptr, "mh_ddpm_sample_loop: needs a non-default stream (hipGraph capture)");
  MH_REQUIRE(workspace_bytes >= mh_ddpm_loop_workspace_bytes(c, N, T, n_steps), "mh_ddpm_sample_loop: workspace too small");
  MH_REQUIRE((inpaint_mask == nullptr) == (inpaint_ref == nullptr), "mh_ddpm_sample_loop: inpaint mask/ref mismatch");
  hipStream_t s = (hipStream_t)stream;
  DiTBuf b;
  const int64_t used = dit_ws_layout(c, N, T, workspace, workspace_bytes, &b);
  float* mout = (float*)((char*)workspace + used);
  void* scratch = (char*)mout + align256((int64_t)N * 4 * T * 4);
  float* cond_all = (float*)((char*)scratch + cond_scratch_bytes(c, n_steps * N));
  float* x0buf = (float*)((char*)cond_all + align256(cond_floats(c, N) * 4 * (int64_t)n_steps));
  MH_TRY(gemm_prepare());
  // hoisted out of the per-step graph: V^T padding, the conditioning of EVERY step, the loop counter
  if (hipMemsetAsync(b.vt, 0, (size_t)N * c->hidden * b.Tpad * 4, s) != hipSuccess) return check_launch("memset vt");
  MH_TRY(dit_conditioning(c, w, t_map, N, y, N, n_steps, cond_all, b, scratch, s));
  hipLaunchKernelGGL(loop_set_kernel, dim3(1), dim3(64), 0, s, b.sel, n_steps - 1);
  MH_TRY(check_launch("loop_set_kernel"));

  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return check_launch("begin capture");
  const long per_step = cond_floats(c, N);
  hipLaunchKernelGGL(select_step_kernel, dim3((unsigned)ceil_div((int)per_step, 256)), dim3(256), 0, s, cond_all, b.sel,
                     per_step, b.cond_cur);
  int rc = check_launch("select_step_kernel");
  if (rc == MH_OK) rc = dit_body(c, w, x_io, cc, cfg_scale, band, open_from, N, T, mout, b, s);
  if (rc == MH_OK && !sliders)
    rc = ddpm_step(mout, x_io, noise, coefs, b.sel, (long)N * 2 * T, inpaint_mask, inpaint_ref, nullptr, 0, N, T, x_io,
                   nullptr, s);
  if (rc == MH_OK && sliders) {   // eps -> x0 | in-paint + slider ends | clamp, posterior mean, noise
    rc = ddpm_step(mout, x_io, noise, coefs, b.sel, (long)N * 2 * T, nullptr, nullptr, nullptr, 1, N, T, x_io, x0buf, s);
    if (rc == MH_OK) rc = slider_project(x0buf, inpaint_mask, inpaint_ref, N, T, *sliders, s);
    if (rc == MH_OK)
      rc = ddpm_step(mout, x_io, noise, coefs, b.sel, (long)N * 2 * T, nullptr, nullptr, x0buf, 0, N, T, x_io, nullptr, s);
  }
  if (rc == MH_OK) {
    hipLaunchKernelGGL(loop_dec_kernel, dim3(1), dim3(64), 0, s, b.sel);
    rc = check_launch("loop_dec_kernel");
  }
  hipError_t ce = hipStreamEndCapture(s, &graph);
  if (rc != MH_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (ce != hipSuccess || !graph) { set_error("mh_ddpm_sample_loop: capture failed: %s", hipGetErrorString(ce)); return MH_ERR_LAUNCH; }
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGraphDestroy(graph);
    return check_launch("graph instantiate");
  }
  int rc2 = MH_OK;
  for (int i = 0; i < n_steps; ++i)
    if (hipGraphLaunch(exec, s) != hipSuccess) { rc2 = check_launch("graph launch"); break; }
  (void)hipStreamSynchronize(s);
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  return rc2;
}
