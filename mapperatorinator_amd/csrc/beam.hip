// One beam-search step as ONE kernel (round 6): HF `GenerationMixin._beam_search` (third-party; the vectorised form of transformers
// >= 4.50 as `model_generate` reaches it with num_beams > 1: osuT5/osuT5/inference/processor.py:147,159; server.py:137; the timing
// pass decodes with two beams, super_timing_generator.py:28) between two decoder positions --
//     log_softmax -> [classifier-free guidance] -> the reference's processor list on LOG-PROBABILITIES (server.py:106-134:
//     MonotonicTimeShift, TimeshiftBias, (Conditional)Temperature, LookbackBias) -> + running beam scores -> the K = max(2, 1 + #eos)
//     x num_beams best continuations of the chunk -> EOS / max_length split -> next running beams -> merge of the finished
//     hypotheses -> the "can a running beam still win" heuristic
// -- beside mh_t5_step (the decoder position) and mh_t5_reorder_cache (MapperatorinatorCache.reorder_cache, inference/cache_utils.py:
// 16-20).  mapperatorinator_amd/beam.py ran these as ~40 ATen launches per token until round 5.
// One workgroup per chunk: the K best of its num_beams x V accumulated scores are found in LDS -- a 4-pass radix select of the K-th
// largest value, a deterministic compaction (ties by ascending flat index), then a bitonic sort of the K candidates only (the first
// version sorted all 8192 values: 86 us per token against ~15 us) --, the selection logic runs on that sorted list, the surviving hypotheses are copied from the IN state to the OUT state (ping-pong: every
// workgroup reads what the previous step wrote).  Greedy beams only (do_sample = 0): beam-SAMPLE draws its continuations with
// torch.multinomial / an injected sampler and stays on the host-side path.
#include <math.h>

#include "internal.hpp"

namespace mh {
namespace {

struct BeamKV { float v; int i; };

// descending by value, ties by ascending flat index (deterministic; NaN never occurs: scores are finite or -inf)
__device__ inline bool beam_before(const BeamKV& a, const BeamKV& b) { return a.v > b.v || (a.v == b.v && a.i < b.i); }

constexpr int kBeamThreads = 512;
constexpr int kBeamWaves = kBeamThreads / 64;
constexpr int kBeamMaxK = 4096;      // candidates per chunk the selection logic holds flags for

// order-preserving integer image of a float (-0.0 folded onto +0.0, so equal floats have equal images)
__device__ inline unsigned beam_okey(float x) {
  const unsigned u = __float_as_uint(x + 0.0f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// exclusive prefix sum of one int per thread over the block (wave shuffles + the 8 wave totals); `total` = the block's sum
__device__ inline int beam_excl_scan(int x, int* s_w, int& total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
  __syncthreads();
  if (lane == 63) s_w[wid] = inc;
  __syncthreads();
  int base = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kBeamWaves; ++w) { const int t = s_w[w]; base += w < wid ? t : 0; total += t; }
  return base + inc - x;
}

// block-wide maximum of a 64-bit key
__device__ inline unsigned long long beam_max_u64(unsigned long long k, unsigned long long* s_w) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)k, o, 64), hi = __shfl_xor((unsigned)(k >> 32), o, 64);
    const unsigned long long y = ((unsigned long long)hi << 32) | lo;
    k = y > k ? y : k;
  }
  __syncthreads();
  if (lane == 0) s_w[wid] = k;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kBeamWaves; ++w) { const unsigned long long y = s_w[w]; k = y > k ? y : k; }
  return k;
}

// (score, position) -> one key whose maximum is "best score, then earliest position"
__device__ inline unsigned long long beam_rank_key(float sc, int pos) {
  return ((unsigned long long)beam_okey(sc) << 32) | (unsigned)(0x7fffffff - pos);
}

__global__ __launch_bounds__(kBeamThreads) void beam_step_kernel(MhBeamStep p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[kBeamThreads / 64];
  __shared__ int s_hist[256];
  __shared__ int s_w[kBeamWaves];
  __shared__ unsigned long long s_w64[kBeamWaves];
  __shared__ unsigned s_prefix;
  __shared__ int s_remaining;
  __shared__ uint8_t s_hit[kBeamMaxK];
  __shared__ int s_sel_run[8], s_sel_fin[8];                        // selected candidate / merged-list positions (num_beams <= 8)
  __shared__ float s_run_lp[8], s_fin_sc[8];
  __shared__ int s_ltv[8];                                          // per beam: value of the last TIME_SHIFT after the last SOS (-1: none)
  __shared__ float s_temp[8];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nb = p.num_beams, V = p.V, T = p.cur_len, L = p.max_length, R = p.G * nb;
  const MhSampling& sp = p.sp;
  const int n = nb * V;
  int k_pad = 1;
  while (k_pad < p.K) k_pad <<= 1;
  float* val = reinterpret_cast<float*>(smem);                                   // [n] accumulated scores, flat index j * V + v
  BeamKV* arr = reinterpret_cast<BeamKV*>(smem + (((size_t)n * 4 + 15) & ~(size_t)15));   // [k_pad] the K best, sorted

  // ---- per-beam processor state from the beam's own sequence: MonotonicTimeShiftLogitsProcessor (logit_processors.py:136-183)
  // and the (Conditional)Temperature of the step (:47-82; row 0 of the WHOLE call picks it unless cond_per_row) ------------------
  for (int j = wid; j < nb; j += kBeamThreads / 64) {
    const int32_t* ids = p.run_in + ((long)g * nb + j) * L;
    int last_ts = -1, last_sos = -1;
    for (int i0 = 0; i0 < T; i0 += 64) {
      const int i = i0 + lane;
      const int id = i < T ? ids[i] : -1;
      const bool is_ts = i < T && id >= sp.ts_start && id < sp.ts_end;
      bool is_sos = false;
      for (int q = 0; q < sp.n_sos; ++q) is_sos |= (i < T && id == sp.sos_ids[q]);
      int a = is_ts ? i : -1, b = is_sos ? i : -1;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a = max(a, __shfl_xor(a, o, 64)); b = max(b, __shfl_xor(b, o, 64)); }
      last_ts = max(last_ts, a);
      last_sos = max(last_sos, b);
    }
    if (lane == 0) {
      s_ltv[j] = (sp.ts_end > sp.ts_start && last_ts != -1 && last_ts > last_sos) ? ids[last_ts] - sp.ts_start : -1;
      float temp = sp.temperature;
      const int32_t* hist = sp.cond_per_row ? ids : p.run_in;       // (row 0 of the call = chunk 0, beam 0)
      for (int q = 0; q < sp.n_cond; ++q) {
        const int off = sp.cond_offset[q];
        if (T >= off && (sp.tok_flags[hist[T - off]] & (2 << q))) { temp = sp.cond_temp[q]; break; }
      }
      s_temp[j] = temp;
    }
  }
  __syncthreads();

  // ---- log_softmax (+ guidance) + processors + running score -> val[] --------------------------------------------------------
  // the prompt rows' logits are staged in LDS once (one round of independent loads); the per-beam maximum / sum then read LDS in the
  // same thread -> column order as before, so the reductions keep their bits
  for (int i = tid; i < n; i += kBeamThreads) {
    const int j = i / V, v = i - j * V, r = g * nb + j;
    val[i] = p.logits[(long)(p.cfg ? R + r : r) * V + v];                 // under guidance the prompt rows are the SECOND half
  }
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    const int r = g * nb + j;
    float* lg_pos = val + j * V;
    const float* lg_neg = p.logits + (long)r * V;
    float mp = -INFINITY, mn = -INFINITY;
    for (int v = tid; v < V; v += kBeamThreads) { mp = fmaxf(mp, lg_pos[v]); if (p.cfg) mn = fmaxf(mn, lg_neg[v]); }
    mp = block_max(mp, red);
    if (p.cfg) mn = block_max(mn, red);
    float sp_ = 0.f, sn_ = 0.f;
    for (int v = tid; v < V; v += kBeamThreads) { sp_ += expf(lg_pos[v] - mp); if (p.cfg) sn_ += expf(lg_neg[v] - mn); }
    sp_ = block_sum(sp_, red);
    if (p.cfg) sn_ = block_sum(sn_, red);
    const float lse_p = logf(sp_), lse_n = p.cfg ? logf(sn_) : 0.f;
    const int ltv = s_ltv[j];
    const float temp = s_temp[j], rs = p.rs_in[r];
    for (int v = tid; v < V; v += kBeamThreads) {      // (each thread rewrites exactly the columns it read)
      float x = (lg_pos[v] - mp) - lse_p;
      if (p.cfg) {   // HF ClassifierFreeGuidanceLogitsProcessor, first in the list, on the reference's row order: second + (first - second) * scale
        const float xn = (lg_neg[v] - mn) - lse_n;
        x = __fadd_rn(x, __fmul_rn(__fsub_rn(xn, x), p.cfg_scale));
      }
      if (ltv >= 0 && v >= sp.ts_start && v < sp.ts_start + ltv) x = -INFINITY;
      if (sp.timeshift_bias != 0.f && v >= sp.ts_start && v < sp.ts_end) x += sp.timeshift_bias;
      x = x / temp;
      if (sp.lookback_mask_end > sp.ts_start && v >= sp.ts_start && v < sp.lookback_mask_end) x = -INFINITY;
      lg_pos[v] = x + rs;
    }
  }
  __syncthreads();

  // ---- the K best of the chunk's num_beams x V accumulated scores, best first, ties by ascending flat index ----------------------
  // (1) radix select of the K-th largest value on the order-preserving integer image of the floats, 8 bits per pass
  const int K = p.K;
  if (tid == 0) { s_prefix = 0u; s_remaining = K; }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const int rem = s_remaining;
    for (int i = tid; i < n; i += kBeamThreads) {
      const unsigned u = beam_okey(val[i]);
      if ((u & mask) == prefix) atomicAdd(&s_hist[(u >> shift) & 255], 1);
    }
    __syncthreads();
    // thread t owns bin 255 - t: the bin where the count from the top first reaches `rem` holds the K-th value
    const int h = tid < 256 ? s_hist[255 - tid] : 0;
    int tot;
    const int before = beam_excl_scan(h, s_w, tot);
    if (tid < 256 && before < rem && before + h >= rem) {      // exactly one thread (the candidates still in play number >= rem)
      s_remaining = rem - before;                               // how many of the elements inside this bin are still wanted
      s_prefix = prefix | ((unsigned)(255 - tid) << shift);
    }
    __syncthreads();
  }
  const unsigned kth = s_prefix;                 // integer image of the K-th largest value; s_remaining of its ties are wanted
  const int want_ties = s_remaining;
  // (2) compaction: everything above the K-th value, then the `want_ties` ties of SMALLEST flat index (threads own contiguous index
  //     ranges, so a block-wide exclusive scan of their tie counts ranks the ties in index order)
  const int chunk = (n + kBeamThreads - 1) / kBeamThreads, i_lo = tid * chunk, i_hi = (i_lo + chunk < n) ? i_lo + chunk : n;
  int above = 0, ties = 0;
  for (int i = i_lo; i < i_hi; ++i) { const unsigned u = beam_okey(val[i]); above += u > kth; ties += u == kth; }
  int n_above, n_ties;
  int wpos = beam_excl_scan(above, s_w, n_above);            // n_above = K - want_ties
  int trank = beam_excl_scan(ties, s_w, n_ties);
  for (int i = i_lo; i < i_hi; ++i) {
    const float x = val[i];
    const unsigned u = beam_okey(x);
    if (u > kth) arr[wpos++] = BeamKV{x, i};
    else if (u == kth) { if (trank < want_ties) arr[n_above + trank] = BeamKV{x, i}; ++trank; }
  }
  for (int i = K + tid; i < k_pad; i += kBeamThreads) arr[i] = BeamKV{-INFINITY, 0x7fffffff};
  __syncthreads();
  // (3) bitonic sort of the K candidates
  for (int k = 2; k <= k_pad; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int t = tid; t < k_pad / 2; t += kBeamThreads) {
        const int i = ((t / jj) * 2 * jj) + (t % jj), ixj = i + jj;
        const bool up = (i & k) == 0;                    // this block sorts "best first"
        const BeamKV a = arr[i], b = arr[ixj];
        if (up ? beam_before(b, a) : beam_before(a, b)) { arr[i] = b; arr[ixj] = a; }
      }
      __syncthreads();
    }
  }

  // ---- stopping criteria on the K best candidates (d.) ------------------------------------------------------------------------
  const bool at_max = T + 1 >= L;
  for (int k = tid; k < K; k += kBeamThreads) s_hit[k] = (at_max || p.eos_table[arr[k].i % V]) ? 1 : 0;
  __syncthreads();

  // e. the running beams of the next step: the num_beams best of run_lp = lp + hit * -1e9 (best first, ties by position).  One
  //    block-wide maximum per beam over (score, position) keys -- the serial insertion of the first version cost ~25 ns per candidate
  {
    unsigned long long prev = ~0ull;
    for (int s = 0; s < nb; ++s) {
      unsigned long long best = 0ull;
      for (int k = tid; k < K; k += kBeamThreads) {
        const float rl = arr[k].v + (s_hit[k] ? -1.0e9f : -0.0f);
        const unsigned long long key = beam_rank_key(rl, k);
        if (key < prev && key > best) best = key;
      }
      best = beam_max_u64(best, s_w64);
      prev = best;
      if (tid == 0) {
        const int k = 0x7fffffff - (int)(unsigned)(best & 0xffffffffu);
        s_sel_run[s] = k;
        s_run_lp[s] = arr[k].v + (s_hit[k] ? -1.0e9f : -0.0f);
      }
    }
  }
  // f. finished hypotheses: only candidates inside the top num_beams count; merged with the chunk's finished set by score
  {
    bool all_fin_in = true;
    for (int s = 0; s < nb; ++s) all_fin_in &= p.fin_in[g * nb + s] != 0;
    const bool full = all_fin_in && p.early_stopping == 1;
    const bool open_in = p.heuristic_open[g] != 0;
    const float div = (float)pow((double)(T + 1 - p.P), (double)p.length_penalty);
    auto merged_score = [&](int m) -> float {        // merged list: the nb old slots, then the K candidates
      if (m < nb) return p.bs_in[g * nb + m];
      const int k = m - nb;
      const bool just = s_hit[k] && k < nb;
      float sc = arr[k].v / div;
      sc = sc + (full ? -1.0e9f : -0.0f);
      sc = sc + (!open_in ? -1.0e9f : -0.0f);
      sc = sc + (!just ? -1.0e9f : -0.0f);
      return sc;
    };
    unsigned long long prev = ~0ull;
    for (int s = 0; s < nb; ++s) {
      unsigned long long best = 0ull;
      for (int m = tid; m < nb + K; m += kBeamThreads) {
        const unsigned long long key = beam_rank_key(merged_score(m), m);
        if (key < prev && key > best) best = key;
      }
      best = beam_max_u64(best, s_w64);
      prev = best;
      if (tid == 0) {
        const int m = 0x7fffffff - (int)(unsigned)(best & 0xffffffffu);
        s_sel_fin[s] = m;
        s_fin_sc[s] = merged_score(m);
      }
    }
  }
  __syncthreads();

  // ---- write the OUT state: rows are copied from the IN state (parents' running rows / old finished slots); every element of the
  // four arrays is one independent load -> store -------------------------------------------------------------------------------
  const int n_new = L - p.P;
  for (int e = tid; e < nb * L; e += kBeamThreads) {
    const int s = e / L, i = e - s * L;
    const int flat = arr[s_sel_run[s]].i, parent = flat / V, tok = flat % V;
    const int a = i == T ? tok : p.run_in[((long)g * nb + parent) * L + i];
    const int m = s_sel_fin[s];
    int b;
    if (m < nb) b = p.seq_in[((long)g * nb + m) * L + i];
    else {
      const int flat2 = arr[m - nb].i, par2 = flat2 / V, tok2 = flat2 % V;
      b = i == T ? tok2 : p.run_in[((long)g * nb + par2) * L + i];
    }
    p.run_out[((long)g * nb + s) * L + i] = a;
    p.seq_out[((long)g * nb + s) * L + i] = b;
  }
  for (int e = tid; e < nb * n_new; e += kBeamThreads) {
    const int s = e / n_new, i = e - s * n_new;
    const int parent = arr[s_sel_run[s]].i / V;
    const int a = i == T - p.P ? parent + g * nb : p.rb_in[((long)g * nb + parent) * n_new + i];
    const int m = s_sel_fin[s];
    int b;
    if (m < nb) b = p.bb_in[((long)g * nb + m) * n_new + i];
    else {
      const int par2 = arr[m - nb].i / V;
      b = i == T - p.P ? par2 + g * nb : p.rb_in[((long)g * nb + par2) * n_new + i];
    }
    p.rb_out[((long)g * nb + s) * n_new + i] = a;
    p.bb_out[((long)g * nb + s) * n_new + i] = b;
  }
  if (tid < nb) {
    const int s = tid;
    const int flat = arr[s_sel_run[s]].i, parent = flat / V, tok = flat % V;
    p.rs_out[g * nb + s] = s_run_lp[s];
    p.src[g * nb + s] = parent + g * nb;             // g. the cache rows follow the beams that keep running
    p.last[g * nb + s] = tok;
    if (p.cfg) {   // `beam_idx.repeat(2)` (cache_utils.py:18): BOTH halves gather from the first half, and both are fed the beam's token
      p.src[R + g * nb + s] = parent + g * nb;
      p.last[R + g * nb + s] = tok;
    }
    const int m = s_sel_fin[s];
    p.fin_out[g * nb + s] = m < nb ? p.fin_in[g * nb + m] : ((s_hit[m - nb] && (m - nb) < nb) ? 1 : 0);
    p.bs_out[g * nb + s] = s_fin_sc[s];
  }

  // ---- "can a running beam still beat the worst finished one" (early_stopping = False heuristic) + the flags the host polls ----
  int my_hit = 1;
  for (int k = tid; k < K; k += kBeamThreads) my_hit &= s_hit[k] != 0;
  const bool all_hit = __syncthreads_and(my_hit) != 0;
  if (tid == 0) {
    const int hyp_len = (p.early_stopping == 2 && p.length_penalty > 0.f) ? L - p.P : T + 1 - p.P;
    const float best_running = s_run_lp[0] / (float)pow((double)hyp_len, (double)p.length_penalty);
    float mn = INFINITY;
    bool all_fin = true;
    for (int s = 0; s < nb; ++s) mn = fminf(mn, s_fin_sc[s]);
    bool any = false;
    for (int s = 0; s < nb; ++s) {
      const int m = s_sel_fin[s];
      const bool f = m < nb ? p.fin_in[g * nb + m] != 0 : (s_hit[m - nb] && (m - nb) < nb);
      all_fin &= f;
      any |= best_running > (f ? mn : -1.0e9f);
    }
    const bool open = (p.heuristic_open[g] != 0) && any;
    p.heuristic_open[g] = open ? 1 : 0;
    p.flags[g * 3 + 0] = open; p.flags[g * 3 + 1] = all_hit; p.flags[g * 3 + 2] = all_fin;
  }
}

}  // namespace
}  // namespace mh

using namespace mh;

extern "C" int64_t mh_beam_step_lds_bytes(int num_beams, int V) {      // the score array; mh_beam_step adds the K candidates itself
  if (num_beams < 1 || V < 1) return -1;
  return (((int64_t)num_beams * V * 4 + 15) & ~(int64_t)15);
}

extern "C" int mh_beam_step(const MhBeamStep* bs, void* stream) {
  MH_REQUIRE(bs && bs->logits && bs->eos_table && bs->run_in && bs->run_out && bs->rs_in && bs->rs_out && bs->rb_in && bs->rb_out &&
             bs->seq_in && bs->seq_out && bs->bs_in && bs->bs_out && bs->bb_in && bs->bb_out && bs->fin_in && bs->fin_out &&
             bs->heuristic_open && bs->src && bs->last && bs->flags, "mh_beam_step: null argument");
  MH_REQUIRE(bs->G >= 1 && bs->num_beams >= 2 && bs->num_beams <= 8, "mh_beam_step: %d chunks x %d beams (2 .. 8 beams)", bs->G, bs->num_beams);
  MH_REQUIRE(bs->K >= bs->num_beams && bs->K <= kBeamMaxK && bs->K <= bs->num_beams * bs->V, "mh_beam_step: K = %d candidates not in [num_beams, %d]", bs->K, kBeamMaxK);
  MH_REQUIRE(bs->P >= 1 && bs->cur_len >= bs->P && bs->cur_len < bs->max_length, "mh_beam_step: cur_len %d not in [P, max_length)", bs->cur_len);
  MH_REQUIRE(bs->sp.do_sample == 0, "mh_beam_step: greedy beams only (beam-sample draws on the host side)");
  MH_REQUIRE(!(bs->sp.lookback_types_first && bs->sp.lookback_mask_end > bs->sp.ts_start), "mh_beam_step: the types_first lookback renormalisation is not built for beams");
  MH_REQUIRE(bs->sp.tok_flags || bs->sp.n_cond == 0, "mh_beam_step: conditional temperature needs tok_flags");
  MH_REQUIRE(bs->sp.temperature > 0.f && bs->sp.n_sos >= 0 && bs->sp.n_sos <= 16 && bs->sp.n_cond >= 0 && bs->sp.n_cond <= 3, "mh_beam_step: bad sampling parameters");
  int64_t k_pad = 1;
  while (k_pad < bs->K) k_pad <<= 1;
  const int64_t lds = mh_beam_step_lds_bytes(bs->num_beams, bs->V) + k_pad * 8;
  MH_REQUIRE(lds > 0 && lds <= 120 * 1024, "mh_beam_step: num_beams x V = %d x %d (+ %d candidates) does not fit 120 KB of LDS", bs->num_beams, bs->V, bs->K);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(beam_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess)
      return check_launch("mh_beam_step: LDS attribute");
    attr_set = true;
  }
  hipLaunchKernelGGL(beam_step_kernel, dim3(bs->G), dim3(kBeamThreads), (size_t)lds, (hipStream_t)stream, *bs);
  return check_launch("beam_step_kernel");
}
