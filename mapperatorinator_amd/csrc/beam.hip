// One beam-search step as ONE kernel (round 6): HF `GenerationMixin._beam_search` (third-party; the vectorised form of transformers
// >= 4.50 as `model_generate` reaches it with num_beams > 1: osuT5/osuT5/inference/processor.py:147,159; server.py:137; the timing
// pass decodes with two beams, super_timing_generator.py:28) between two decoder positions --
//     log_softmax -> [classifier-free guidance] -> the reference's processor list on LOG-PROBABILITIES (server.py:106-134:
//     MonotonicTimeShift, TimeshiftBias, (Conditional)Temperature, LookbackBias) -> + running beam scores -> the K = max(2, 1 + #eos)
//     x num_beams best continuations of the chunk -> EOS / max_length split -> next running beams -> merge of the finished
//     hypotheses -> the "can a running beam still win" heuristic
// -- beside mh_t5_step (the decoder position) and mh_t5_reorder_cache (MapperatorinatorCache.reorder_cache, inference/cache_utils.py:
// 16-20).  mapperatorinator_amd/beam.py ran these as ~40 ATen launches per token until round 5.
// One workgroup per chunk: its num_beams x V accumulated scores are sorted in LDS (bitonic, descending, ties by flat index), the
// selection logic runs on the sorted list, the surviving hypotheses are copied from the IN state to the OUT state (ping-pong: every
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
constexpr int kBeamMaxK = 4096;      // candidates per chunk the selection logic holds flags for

__global__ __launch_bounds__(kBeamThreads) void beam_step_kernel(MhBeamStep p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  BeamKV* arr = reinterpret_cast<BeamKV*>(smem);                    // [n_pad]
  __shared__ float red[kBeamThreads / 64];
  __shared__ uint8_t s_hit[kBeamMaxK];
  __shared__ int s_sel_run[8], s_sel_fin[8];                        // selected candidate / merged-list positions (num_beams <= 8)
  __shared__ float s_run_lp[8], s_fin_sc[8];
  __shared__ int s_ltv[8];                                          // per beam: value of the last TIME_SHIFT after the last SOS (-1: none)
  __shared__ float s_temp[8];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nb = p.num_beams, V = p.V, T = p.cur_len, L = p.max_length, R = p.G * nb;
  const MhSampling& sp = p.sp;
  int n_pad = 1;
  while (n_pad < nb * V) n_pad <<= 1;

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

  // ---- log_softmax (+ guidance) + processors + running score -> the sort array --------------------------------------------
  for (int j = 0; j < nb; ++j) {
    const int r = g * nb + j;
    const float* lg_pos = p.logits + (long)(p.cfg ? R + r : r) * V;       // under guidance the prompt rows are the SECOND half
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
    for (int v = tid; v < V; v += kBeamThreads) {
      float x = (lg_pos[v] - mp) - lse_p;
      if (p.cfg) {   // HF ClassifierFreeGuidanceLogitsProcessor, first in the list, on the reference's row order: second + (first - second) * scale
        const float xn = (lg_neg[v] - mn) - lse_n;
        x = __fadd_rn(x, __fmul_rn(__fsub_rn(xn, x), p.cfg_scale));
      }
      if (ltv >= 0 && v >= sp.ts_start && v < sp.ts_start + ltv) x = -INFINITY;
      if (sp.timeshift_bias != 0.f && v >= sp.ts_start && v < sp.ts_end) x += sp.timeshift_bias;
      x = x / temp;
      if (sp.lookback_mask_end > sp.ts_start && v >= sp.ts_start && v < sp.lookback_mask_end) x = -INFINITY;
      arr[j * V + v] = BeamKV{x + rs, j * V + v};
    }
  }
  for (int i = nb * V + tid; i < n_pad; i += kBeamThreads) arr[i] = BeamKV{-INFINITY, 0x7fffffff};
  __syncthreads();

  // ---- bitonic sort of the chunk's num_beams x V accumulated scores, best first -------------------------------------------------
  for (int k = 2; k <= n_pad; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int t = tid; t < n_pad / 2; t += kBeamThreads) {
        const int i = ((t / jj) * 2 * jj) + (t % jj), ixj = i + jj;
        const bool up = (i & k) == 0;                    // this block sorts "best first"
        const BeamKV a = arr[i], b = arr[ixj];
        if (up ? beam_before(b, a) : beam_before(a, b)) { arr[i] = b; arr[ixj] = a; }
      }
      __syncthreads();
    }
  }

  // ---- stopping criteria on the K best candidates (d.) ------------------------------------------------------------------------
  const int K = p.K;
  const bool at_max = T + 1 >= L;
  for (int k = tid; k < K; k += kBeamThreads) s_hit[k] = (at_max || p.eos_table[arr[k].i % V]) ? 1 : 0;
  __syncthreads();

  if (tid == 0) {
    // e. the running beams of the next step: the num_beams best of run_lp = lp + hit * -1e9 (best first, ties by position)
    for (int s = 0; s < nb; ++s) { s_sel_run[s] = -1; s_run_lp[s] = 0.f; }
    for (int k = 0; k < K; ++k) {
      const float rl = arr[k].v + (s_hit[k] ? -1.0e9f : -0.0f);
      int pos = nb;
      while (pos > 0 && (s_sel_run[pos - 1] < 0 || rl > s_run_lp[pos - 1])) --pos;
      if (pos < nb) {
        for (int s = nb - 1; s > pos; --s) { s_sel_run[s] = s_sel_run[s - 1]; s_run_lp[s] = s_run_lp[s - 1]; }
        s_sel_run[pos] = k; s_run_lp[pos] = rl;
      }
    }
    // f. finished hypotheses: only candidates inside the top num_beams count; merged with the chunk's finished set by score
    bool all_fin_in = true;
    for (int s = 0; s < nb; ++s) all_fin_in &= p.fin_in[g * nb + s] != 0;
    const bool full = all_fin_in && p.early_stopping == 1;
    const bool open_in = p.heuristic_open[g] != 0;
    const float div = (float)pow((double)(T + 1 - p.P), (double)p.length_penalty);
    for (int s = 0; s < nb; ++s) { s_sel_fin[s] = -1; s_fin_sc[s] = 0.f; }
    for (int m = 0; m < nb + K; ++m) {        // merged list: the nb old slots, then the K candidates
      float sc;
      if (m < nb) sc = p.bs_in[g * nb + m];
      else {
        const int k = m - nb;
        const bool just = s_hit[k] && k < nb;
        sc = arr[k].v / div;
        sc = sc + (full ? -1.0e9f : -0.0f);
        sc = sc + (!open_in ? -1.0e9f : -0.0f);
        sc = sc + (!just ? -1.0e9f : -0.0f);
      }
      int pos = nb;
      while (pos > 0 && (s_sel_fin[pos - 1] < 0 || sc > s_fin_sc[pos - 1])) --pos;
      if (pos < nb) {
        for (int s = nb - 1; s > pos; --s) { s_sel_fin[s] = s_sel_fin[s - 1]; s_fin_sc[s] = s_fin_sc[s - 1]; }
        s_sel_fin[pos] = m; s_fin_sc[pos] = sc;
      }
    }
  }
  __syncthreads();

  // ---- write the OUT state: rows are copied from the IN state (parents' running rows / old finished slots) ---------------------
  const int n_new = L - p.P;
  for (int s = 0; s < nb; ++s) {
    // running beam s <- candidate s_sel_run[s]
    const int k = s_sel_run[s];
    const int flat = arr[k].i, parent = flat / V, tok = flat % V;
    const int32_t* src_seq = p.run_in + ((long)g * nb + parent) * L;
    const int32_t* src_bi = p.rb_in + ((long)g * nb + parent) * n_new;
    int32_t* dst_seq = p.run_out + ((long)g * nb + s) * L;
    int32_t* dst_bi = p.rb_out + ((long)g * nb + s) * n_new;
    for (int i = tid; i < L; i += kBeamThreads) dst_seq[i] = i == T ? tok : src_seq[i];
    for (int i = tid; i < n_new; i += kBeamThreads) dst_bi[i] = i == T - p.P ? parent + g * nb : src_bi[i];
    if (tid == 0) {
      p.rs_out[g * nb + s] = s_run_lp[s];
      p.src[g * nb + s] = parent + g * nb;             // g. the cache rows follow the beams that keep running
      p.last[g * nb + s] = tok;
      if (p.cfg) {   // `beam_idx.repeat(2)` (cache_utils.py:18): BOTH halves gather from the first half, and both are fed the beam's token
        p.src[R + g * nb + s] = parent + g * nb;
        p.last[R + g * nb + s] = tok;
      }
    }
    // finished slot s <- merged entry s_sel_fin[s]
    const int m = s_sel_fin[s];
    int32_t* fs = p.seq_out + ((long)g * nb + s) * L;
    int32_t* fb = p.bb_out + ((long)g * nb + s) * n_new;
    if (m < nb) {
      const int32_t* os = p.seq_in + ((long)g * nb + m) * L;
      const int32_t* ob = p.bb_in + ((long)g * nb + m) * n_new;
      for (int i = tid; i < L; i += kBeamThreads) fs[i] = os[i];
      for (int i = tid; i < n_new; i += kBeamThreads) fb[i] = ob[i];
      if (tid == 0) p.fin_out[g * nb + s] = p.fin_in[g * nb + m];
    } else {
      const int k2 = m - nb, flat2 = arr[k2].i, par2 = flat2 / V, tok2 = flat2 % V;
      const int32_t* os = p.run_in + ((long)g * nb + par2) * L;
      const int32_t* ob = p.rb_in + ((long)g * nb + par2) * n_new;
      for (int i = tid; i < L; i += kBeamThreads) fs[i] = i == T ? tok2 : os[i];
      for (int i = tid; i < n_new; i += kBeamThreads) fb[i] = i == T - p.P ? par2 + g * nb : ob[i];
      if (tid == 0) p.fin_out[g * nb + s] = (s_hit[k2] && k2 < nb) ? 1 : 0;
    }
    if (tid == 0) p.bs_out[g * nb + s] = s_fin_sc[s];
  }
  __syncthreads();

  // ---- "can a running beam still beat the worst finished one" (early_stopping = False heuristic) + the flags the host polls ----
  if (tid == 0) {
    const int hyp_len = (p.early_stopping == 2 && p.length_penalty > 0.f) ? L - p.P : T + 1 - p.P;
    const float best_running = s_run_lp[0] / (float)pow((double)hyp_len, (double)p.length_penalty);
    float mn = INFINITY;
    bool all_fin = true, all_hit = true;
    for (int s = 0; s < nb; ++s) mn = fminf(mn, s_fin_sc[s]);
    bool any = false;
    for (int s = 0; s < nb; ++s) {
      const int m = s_sel_fin[s];
      const bool f = m < nb ? p.fin_in[g * nb + m] != 0 : (s_hit[m - nb] && (m - nb) < nb);
      all_fin &= f;
      any |= best_running > (f ? mn : -1.0e9f);
    }
    for (int k = 0; k < K; ++k) all_hit &= s_hit[k] != 0;
    const bool open = (p.heuristic_open[g] != 0) && any;
    p.heuristic_open[g] = open ? 1 : 0;
    p.flags[g * 3 + 0] = open; p.flags[g * 3 + 1] = all_hit; p.flags[g * 3 + 2] = all_fin;
  }
}

}  // namespace
}  // namespace mh

using namespace mh;

extern "C" int64_t mh_beam_step_lds_bytes(int num_beams, int V) {
  if (num_beams < 1 || V < 1) return -1;
  int64_t n = 1;
  while (n < (int64_t)num_beams * V) n <<= 1;
  return n * 8;
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
  const int64_t lds = mh_beam_step_lds_bytes(bs->num_beams, bs->V);
  MH_REQUIRE(lds > 0 && lds <= 128 * 1024, "mh_beam_step: num_beams x V = %d x %d does not fit the 128 KB sort buffer", bs->num_beams, bs->V);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(beam_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
      return check_launch("mh_beam_step: LDS attribute");
    attr_set = true;
  }
  hipLaunchKernelGGL(beam_step_kernel, dim3(bs->G), dim3(kBeamThreads), (size_t)lds, (hipStream_t)stream, *bs);
  return check_launch("beam_step_kernel");
}
