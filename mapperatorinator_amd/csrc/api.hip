// Error plumbing + version of libmapperhip's C ABI (include/mapperhip.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <new>

#include "common.hpp"

namespace mh {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MH_ERR_LAUNCH;
  }
  return MH_OK;
}

// ---- tuning options: a default, an environment override read at first use, and mh_set_option() at run time --------
// (process-wide, not per call: they select between kernels that compute the same results)
// (atomics: the decode chains' launcher threads read options while a caller's thread may set one -- a torn or stale read is
// excluded; two engines that want DIFFERENT variants in one process still share the switch: options are process-wide)
struct OptionSlot { const char* name; const char* env; std::atomic<long> value; std::atomic<bool> resolved; };
static OptionSlot g_options[OPT_COUNT] = {
    {"gemm_splitk_tiles", "MH_GEMM_SPLITK_TILES", 192, false},   // below this many 32x32 tiles the 16x16 split-K tile is used (0 = never)
    {"decode_chains", "MH_DECODE_CHAINS", 0, false},             // independent row chains of the decode step (0 = automatic)
    {"decode_prefill", "MH_DECODE_PREFILL", 1, false},           // 1: batched prompt prefill, 0: feed the prompt token by token
    {"decode_gemv_cols", "MH_DECODE_GEMV_COLS", 0, false},       // valid weight rows per 16-column MFMA tile of the decode GEMVs (0 = automatic)
    {"decode_fused_proj", "MH_DECODE_FUSED_PROJ", 1, false},     // 1: attention kernels project their own q / k / v, 0: stand-alone GEMVs
    {"gemm_tile128_min", "MH_GEMM_TILE128_MIN", 192, false},     // the 128x128 GEMM tile is used from this many tiles on (else 64x64 / smaller)
    {"dit_split3_min_rows", "MH_DIT_SPLIT3_MIN_ROWS", 2048, false},   // DiT denoiser batches of >= this many rows (N*T) run their big GEMMs as bf16 x 3 (0 = never)
    {"gemm_glds", "MH_GEMM_GLDS", 3, false},                     // bf16 GEMM operands by LDS-DMA (global_load_lds): 3 = three-stage kernel, 256x128 tiles or 128x128 where fewer than 128 of the big ones exist; 2 = 256x128 only; 1 = two-stage 128x128 only; 0 = register staging
    {"decode_launch_threads", "MH_DECODE_LAUNCH_THREADS", 1, false},   // 1: one host launcher thread per decode chain (graph replay costs ~0.4 ms of host time per step); 0: one thread feeds all chains round robin -- for profilers whose counter passes do not survive concurrent launcher threads (rocprofv3 --pmc)
    {"decode_graph_cache", "MH_DECODE_GRAPH_CACHE", 1, false},   // 1: instantiated step graphs are kept across mh_t5_generate calls (LRU of 16, exact-description match); 0: captured per call
    {"gemm_tile256sq_min", "MH_GEMM_TILE256SQ_MIN", 440, false},   // bf16 GEMM: the 256 x 256 tile (two LDS stages, 128 x 64 wave tiles, AGPR accumulators) from this many tiles on, if its rounds of 256 workgroups are >= 88 % full (0 = never, 1 = whenever the three-stage kernel would run: tests)
    {"gemm_2stage_max_k", "MH_GEMM_2STAGE_MAX_K", 512, false},   // bf16 GEMM with K <= this: the two-stage 128 x 128 kernel (64 KB of LDS: two workgroups per CU) instead of the three-stage forms (0 = never).  Batched DiT-S bf16 (K = 384 on three of four projections): 130.3 -> 121.6 ms per 100 steps; at 1024 DiT-B (K = 768) 284.6 -> 287.7, at 4096 299.7
    {"dit_skinny_max_rows", "MH_DIT_SKINNY_MAX_ROWS", 512, false},   // fp32-semantics DiT with at most this many rows (N T: one chunk = 256): the four block GEMMs as one-round-trip 16 x 16 latency kernels with the LayerNorm taken from registers (dit.hip dit_skinny_kernel); 0 = the LDS-tiled GEMMs (different fp32 summation order)
};

static thread_local const MhOptionSet* tl_option_set = nullptr;

}  // namespace mh

struct MhOptionSet {
  long value[mh::OPT_COUNT];
  bool has[mh::OPT_COUNT];
};

namespace mh {

OptionScope::OptionScope(const MhOptionSet* set) : prev(tl_option_set) { tl_option_set = set; }
OptionScope::~OptionScope() { tl_option_set = prev; }
const MhOptionSet* current_option_set() { return tl_option_set; }

long option(int id) {
  if (tl_option_set && tl_option_set->has[id]) return tl_option_set->value[id];
  OptionSlot& o = g_options[id];
  if (!o.resolved.load(std::memory_order_acquire)) {
    const char* e = getenv(o.env);
    if (e && *e) o.value.store(atol(e), std::memory_order_relaxed);
    o.resolved.store(true, std::memory_order_release);
  }
  return o.value.load(std::memory_order_relaxed);
}

}  // namespace mh

extern "C" int mh_set_option(const char* name, long value) {
  for (int i = 0; name && i < mh::OPT_COUNT; ++i)
    if (strcmp(mh::g_options[i].name, name) == 0) {
      mh::g_options[i].value.store(value, std::memory_order_relaxed);
      mh::g_options[i].resolved.store(true, std::memory_order_release);
      return MH_OK;
    }
  mh::set_error("mh_set_option: unknown option '%s'", name ? name : "(null)");
  return MH_ERR_ARG;
}

extern "C" long mh_get_option(const char* name) {
  for (int i = 0; name && i < mh::OPT_COUNT; ++i)
    if (strcmp(mh::g_options[i].name, name) == 0) return mh::option(i);
  mh::set_error("mh_get_option: unknown option '%s'", name ? name : "(null)");
  return -1;
}

static int option_index(const char* name) {
  for (int i = 0; name && i < mh::OPT_COUNT; ++i)
    if (strcmp(mh::g_options[i].name, name) == 0) return i;
  return -1;
}

extern "C" MhOptionSet* mh_options_create(void) { return new (std::nothrow) MhOptionSet(); }   // value-initialised: no overrides
extern "C" void mh_options_destroy(MhOptionSet* set) { delete set; }

extern "C" int mh_options_set(MhOptionSet* set, const char* name, long value) {
  const int i = option_index(name);
  if (!set || i < 0) { mh::set_error("mh_options_set: %s '%s'", set ? "unknown option" : "null set,", name ? name : "(null)"); return MH_ERR_ARG; }
  set->value[i] = value;
  set->has[i] = true;
  return MH_OK;
}

extern "C" int mh_options_clear(MhOptionSet* set, const char* name) {
  if (!set) { mh::set_error("mh_options_clear: null set"); return MH_ERR_ARG; }
  if (!name) { for (int i = 0; i < mh::OPT_COUNT; ++i) set->has[i] = false; return MH_OK; }
  const int i = option_index(name);
  if (i < 0) { mh::set_error("mh_options_clear: unknown option '%s'", name); return MH_ERR_ARG; }
  set->has[i] = false;
  return MH_OK;
}

extern "C" long mh_options_get(const MhOptionSet* set, const char* name) {
  const int i = option_index(name);
  if (i < 0) { mh::set_error("mh_options_get: unknown option '%s'", name ? name : "(null)"); return -1; }
  mh::OptionScope sc(set);
  return mh::option(i);
}

extern "C" const char* mh_last_error(void) { return mh::g_err; }
extern "C" int mh_abi_version(void) { return MH_ABI_VERSION; }

extern "C" int mh_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(MhGemm);
    case 1: return (int)sizeof(MhT5Config);
    case 2: return (int)sizeof(MhT5Weights);
    case 3: return (int)sizeof(MhSampling);
    case 4: return (int)sizeof(MhDiTConfig);
    case 5: return (int)sizeof(MhDiTWeights);
    case 6: return (int)sizeof(MhSliderSet);
    case 7: return (int)sizeof(MhBeamStep);
  }
  return -1;
}
