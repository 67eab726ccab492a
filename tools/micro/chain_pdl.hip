// Micro-benchmark: what does "dependent-launch overlap" buy a decode chain on MI355X?
// VERDICT r3 item 2 asks for a structural change of the 74-launch token step.  Before rebuilding the product kernels this
// probe replays SYNTHETIC chains with the step's launch geometry and memory behaviour (tools/micro/chain_overlap.hip) in two
// forms and prices the difference, alone and beside a second chain:
//   plain   : 12 x 6 dependent kernels on one stream, replayed as a hipGraph (what mh_t5_generate does today);
//   overlap : the same kernels, but kernel k + 1 is launched BEFORE kernel k has finished -- even kernels on stream E, odd ones on
//             stream O of the chain (one captured graph with two parallel branches) -- and the dependence is carried by a
//             device-side progress word: a kernel first requests everything that does NOT depend on its predecessor (its
//             weights; the streaming kernel its first K/V tiles), then one lane polls the progress word (relaxed agent-scope
//             loads + s_sleep, bounded), then it loads the activations the predecessor wrote (sc1 loads: the CU's L1 was
//             filled while the producer was still running), works, stores write-through (sc1), waits vmcnt(0) and arrives
//             on a ticket sharded by XCD (blockIdx & 7); the last arriver publishes the new progress value.
// What the overlap form removes from the critical path: the dispatch of kernel k + 1 (~1.65 us) and the first round trip for its
// weights; what it adds: the fan-in + poll (~1 us) and the resources its spinning workgroups hold.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

struct Dep {
  unsigned* progress;     // chain progress word
  unsigned* tickets;      // this kernel slot's 9 counters (8 XCD shards + top), 32 words apart
  const unsigned* epoch;  // step counter of the chain (bumped by the step's last node)
  unsigned slot;          // this kernel waits for progress >= epoch * 128 + slot and publishes epoch * 128 + slot + 1
  unsigned nwg, prev_nwg;
  unsigned* err;
  int on;                 // 0: plain form (stream order carries the dependence)
  int proto, sleep;
};

__device__ inline unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Arrival counters: 8 monotonic words per kernel slot, one per XCD shard (blockIdx & 7), each in a cache line of its own.  A
// producer workgroup arrives with ONE non-returning agent-scope add behind its vmcnt(0) -- nobody waits for an atomic round trip;
// a consumer's poll is one wave-wide load of the 8 words: done when every shard has reached (epoch + 1) x its workgroup count.
// `tickets` of the Dep struct points at the PRODUCER slot's 8 x 32 words for dep_wait and at the kernel's own for dep_signal.
__device__ inline void nap(int s) { if (s >= 16) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(2); }
__device__ inline unsigned dep_epoch(const Dep& d) { return d.on ? ld_relaxed(d.epoch) : 0u; }
// proto 0: returning tickets (8 shards -> top), the last arriver publishes the chain's progress word, consumers poll that word
// proto 1: monotonic non-returning shard counters, EVERY consumer workgroup polls the 8 counter lines
// proto 2: the same counters, polled by workgroup 0 of the consumer only, which publishes the progress word for the others
__device__ inline bool counters_reached(const Dep& d, unsigned epoch, unsigned lane) {
  const unsigned* t = d.tickets - 9 * 32 + (lane & 7) * 32;       // the predecessor's counters
  const unsigned per = (d.prev_nwg + 7u - (lane & 7)) / 8u;
  const unsigned v = ld_relaxed(t);
  const bool ok = (int)(v - (epoch + 1u) * per) >= 0;
  return __builtin_amdgcn_ballot_w64(lane < 8 ? !ok : false) == 0;
}
__device__ inline void dep_wait(const Dep& d, unsigned epoch) {
  if (!d.on) return;
  if (threadIdx.x < 64 && d.slot > 0) {
    const unsigned lane = threadIdx.x;
    const unsigned want = epoch * 128u + d.slot;
    int n = 0;
    if (d.proto == 1 || (d.proto == 2 && blockIdx.x == 0)) {
      while (!counters_reached(d, epoch, lane)) { nap(d.sleep); if (++n > (1 << 14)) { if (lane == 0) atomicAdd(d.err, 1u); break; } }
      if (d.proto == 2 && lane == 0) __hip_atomic_store(d.progress, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (lane == 0) {
      while ((int)(ld_relaxed(d.progress) - want) < 0) { nap(d.sleep); if (++n > (1 << 14)) { atomicAdd(d.err, 1u); break; } }
    }
  }
  __syncthreads();
}
__device__ inline void dep_signal(const Dep& d, unsigned epoch) {
  if (!d.on) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's write-through stores have left
  __syncthreads();
  if (threadIdx.x != 0) return;
  unsigned* t = d.tickets + (blockIdx.x & 7u) * 32;
  if (d.proto != 0) { __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }   // (result unused: no return)
  const unsigned shard = blockIdx.x & 7u;
  const unsigned per = (d.nwg + 7u - shard) / 8u;
  if (__hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == per - 1) {
    __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned shards = d.nwg < 8u ? d.nwg : 8u;
    unsigned* top = d.tickets + 8 * 32;
    if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1) {
      __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(d.progress, epoch * 128u + d.slot + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__device__ inline u32x4_t ld16_dep(const void* p, int on) {       // data the predecessor wrote: L1-bypassing when it may still have been running
  u32x4_t v;
  if (on) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else v = *reinterpret_cast<const u32x4_t*>(p);
  return v;
}
__device__ inline void st16_wt(void* p, u32x4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }

// GEMV-like (o / co / wi / wo): 192 workgroups x 256 threads; WL x 16 B of private weights per thread (12 KB per workgroup at
// WL = 3), then the activations every workgroup reads (6 x 16 B per thread = 24.5 KB, written by the predecessor), LDS reduce,
// 512 B of write-through output per workgroup
template <int WL>
__global__ __launch_bounds__(256) void k_gemv(const uint4* __restrict__ w, const u32x4_t* act_in, u32x4_t* act_out, Dep d) {
  const uint4* p = w + (long)blockIdx.x * WL * 256 + threadIdx.x;
  uint4 v[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) v[i] = p[i * 256];                  // independent of the predecessor: requested before the wait
  const unsigned ep_ = dep_epoch(d);
  dep_wait(d, ep_);
  u32x4_t a[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) a[i] = ld16_dep(act_in + i * 256 + threadIdx.x, d.on);
  if (d.on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < WL; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
#pragma unroll
  for (int i = 0; i < 6; ++i) acc += a[i][0] ^ a[i][1] ^ a[i][2] ^ a[i][3];
  __shared__ unsigned red[4];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x < 32) {
    const unsigned r = (red[0] + red[1] + red[2] + red[3]) & 1u;
    st16_wt(act_out + (blockIdx.x * 32 + threadIdx.x) % 1536, u32x4_t{r, r, r, r});
  }
  dep_signal(d, ep_);
}

// self-attention-like: 192 x 1024 threads; 18 x 16 B of weights per thread (295 KB per workgroup, shared by the workgroups of a
// head), the residual row (dependent), two LDS reductions, 128 B of output per workgroup
__global__ __launch_bounds__(1024) void k_heavy(const uint4* __restrict__ w, const u32x4_t* act_in, u32x4_t* act_out, int heads, Dep d) {
  const int head = blockIdx.x % heads;
  const uint4* p = w + (long)head * 18 * 1024 + threadIdx.x;
  uint4 v[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) v[i] = p[i * 1024];
  const unsigned ep_ = dep_epoch(d);
  dep_wait(d, ep_);
  const u32x4_t r4 = ld16_dep(act_in + (blockIdx.x / heads) * 48 + (threadIdx.x % 48), d.on);
  if (d.on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __shared__ float red[16];
  float s = __uint_as_float(r4[0]) * __uint_as_float(r4[1]);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float tot = 0.f;
  for (int i = 0; i < 16; ++i) tot += red[i];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 18; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (float)(acc & 1) + tot;
  __syncthreads();
  if (threadIdx.x < 8) { const unsigned r = (unsigned)red[3] & 1u; st16_wt(act_out + (blockIdx.x * 8 + threadIdx.x) % 1536, u32x4_t{r, r, r, r}); }
  dep_signal(d, ep_);
}

// cross-attention-like: 192 x 1024 threads; 6 x 16 B of query-projection weights per thread (98 KB), the residual row (dependent),
// then `iters` x 2 x 16 B per thread of the K/V stream (non-temporal; independent of the predecessor -- the first PF tiles are
// requested before the wait), online softmax flops, LDS merge, 128 B of output
template <int PF>
__global__ __launch_bounds__(1024) void k_stream(const uint4* __restrict__ wq, const u32x4_t* __restrict__ kv, long wg_stride16, int iters,
                                                 const u32x4_t* act_in, u32x4_t* act_out, Dep d) {
  const u32x4_t* pk = kv + (long)blockIdx.x * wg_stride16 + threadIdx.x;
  const u32x4_t* pv = pk + (long)iters * 1024;
  const uint4* pw = wq + (long)(blockIdx.x % 12) * 6 * 1024 + threadIdx.x;
  uint4 wv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) wv[i] = pw[i * 1024];
  u32x4_t kf[PF > 0 ? PF : 1], vf[PF > 0 ? PF : 1];
  if (d.on) {
#pragma unroll
    for (int i = 0; i < PF; ++i) { kf[i] = __builtin_nontemporal_load(pk + (long)i * 1024); vf[i] = __builtin_nontemporal_load(pv + (long)i * 1024); }
  }
  const unsigned ep_ = dep_epoch(d);
  dep_wait(d, ep_);
  const u32x4_t r4 = ld16_dep(act_in + (blockIdx.x / 12) * 48 + (threadIdx.x % 48), d.on);
  if (d.on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned wacc = r4[0];
#pragma unroll
  for (int i = 0; i < 6; ++i) wacc += wv[i].x ^ wv[i].y ^ wv[i].z ^ wv[i].w;
  float m = -1e30f + (float)(wacc & 1), l = 0.f, a = 0.f;
  auto eat = [&](u32x4_t k, u32x4_t v) {
    float dd = __uint_as_float(k[0] << 16) + __uint_as_float(k[1] << 16) + __uint_as_float(k[2] << 16) + __uint_as_float(k[3] << 16);
    dd += __shfl_xor(dd, 1, 64); dd += __shfl_xor(dd, 2, 64); dd += __shfl_xor(dd, 4, 64);
    const float mn = fmaxf(m, dd);
    const float fa = __expf(m - mn), pu = __expf(dd - mn);
    l = l * fa + pu;
    a = a * fa + pu * (__uint_as_float(v[0] << 16) + __uint_as_float(v[3] << 16));
    m = mn;
  };
  int i0 = 0;
  if (d.on) {
#pragma unroll
    for (int i = 0; i < PF; ++i) eat(kf[i], vf[i]);
    i0 = PF;
  }
  for (int i = i0; i < iters; ++i) eat(__builtin_nontemporal_load(pk + (long)i * 1024), __builtin_nontemporal_load(pv + (long)i * 1024));
  __shared__ float red[16];
  float s = a / (l + 1.f);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x < 8) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    const unsigned r = (unsigned)(t * 1e-20f) & 1u;
    st16_wt(act_out + (blockIdx.x * 8 + threadIdx.x) % 1536, u32x4_t{r, r, r, r});
  }
  dep_signal(d, ep_);
}

__global__ void k_bump(unsigned* epoch, unsigned* progress) {       // end of a step: the next step's targets move up
  const unsigned e = *epoch + 1;
  __hip_atomic_store(epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(progress, e * 128u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Bufs {
  uint4* w; u32x4_t* kv; u32x4_t* act[2];
  unsigned *progress, *tickets, *epoch, *err;
};
static const long kLayerW16 = (16L << 20) / 16;
static const int kIters = 10;
static const long kKvWg16 = 2L * kIters * 1024;
static const long kKvLayer16 = 192L * kKvWg16;

// one step (12 layers x 6 kernels); mode 0 plain on s0; mode 1 overlap: kernel k on (k & 1 ? s1 : s0)
static int g_proto = 0, g_sleep = 2;
// modes: 0 plain; 1 overlap, ONE graph with two branches; 2 overlap, TWO graphs (even / odd kernels) on two streams; 3 the overlap
// protocol on a single stream (prices the protocol alone).  which: -1 every kernel, 0 / 1 the even / odd ones only.
static void enqueue_step(hipStream_t s0, hipStream_t s1, const Bufs& b, int mode, int which) {
  unsigned k = 0;
  unsigned prev = 0;
  auto dep = [&](unsigned nwg) { Dep d{b.progress, b.tickets + (long)k * 9 * 32, b.epoch, k, nwg, prev, b.err, mode != 0, g_proto, g_sleep}; prev = nwg; return d; };
  auto st = [&]() { return ((mode == 1 || mode == 2) && (k & 1)) ? s1 : s0; };
  auto take = [&]() { return which < 0 || (int)(k & 1) == which; };
  for (int l = 0; l < 12; ++l) {
    const uint4* w = b.w + (long)l * kLayerW16;
    if (take()) hipLaunchKernelGGL(k_heavy, dim3(192), dim3(1024), 0, st(), w, b.act[k & 1], b.act[(k + 1) & 1], 12, dep(192));
    ++k;
    if (take()) hipLaunchKernelGGL((k_gemv<3>), dim3(192), dim3(256), 0, st(), w + 400000, b.act[k & 1], b.act[(k + 1) & 1], dep(192));
    ++k;
    if (take()) hipLaunchKernelGGL((k_stream<2>), dim3(192), dim3(1024), 0, st(), w + 200000, b.kv + (long)l * kKvLayer16, kKvWg16, kIters, b.act[k & 1],
                                   b.act[(k + 1) & 1], dep(192));
    ++k;
    if (take()) hipLaunchKernelGGL((k_gemv<3>), dim3(192), dim3(256), 0, st(), w + 500000, b.act[k & 1], b.act[(k + 1) & 1], dep(192));
    ++k;
    if (take()) hipLaunchKernelGGL((k_gemv<6>), dim3(256), dim3(256), 0, st(), w + 600000, b.act[k & 1], b.act[(k + 1) & 1], dep(256));
    ++k;
    if (take()) hipLaunchKernelGGL((k_gemv<8>), dim3(192), dim3(256), 0, st(), w + 800000, b.act[k & 1], b.act[(k + 1) & 1], dep(192));
    ++k;
  }
}

static double run_chain(hipStream_t s0, hipStream_t s1, const Bufs& b, int mode, int reps, unsigned* err_out) {
  hipGraph_t g, g2 = nullptr; hipGraphExec_t ge, ge2 = nullptr;
  hipEvent_t fork, join; CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  CHECK(hipMemsetAsync(b.progress, 0, 4, s0)); CHECK(hipMemsetAsync(b.epoch, 0, 4, s0)); CHECK(hipMemsetAsync(b.tickets, 0, 80 * 9 * 32 * 4, s0));
  CHECK(hipMemsetAsync(b.err, 0, 4, s0));
  CHECK(hipStreamSynchronize(s0));
  if (mode == 2) {
    CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    enqueue_step(s0, s1, b, mode, 0);
    CHECK(hipStreamEndCapture(s0, &g));
    CHECK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    enqueue_step(s0, s1, b, mode, 1);
    CHECK(hipStreamEndCapture(s1, &g2));
    CHECK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  } else {
    CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    if (mode == 1) { CHECK(hipEventRecord(fork, s0)); CHECK(hipStreamWaitEvent(s1, fork, 0)); }
    enqueue_step(s0, s1, b, mode, -1);
    if (mode == 1) { CHECK(hipEventRecord(join, s1)); CHECK(hipStreamWaitEvent(s0, join, 0)); }
    hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, s0, b.epoch, b.progress);
    CHECK(hipStreamEndCapture(s0, &g));
  }
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  auto one = [&]() {
    if (mode != 2) { CHECK(hipGraphLaunch(ge, s0)); return; }
    CHECK(hipGraphLaunch(ge, s0));                      // even kernels
    CHECK(hipGraphLaunch(ge2, s1));                     // odd kernels: progress words order them against the even ones
    CHECK(hipEventRecord(join, s1)); CHECK(hipStreamWaitEvent(s0, join, 0));
    hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, s0, b.epoch, b.progress);
    CHECK(hipEventRecord(fork, s0)); CHECK(hipStreamWaitEvent(s1, fork, 0));
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) one();
  CHECK(hipStreamSynchronize(s0));
  CHECK(hipEventRecord(e0, s0));
  for (int i = 0; i < reps; ++i) one();
  CHECK(hipEventRecord(e1, s0));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipStreamSynchronize(s1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(err_out, b.err, 4, hipMemcpyDeviceToHost));
  CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  if (ge2) { CHECK(hipGraphExecDestroy(ge2)); CHECK(hipGraphDestroy(g2)); }
  return ms * 1e3 / ((double)reps * 12);
}

int main(int argc, char** argv) {
  const int NC = 2, reps = argc > 1 ? atoi(argv[1]) : 40;
  Bufs b[NC];
  uint4* w; CHECK(hipMalloc(&w, (size_t)14 * kLayerW16 * 16)); CHECK(hipMemset(w, 1, (size_t)14 * kLayerW16 * 16));
  hipStream_t s[NC][2];
  for (int c = 0; c < NC; ++c) {
    b[c].w = w;
    CHECK(hipMalloc(&b[c].kv, (size_t)12 * kKvLayer16 * 16)); CHECK(hipMemset(b[c].kv, 0, (size_t)12 * kKvLayer16 * 16));
    for (int i = 0; i < 2; ++i) { CHECK(hipMalloc(&b[c].act[i], 1536 * 16)); CHECK(hipMemset(b[c].act[i], 0, 1536 * 16)); }
    CHECK(hipMalloc(&b[c].progress, 256)); CHECK(hipMalloc(&b[c].epoch, 256)); CHECK(hipMalloc(&b[c].err, 256));
    CHECK(hipMalloc(&b[c].tickets, 80 * 9 * 32 * 4));
    for (int i = 0; i < 2; ++i) CHECK(hipStreamCreateWithFlags(&s[c][i], hipStreamNonBlocking));
  }
  const char* names[4] = {"plain   (stream order)", "overlap, one graph with two branches", "overlap, two graphs on two streams", "protocol only (one stream)"};
  const int two_graphs = argc > 2 ? atoi(argv[2]) : 0;     // (the two-graph form loses dependences when its streams share a hardware queue)
  for (int proto = 0; proto < 3; ++proto)
  for (int sl = 0; sl < (proto == 0 ? 1 : 2); ++sl)
  for (int n = 1; n <= 2; ++n)
    for (int mode = (proto == 0 && sl == 0) ? 0 : 1; mode < 4; ++mode) {
      if (mode == 2 && !two_graphs) continue;
      g_proto = proto; g_sleep = sl == 0 ? 2 : 16;
      if (mode == 1 && n == 1) printf("-- protocol %d, poll sleep %d\n", proto, g_sleep);
      double r[NC] = {0, 0}; unsigned err[NC] = {0, 0};
      std::vector<std::thread> th;
      for (int i = 1; i < n; ++i) th.emplace_back([&, i] { r[i] = run_chain(s[i][0], s[i][1], b[i], mode, reps, &err[i]); });
      r[0] = run_chain(s[0][0], s[0][1], b[0], mode, reps, &err[0]);
      for (auto& t : th) t.join();
      printf("%-40s x%d  us per layer per chain:", names[mode], n);
      for (int i = 0; i < n; ++i) printf(" %7.2f", r[i]);
      printf("   lost dependences: %u %u\n", err[0], err[1]);
      fflush(stdout);
    }
  return 0;
}
