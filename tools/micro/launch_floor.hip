// Micro-benchmark: what does ONE dependent kernel cost inside a replayed hipGraph (or an eager stream) on MI355X?
// The decode step of libmapperhip is 74 dependent kernels per 16-row chain; rocprofv3 shows ~4 us for kernels that
// do almost nothing.  This probe separates the boundary cost from what the kernel body adds:
//   empty bodies at three launch geometries, a dependent 1-float chain, a large kernarg struct, a weight-streaming
//   GEMV-like body at 48 / 256 workgroups, six alternating fat-code kernels (instruction-cache pressure), eager
//   launches instead of a graph, and two graphs replayed concurrently from two host threads.
// Output: wall microseconds per kernel (HIP events around `reps` replays of a `chain`-kernel graph).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct BigArgs { float* buf; int n; int pad[96]; };   // ~400 bytes of kernarg

__global__ void k_empty() {}
__global__ void k_chain(float* buf) {   // reads what the previous kernel wrote
  if (threadIdx.x == 0) buf[blockIdx.x] = buf[blockIdx.x] + 1.0f;
}
__global__ void k_chain_big(BigArgs a) {
  if (threadIdx.x == 0) a.buf[blockIdx.x] = a.buf[blockIdx.x] + (float)a.pad[95];
}
// every workgroup streams `bytes_per_wg` of "weights" (all loads in flight, one round trip), reduces, writes one float
template <int LOADS>
__global__ __launch_bounds__(256) void k_gemv(const uint4* __restrict__ w, long wg_stride16, float* out) {
  const uint4* p = w + (long)blockIdx.x * wg_stride16 + threadIdx.x;
  uint4 v[LOADS];
#pragma unroll
  for (int i = 0; i < LOADS; ++i) v[i] = p[i * 256];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < LOADS; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  __shared__ unsigned red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int i = 0; i < 256; ++i) t += red[i];
    out[blockIdx.x] = out[blockIdx.x] + (float)(t & 1);
  }
}
// fat straight-line code (executed): ~2K dependent-free FMAs per thread, unique constants per ID
template <int ID>
__global__ __launch_bounds__(256) void k_fat(float* buf) {
  float a = buf[blockIdx.x], b = a + 1.f, c = a + 2.f, d = a + 3.f;
#pragma unroll
  for (int i = 0; i < 512; ++i) {
    a = a * (1.0f + 1e-7f * (float)(ID * 1000 + i)) + 1e-9f;
    b = b * (1.0f + 2e-7f * (float)(ID * 1000 + i)) + 2e-9f;
    c = c * (1.0f + 3e-7f * (float)(ID * 1000 + i)) + 3e-9f;
    d = d * (1.0f + 4e-7f * (float)(ID * 1000 + i)) + 4e-9f;
  }
  if (threadIdx.x == 0) buf[blockIdx.x] = a + b + c + d;
}

typedef void (*EnqueueFn)(hipStream_t, int idx);

static float* g_buf;
static uint4* g_w;
static long g_layer16;   // uint4 elements per "layer" of weights

static double run_graph(EnqueueFn fn, int chain, int reps, hipStream_t s) {
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < chain; ++i) fn(s, i);
  CHECK(hipStreamEndCapture(s, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(ge, s));
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(ge, s));
  CHECK(hipEventRecord(e1, s));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  return ms * 1e3 / ((double)reps * chain);
}
static double run_eager(EnqueueFn fn, int chain, int reps, hipStream_t s) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < chain; ++i) fn(s, i);
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) for (int i = 0; i < chain; ++i) fn(s, i);
  CHECK(hipEventRecord(e1, s));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / ((double)reps * chain);
}

static void f_empty_1x64(hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }
static void f_empty_256x256(hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }
static void f_empty_192x1024(hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(192), dim3(1024), 0, s); }
static void f_chain_1(hipStream_t s, int) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s, g_buf); }
static void f_chain(hipStream_t s, int) { hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, s, g_buf); }
static void f_chain_big(hipStream_t s, int) { BigArgs a{}; a.buf = g_buf; hipLaunchKernelGGL(k_chain_big, dim3(256), dim3(256), 0, s, a); }
static void f_gemv48(hipStream_t s, int i) {   // 48 workgroups x 24 KB  (the o-projection of a 16-row chain)
  hipLaunchKernelGGL((k_gemv<6>), dim3(48), dim3(256), 0, s, g_w + (long)(i % 12) * g_layer16, (long)6 * 256, g_buf);
}
static void f_gemv192(hipStream_t s, int i) {  // the same bytes over 192 workgroups x 6 KB (+2 idle loads)
  hipLaunchKernelGGL((k_gemv<2>), dim3(192), dim3(256), 0, s, g_w + (long)(i % 12) * g_layer16, (long)2 * 256, g_buf);
}
static void f_gemv256_big(hipStream_t s, int i) {   // 256 workgroups x 24 KB = 6.3 MB (the wi GEMV)
  hipLaunchKernelGGL((k_gemv<6>), dim3(256), dim3(256), 0, s, g_w + (long)(i % 12) * g_layer16, (long)6 * 256, g_buf);
}
static void f_fat(hipStream_t s, int i) {
  switch (i % 6) {
    case 0: hipLaunchKernelGGL(k_fat<0>, dim3(256), dim3(256), 0, s, g_buf); break;
    case 1: hipLaunchKernelGGL(k_fat<1>, dim3(256), dim3(256), 0, s, g_buf); break;
    case 2: hipLaunchKernelGGL(k_fat<2>, dim3(256), dim3(256), 0, s, g_buf); break;
    case 3: hipLaunchKernelGGL(k_fat<3>, dim3(256), dim3(256), 0, s, g_buf); break;
    case 4: hipLaunchKernelGGL(k_fat<4>, dim3(256), dim3(256), 0, s, g_buf); break;
    default: hipLaunchKernelGGL(k_fat<5>, dim3(256), dim3(256), 0, s, g_buf); break;
  }
}
static void f_fat_same(hipStream_t s, int) { hipLaunchKernelGGL(k_fat<0>, dim3(256), dim3(256), 0, s, g_buf); }

int main() {
  const int chain = 72, reps = 40;
  CHECK(hipMalloc(&g_buf, 1 << 20));
  CHECK(hipMemset(g_buf, 0, 1 << 20));
  g_layer16 = (long)(16 << 20) / 16;   // 16 MB per layer, 12 layers = 192 MB
  CHECK(hipMalloc(&g_w, (size_t)12 * g_layer16 * 16));
  CHECK(hipMemset(g_w, 1, (size_t)12 * g_layer16 * 16));
  hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  struct { const char* name; EnqueueFn fn; } cases[] = {
      {"empty 1x64", f_empty_1x64}, {"empty 256x256", f_empty_256x256}, {"empty 192x1024", f_empty_192x1024},
      {"dependent 1-float chain 1x64", f_chain_1}, {"dependent 1-float chain 256x256", f_chain},
      {"same + 400-byte kernarg", f_chain_big}, {"gemv-like 48 WG x 24 KB", f_gemv48},
      {"gemv-like 192 WG x 8 KB", f_gemv192}, {"gemv-like 256 WG x 24 KB", f_gemv256_big},
      {"fat code, one kernel", f_fat_same}, {"fat code, six alternating kernels", f_fat}};
  for (auto& c : cases) {
    const double g = run_graph(c.fn, chain, reps, s);
    const double e = run_eager(c.fn, chain, reps, s);
    printf("%-40s graph %6.2f us/kernel   eager %6.2f us/kernel\n", c.name, g, e);
  }
  // two graphs replayed concurrently from two host threads (the two decode chains)
  hipStream_t s2; CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (EnqueueFn fn : {f_chain, f_gemv48}) {
    double r[2];
    std::thread t([&] { r[1] = run_graph(fn, chain, reps, s2); });
    r[0] = run_graph(fn, chain, reps, s);
    t.join();
    printf("two concurrent graphs (%s): %6.2f / %6.2f us/kernel per stream\n", fn == f_chain ? "1-float chain" : "gemv 48 WG", r[0], r[1]);
  }
  // four graphs from four host threads (why do 3-4 decode chains run slower than 2?)
  {
    hipStream_t ss[4] = {s, s2, nullptr, nullptr};
    CHECK(hipStreamCreateWithFlags(&ss[2], hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&ss[3], hipStreamNonBlocking));
    for (EnqueueFn fn : {f_chain, f_gemv48}) {
      double r[4];
      std::vector<std::thread> th;
      for (int i = 1; i < 4; ++i) th.emplace_back([&, i] { r[i] = run_graph(fn, chain, reps, ss[i]); });
      r[0] = run_graph(fn, chain, reps, ss[0]);
      for (auto& t : th) t.join();
      printf("four concurrent graphs (%s): %6.2f / %6.2f / %6.2f / %6.2f us/kernel per stream\n",
             fn == f_chain ? "1-float chain" : "gemv 48 WG", r[0], r[1], r[2], r[3]);
    }
    // the same four streams fed from ONE host thread (round-robin graph launches)
    for (EnqueueFn fn : {f_chain, f_gemv48}) {
      hipGraph_t g[4]; hipGraphExec_t ge[4];
      for (int i = 0; i < 4; ++i) {
        CHECK(hipStreamBeginCapture(ss[i], hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < chain; ++k) fn(ss[i], k);
        CHECK(hipStreamEndCapture(ss[i], &g[i]));
        CHECK(hipGraphInstantiate(&ge[i], g[i], nullptr, nullptr, 0));
      }
      for (int i = 0; i < 4; ++i) CHECK(hipGraphLaunch(ge[i], ss[i]));
      CHECK(hipDeviceSynchronize());
      timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
      for (int r2 = 0; r2 < reps; ++r2) for (int i = 0; i < 4; ++i) CHECK(hipGraphLaunch(ge[i], ss[i]));
      CHECK(hipDeviceSynchronize());
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
      printf("four streams, one launcher thread (%s): %6.2f us per kernel per stream (%.2f us per kernel overall)\n",
             fn == f_chain ? "1-float chain" : "gemv 48 WG", us / ((double)reps * chain), us / ((double)reps * chain * 4));
    }
  }
  // host cost of replaying a 72-node graph (launch call only, GPU idle wait excluded)
  {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < chain; ++i) f_empty_1x64(s, i);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s)); CHECK(hipStreamSynchronize(s));
    timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < 20; ++i) CHECK(hipGraphLaunch(ge, s));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    CHECK(hipStreamSynchronize(s));
    printf("host time of hipGraphLaunch: %.2f us per node (72-node graph)\n",
           ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3) / (20.0 * chain));
  }
  return 0;
}
