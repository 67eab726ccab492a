// Micro-benchmark: why do two decode chains overlap only 1.5x?
// A decode chain is 12 x [self-attention (192 x 1024 threads, 295 KB of shared weights per workgroup), GEMV, cross-
// attention (192 x 1024 threads streaming 61.5 MB from HBM), GEMV, GEMV, GEMV] of dependent kernels replayed as a graph.
// This probe replays SYNTHETIC chains with the same launch geometry and memory behaviour -- alone, two and three at a
// time from separate host threads -- and ablates the kernel kinds, to separate "streaming kernels slow everybody's
// memory round trips" from "wide workgroups monopolise CUs" from "concurrent queues do not overlap".
// Output: microseconds per layer (6 kernels) per chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// GEMV-like: 192 workgroups x 256 threads, LOADS x 16 B per thread in one round trip, LDS reduce, dependent update
template <int LOADS>
__global__ __launch_bounds__(256) void k_gemv(const uint4* __restrict__ w, long wg_stride16, float* out) {
  const uint4* p = w + (long)blockIdx.x * wg_stride16 + threadIdx.x;
  const float old = out[blockIdx.x];
  uint4 v[LOADS];
#pragma unroll
  for (int i = 0; i < LOADS; ++i) v[i] = p[i * 256];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < LOADS; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  __shared__ unsigned red[4];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = old + (float)((red[0] + red[1] + red[2] + red[3]) & 1);
}

// self-attention-like: 1024 threads, first a dependent row load (the residual row), then 18 x 16 B of weights per thread
// (295 KB per workgroup, the slice shared by the workgroups of the same head), LDS reduce
__global__ __launch_bounds__(1024) void k_heavy(const uint4* __restrict__ w, const float* row, float* out, int heads) {
  const int head = blockIdx.x % heads;
  const float r = row[(blockIdx.x / heads) * 1024 + threadIdx.x];
  __shared__ float red[16];
  float s = r * r;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float tot = 0.f;
  for (int i = 0; i < 16; ++i) tot += red[i];
  const uint4* p = w + (long)head * 18 * 1024 + threadIdx.x;
  uint4 v[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) v[i] = p[i * 1024];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 18; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (float)(acc & 1) + tot;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = out[blockIdx.x] * 0.5f + red[3];
}

// cross-attention-like: 1024 threads stream `iters` x 2 x 16 B per thread (non-temporal), two loads in flight, a few
// dependent flops per pair (online softmax), LDS merge
__global__ __launch_bounds__(1024) void k_stream(const u32x4_t* __restrict__ kv, long wg_stride16, int iters, float* out) {
  const u32x4_t* pk = kv + (long)blockIdx.x * wg_stride16 + threadIdx.x;
  const u32x4_t* pv = pk + (long)iters * 1024;
  float m = -1e30f, l = 0.f, a = 0.f;
  for (int i = 0; i < iters; ++i) {
    const u32x4_t k = __builtin_nontemporal_load(pk + (long)i * 1024);
    const u32x4_t v = __builtin_nontemporal_load(pv + (long)i * 1024);
    float d = __uint_as_float(k[0] << 16) + __uint_as_float(k[1] << 16) + __uint_as_float(k[2] << 16) + __uint_as_float(k[3] << 16);
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
    const float mn = fmaxf(m, d);
    const float fa = __expf(m - mn), pu = __expf(d - mn);
    l = l * fa + pu;
    a = a * fa + pu * (__uint_as_float(v[0] << 16) + __uint_as_float(v[3] << 16));
    m = mn;
  }
  __shared__ float red[16];
  float s = a / (l + 1.f);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    out[blockIdx.x] = out[blockIdx.x] * 0.5f + t * 1e-20f;
  }
}

struct Bufs {
  uint4* w;        // 12 layers x 16 MB of "weights" (shared by the chains)
  u32x4_t* kv;     // this chain's 12 x 63 MB K/V stream
  float* row;      // 16 x 1024 floats
  float* out;      // 1024 floats
};
static const long kLayerW16 = (16L << 20) / 16;
static const int kIters = 10;                                     // 192 x 1024 x 10 x 2 x 16 B = 62.9 MB per launch
static const long kKvWg16 = 2L * kIters * 1024;                   // uint4 per workgroup
static const long kKvLayer16 = 192L * kKvWg16;

enum { HEAVY = 1, STREAM = 2 };   // which kinds a chain keeps (the others become GEMVs)

static void enqueue_layer(hipStream_t s, const Bufs& b, int l, int kinds, int stream_wgs) {
  const uint4* w = b.w + (long)l * kLayerW16;
  if (kinds & HEAVY) hipLaunchKernelGGL(k_heavy, dim3(192), dim3(1024), 0, s, w, b.row, b.out, 12);
  else hipLaunchKernelGGL((k_gemv<6>), dim3(192), dim3(256), 0, s, w, 6L * 256, b.out);
  hipLaunchKernelGGL((k_gemv<6>), dim3(192), dim3(256), 0, s, w + 400000, 6L * 256, b.out);
  if (kinds & STREAM)
    hipLaunchKernelGGL(k_stream, dim3(stream_wgs), dim3(1024), 0, s, b.kv + (long)l * kKvLayer16, kKvWg16 * 192 / stream_wgs,
                       kIters * 192 / stream_wgs, b.out);
  else hipLaunchKernelGGL((k_gemv<6>), dim3(192), dim3(256), 0, s, w + 100000, 6L * 256, b.out);
  hipLaunchKernelGGL((k_gemv<6>), dim3(192), dim3(256), 0, s, w + 500000, 6L * 256, b.out);
  hipLaunchKernelGGL((k_gemv<6>), dim3(192), dim3(256), 0, s, w + 600000, 6L * 256, b.out);
  hipLaunchKernelGGL((k_gemv<6>), dim3(192), dim3(256), 0, s, w + 800000, 6L * 256, b.out);
}

static double run_chain(hipStream_t s, const Bufs& b, int kinds, int stream_wgs, int reps) {
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < 12; ++l) enqueue_layer(s, b, l, kinds, stream_wgs);
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
  return ms * 1e3 / ((double)reps * 12);
}

int main() {
  const int NC = 3, reps = 60;
  Bufs b[NC];
  uint4* w; CHECK(hipMalloc(&w, (size_t)14 * kLayerW16 * 16)); CHECK(hipMemset(w, 1, (size_t)14 * kLayerW16 * 16));
  hipStream_t s[NC];
  for (int c = 0; c < NC; ++c) {
    b[c].w = w;
    CHECK(hipMalloc(&b[c].kv, (size_t)12 * kKvLayer16 * 16)); CHECK(hipMemset(b[c].kv, 0, (size_t)12 * kKvLayer16 * 16));
    CHECK(hipMalloc(&b[c].row, 16 * 1024 * 4)); CHECK(hipMemset(b[c].row, 0, 16 * 1024 * 4));
    CHECK(hipMalloc(&b[c].out, 4096)); CHECK(hipMemset(b[c].out, 0, 4096));
    CHECK(hipStreamCreateWithFlags(&s[c], hipStreamNonBlocking));
  }
  struct Case { const char* name; int kinds[NC]; int n; int stream_wgs; };
  const Case cases[] = {
      {"gemv only                 x1", {0, 0, 0}, 1, 192},
      {"gemv only                 x2", {0, 0, 0}, 2, 192},
      {"gemv only                 x3", {0, 0, 0}, 3, 192},
      {"gemv + heavy              x1", {HEAVY, HEAVY, HEAVY}, 1, 192},
      {"gemv + heavy              x2", {HEAVY, HEAVY, HEAVY}, 2, 192},
      {"gemv + stream             x1", {STREAM, STREAM, STREAM}, 1, 192},
      {"gemv + stream             x2", {STREAM, STREAM, STREAM}, 2, 192},
      {"gemv + stream             x3", {STREAM, STREAM, STREAM}, 3, 192},
      {"full (heavy + stream)     x1", {HEAVY | STREAM, HEAVY | STREAM, HEAVY | STREAM}, 1, 192},
      {"full (heavy + stream)     x2", {HEAVY | STREAM, HEAVY | STREAM, HEAVY | STREAM}, 2, 192},
      {"full (heavy + stream)     x3", {HEAVY | STREAM, HEAVY | STREAM, HEAVY | STREAM}, 3, 192},
      {"victim gemv-only | streamer  ", {0, STREAM, 0}, 2, 192},
      {"victim gemv-only | 2 streamers", {0, STREAM, STREAM}, 3, 192},
      {"gemv + stream on 96 WGs   x1", {STREAM, STREAM, STREAM}, 1, 96},
      {"gemv + stream on 96 WGs   x2", {STREAM, STREAM, STREAM}, 2, 96},
      {"gemv + stream on 48 WGs   x1", {STREAM, STREAM, STREAM}, 1, 48},
      {"gemv + stream on 48 WGs   x2", {STREAM, STREAM, STREAM}, 2, 48},
  };
  for (const Case& c : cases) {
    double r[NC] = {0, 0, 0};
    std::vector<std::thread> th;
    for (int i = 1; i < c.n; ++i) th.emplace_back([&, i] { r[i] = run_chain(s[i], b[i], c.kinds[i], c.stream_wgs, reps); });
    r[0] = run_chain(s[0], b[0], c.kinds[0], c.stream_wgs, reps);
    for (auto& t : th) t.join();
    printf("%-34s us per layer per chain:", c.name);
    for (int i = 0; i < c.n; ++i) printf(" %7.2f", r[i]);
    printf("\n");
  }
  return 0;
}
