// Micro-benchmark: can one kernel warm an XCD's L2 for the NEXT kernel of the same stream, and does the warm data
// survive a K/V-sized stream of loads in between?  (The decode GEMVs fetch 1.2 - 6.3 MB of weights per launch from the
// memory side -- PMC FETCH_SIZE, profiles/r02b_pmc_hbm_traffic.txt -- and sit on that first-load latency.)
//
//   consume : 96 workgroups x 256 threads, workgroup t reads its 12 KB weight tile (3 x 16 B per thread, all in flight)
//             and records the cycles from issue to arrival -- the o-projection GEMV's weight read.
//   touch   : 192 workgroups; a workgroup that runs on XCD x (HW_REG_XCC_ID) touches one dword per 128-byte line of the
//             tiles t with t % 8 == (x + shift) % 8 -- shift 0: the L2 the consumer is expected on, 1: a neighbour's.
//   stream  : 192 x 1024 threads read `mb` MB once (non-temporal or plain loads) -- the cross-attention K/V stream.
//   evict   : reads 768 MB (L2 and Infinity Cache cold afterwards).
// Output per sequence: mean / median / max nanoseconds of the consumers' load round trip and the share of consumer
// workgroups that ran on XCD t % 8.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int kTileBytes = 12288, kTiles = 96;

__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15; }

__global__ __launch_bounds__(256) void k_consume(const char* w, unsigned long long* rec, unsigned* sink) {
  const uint4* p = reinterpret_cast<const uint4*>(w + (long)blockIdx.x * kTileBytes) + threadIdx.x;
  const unsigned long long t0 = clock64();
  const unsigned long long w0 = wall_clock64();
  uint4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = p[i * 256];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64();
  const unsigned long long w1 = wall_clock64();
  if (acc == 0x12345u) sink[0] = acc;
  if (threadIdx.x == 0) {
    rec[blockIdx.x * 4 + 0] = t1 - t0;
    rec[blockIdx.x * 4 + 1] = w1 - w0;
    rec[blockIdx.x * 4 + 2] = (unsigned long long)xcc_id();
  }
}

__global__ __launch_bounds__(256) void k_touch(const char* w, int shift, unsigned* sink) {
  const int x = (xcc_id() + shift) & 7, sub = blockIdx.x >> 3, nsub = gridDim.x >> 3;
  unsigned acc = 0;
  for (int t = x + 8 * sub; t < kTiles; t += 8 * nsub)
    if (threadIdx.x < kTileBytes / 128) acc ^= *reinterpret_cast<const unsigned*>(w + (long)t * kTileBytes + threadIdx.x * 128);
  if (acc == 0x12345u) sink[1] = acc;
}

template <bool NT>
__global__ __launch_bounds__(1024) void k_stream(const char* kv, long bytes, unsigned* sink) {
  const long n16 = bytes / 16;
  const u32x4_t* p = reinterpret_cast<const u32x4_t*>(kv);
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (long)gridDim.x * 1024) {
    const u32x4_t v = NT ? __builtin_nontemporal_load(p + i) : p[i];
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345u) sink[2] = acc;
}

static void report(const char* name, const std::vector<unsigned long long>& rec, int reps, double wall_khz) {
  std::vector<double> ns, cyc;
  int match = 0, n = 0;
  for (int r = 0; r < reps; ++r)
    for (int t = 0; t < kTiles; ++t) {
      const unsigned long long* q = &rec[((long)r * kTiles + t) * 4];
      cyc.push_back((double)q[0]);
      ns.push_back((double)q[1] / wall_khz * 1e6);
      match += ((int)q[2] == (t & 7));
      ++n;
    }
  std::sort(ns.begin(), ns.end());
  std::sort(cyc.begin(), cyc.end());
  double m = 0, mc = 0;
  for (double v : ns) m += v;
  for (double v : cyc) mc += v;
  printf("  %-46s  load round trip mean %7.0f ns  median %7.0f  p90 %7.0f  max %7.0f | shader cycles mean %7.0f | on XCD t%%8: %3.0f %%\n",
         name, m / n, ns[n / 2], ns[n * 9 / 10], ns[n - 1], mc / n, 100.0 * match / n);
}

int main() {
  int dev = 0, khz = 100000;
  CHECK(hipSetDevice(dev));
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  const long evict_bytes = 768L << 20, kv_bytes = 64L << 20;
  const int reps = 20;
  char *w, *ev, *kv;
  unsigned long long* rec;
  unsigned* sink;
  CHECK(hipMalloc(&w, (long)kTiles * kTileBytes));
  CHECK(hipMalloc(&ev, evict_bytes));
  CHECK(hipMalloc(&kv, kv_bytes));
  CHECK(hipMalloc(&rec, (long)reps * kTiles * 4 * 8));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(w, 1, (long)kTiles * kTileBytes));
  CHECK(hipMemset(ev, 1, evict_bytes));
  CHECK(hipMemset(kv, 1, kv_bytes));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  std::vector<unsigned long long> host((long)reps * kTiles * 4);
  auto evict = [&] { hipLaunchKernelGGL(k_stream<false>, dim3(1024), dim3(1024), 0, s, ev, evict_bytes, sink); };
  auto consume = [&](int r) { hipLaunchKernelGGL(k_consume, dim3(kTiles), dim3(256), 0, s, w, rec + (long)r * kTiles * 4, sink); };
  auto touch = [&](int shift) { hipLaunchKernelGGL(k_touch, dim3(192), dim3(256), 0, s, w, shift, sink); };
  auto fetch = [&](const char* name) {
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(host.data(), rec, host.size() * 8, hipMemcpyDeviceToHost));
    report(name, host, reps, (double)khz);
  };
  printf("l2_persist: %d tiles x %d bytes, wall clock %d kHz\n", kTiles, kTileBytes, khz);
  for (int r = 0; r < reps; ++r) { evict(); consume(r); }
  fetch("cold (after a 768 MB sweep)");
  for (int r = 0; r < reps; ++r) { evict(); consume(r); consume(r); }
  fetch("second of two consumers back to back");
  for (int r = 0; r < reps; ++r) { evict(); touch(0); consume(r); }
  fetch("touched by the kernel before, same XCD");
  for (int r = 0; r < reps; ++r) { evict(); touch(1); consume(r); }
  fetch("touched by the kernel before, neighbour XCD");
  for (int r = 0; r < reps; ++r) { evict(); touch(0); hipLaunchKernelGGL(k_stream<true>, dim3(192), dim3(1024), 0, s, kv, kv_bytes, sink); consume(r); }
  fetch("touched, then 64 MB non-temporal stream");
  for (int r = 0; r < reps; ++r) { evict(); touch(0); hipLaunchKernelGGL(k_stream<false>, dim3(192), dim3(1024), 0, s, kv, kv_bytes, sink); consume(r); }
  fetch("touched, then 64 MB plain stream");
  for (int r = 0; r < reps; ++r) { evict(); touch(0); hipLaunchKernelGGL(k_stream<true>, dim3(192), dim3(1024), 0, s, kv, 16L << 20, sink); consume(r); }
  fetch("touched, then 16 MB non-temporal stream");
  for (int r = 0; r < reps; ++r) { evict(); touch(0); touch(0); touch(0); consume(r); }
  fetch("touched three kernels ago (two kernels between)");
  // Infinity Cache only: warm by a neighbour-XCD touch two sweeps of a small buffer ago
  for (int r = 0; r < reps; ++r) { evict(); touch(1); hipLaunchKernelGGL(k_stream<true>, dim3(192), dim3(1024), 0, s, kv, kv_bytes, sink); consume(r); }
  fetch("neighbour touch, then 64 MB nt stream");
  // ---- second question: how fast is a re-read that hits the Infinity Cache?  `mb` MB read twice, each pass timed ----
  hipEvent_t e0, e1, e2;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
  for (int nt = 0; nt < 2; ++nt)
    for (long mb : {16L, 32L, 64L, 128L, 192L}) {
      double t1 = 0, t2 = 0;
      const int R = 5;
      for (int r = 0; r < R; ++r) {
        evict();
        CHECK(hipEventRecord(e0, s));
        if (nt) hipLaunchKernelGGL(k_stream<true>, dim3(512), dim3(1024), 0, s, ev, mb << 20, sink);
        else hipLaunchKernelGGL(k_stream<false>, dim3(512), dim3(1024), 0, s, ev, mb << 20, sink);
        CHECK(hipEventRecord(e1, s));
        if (nt) hipLaunchKernelGGL(k_stream<true>, dim3(512), dim3(1024), 0, s, ev, mb << 20, sink);
        else hipLaunchKernelGGL(k_stream<false>, dim3(512), dim3(1024), 0, s, ev, mb << 20, sink);
        CHECK(hipEventRecord(e2, s));
        CHECK(hipEventSynchronize(e2));
        float a, b;
        CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
        t1 += a; t2 += b;
      }
      printf("  %3ld MB %s loads: first pass %6.1f us = %5.2f TB/s, second pass %6.1f us = %5.2f TB/s\n", mb, nt ? "non-temporal" : "plain       ",
             t1 / R * 1e3, (double)(mb << 20) / (t1 / R * 1e-3) / 1e12, t2 / R * 1e3, (double)(mb << 20) / (t2 / R * 1e-3) / 1e12);
    }
  return 0;
}
