// Micro-benchmark: cost of an in-kernel grid-wide barrier on MI355X (decides whether a persistent per-token-step
// decode kernel can beat ~100 dependent kernel launches of ~4 us each).  Two barrier forms, every spin bounded.
//   flat : one monotonic counter, lane 0 of every block: release fence -> atomic add -> relaxed poll -> acquire fence
//   xcd  : per-XCD counters (block b -> XCD b % 8), XCD leader aggregates into a top counter, generation words
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline bool spin_until(const int* p, int target, int max_iter) {
  for (int i = 0; i < max_iter; ++i) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

__global__ void flat_barrier_kernel(int* counter, int* fail, int n_iter, float* sink, int fences) {
  const int nb = gridDim.x;
  float acc = 0.f;
  for (int it = 1; it <= n_iter; ++it) {
    acc += (float)it * 1e-6f;   // token work
    __syncthreads();
    if (threadIdx.x == 0) {
      if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!spin_until(counter, it * nb, 2000000)) *fail = 1;
      if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (*fail) break;
  }
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

__global__ void xcd_barrier_kernel(int* xcd_cnt /*8*16 ints*/, int* top, int* gen, int* fail, int n_iter, float* sink, int fences) {
  const int nb = gridDim.x;
  const int x = blockIdx.x & 7;
  const int per_x = (nb + 7 - x) / 8;   // blocks with b % 8 == x
  float acc = 0.f;
  for (int it = 1; it <= n_iter; ++it) {
    acc += (float)it * 1e-6f;
    __syncthreads();
    if (threadIdx.x == 0) {
      if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const int t = __hip_atomic_fetch_add(xcd_cnt + x * 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == it * per_x - 1) {   // last arriver of this XCD
        const int tt = __hip_atomic_fetch_add(top, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tt == it * 8 - 1) __hip_atomic_store(gen, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!spin_until(gen, it, 2000000)) *fail = 1;
      if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (*fail) break;
  }
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

int main() {
  int *d; float* sink;
  CHECK(hipMalloc(&d, 4096)); CHECK(hipMalloc(&sink, 4096 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int n_iter = 2000;
  for (int nb : {128, 256, 512}) for (int fences : {0, 1}) for (int kind : {0, 1}) {
    CHECK(hipMemset(d, 0, 4096));
    CHECK(hipEventRecord(e0));
    if (kind == 0) hipLaunchKernelGGL(flat_barrier_kernel, dim3(nb), dim3(256), 0, 0, d, d + 512, n_iter, sink, fences);
    else hipLaunchKernelGGL(xcd_barrier_kernel, dim3(nb), dim3(256), 0, 0, d, d + 256, d + 320, d + 512, n_iter, sink, fences);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    int fail; CHECK(hipMemcpy(&fail, d + 512, 4, hipMemcpyDeviceToHost));
    printf("%s barrier, %d blocks, fences=%d: %.2f us per barrier%s\n", kind ? "xcd " : "flat", nb, fences, ms * 1e3 / n_iter, fail ? "  (TIMEOUT)" : "");
  }
  return 0;
}
