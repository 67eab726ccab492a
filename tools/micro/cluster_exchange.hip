// Micro-benchmark: cost and correctness of an in-kernel activation exchange inside a CLUSTER of workgroups that the
// dispatcher places on one XCD (workgroup id % 8 == cluster id).  This is the primitive a persistent "row-cluster"
// decode kernel would use instead of a kernel boundary: every member writes its 64-column slice of R activation rows,
// arrives at the cluster's counter, waits for the other 11 members, reads all R x 768 values back.
// Variants of the memory protocol (all spins bounded):
//   0  agent-scope relaxed atomic stores / loads for the data (write-through, cache-bypassing), relaxed counter
//   1  plain stores + release fence (agent), counter, acquire fence (agent) + plain 16-byte loads
//   2  agent-scope atomic stores (write-through) for the data, counter, acquire fence (agent) + plain 16-byte loads
//   3  plain stores + workgroup-scope fence, counter, sc0 ("group scope": L1 bypass) loads -- valid only if all members
//      share an L2, i.e. sit on one XCD
// Output: microseconds per exchange, number of stale / wrong values read, XCD ids seen per cluster.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NCL = 8, NMEM = 12, R = 4, D = 768;

__device__ inline bool spin_until(const int* p, int target, int max_iter) {
  for (int i = 0; i < max_iter; ++i) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
  }
  return false;
}
__device__ inline float value_of(int it, int row, int col) { return (float)((it * 131 + row * 17 + col) & 0xffff); }

template <int MODE, int WLOADS>
__global__ __launch_bounds__(1024) void exchange_kernel(float* xch /*[2][NCL][R][D]*/, int* counters /*[NCL][16]*/, int* fail,
                                                        int* errors, int* xcc_seen /*[NCL]*/, const uint4* w, int n_iter,
                                                        float* sink) {
  const int cl = blockIdx.x % NCL, mem = blockIdx.x / NCL;
  const int tid = threadIdx.x;
  if (tid == 0) {
    const int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15;   // HW_REG_XCC_ID, 4 bits
    atomicOr(xcc_seen + cl, 1 << xcc);
  }
  int* cnt = counters + cl * 16;
  unsigned acc = 0;
  int nerr = 0;
  for (int it = 0; it < n_iter; ++it) {
    float* buf = xch + ((long)(it & 1) * NCL + cl) * R * D;
    // independent "weight" loads of the next phase, requested before the wait
    uint4 wv[WLOADS > 0 ? WLOADS : 1];
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) wv[i] = w[((long)(it % 12) * 96 + blockIdx.x) * WLOADS * 1024 + i * 1024 + tid];
    // produce: R rows x 64 columns of this member
    if (tid < R * 64) {
      const int row = tid >> 6, col = mem * 64 + (tid & 63);
      const float v = value_of(it, row, col);
      if (MODE == 0 || MODE == 2) __hip_atomic_store(buf + row * D + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else buf[row * D + col] = v;
    }
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!spin_until(cnt, (it + 1) * NMEM, 4000000)) *fail = 1;
    }
    __syncthreads();
    if (*fail) break;
    // consume: all R x D values (3 floats per thread)
    if (MODE == 1 || MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int idx = j * 1024 + tid;
        const float v = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nerr += (v != value_of(it, idx / D, idx % D));
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int idx = j * 1024 + tid;
        const float v = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        nerr += (v != value_of(it, idx / D, idx % D));
      }
    } else {
      if (tid < R * D / 4) {
        typedef __attribute__((ext_vector_type(4))) float f4_t;
        const f4_t v4 = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(buf + tid * 4));
        const float4 v = make_float4(v4[0], v4[1], v4[2], v4[3]);
        const int idx = tid * 4;
        nerr += (v.x != value_of(it, idx / D, idx % D)) + (v.w != value_of(it, (idx + 3) / D, (idx + 3) % D));
      }
    }
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) acc += wv[i].x ^ wv[i].w;
  }
  if (nerr) atomicAdd(errors, nerr);
  if (tid == 0) sink[blockIdx.x] = (float)acc;
}

template <int MODE, int WLOADS>
static void run(const char* name, float* xch, int* ctl, const uint4* w, float* sink) {
  const int n_iter = 3000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipMemset(ctl, 0, 4096));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((exchange_kernel<MODE, WLOADS>), dim3(NCL * NMEM), dim3(1024), 0, 0, xch, ctl, ctl + 512, ctl + 513, ctl + 520, w, n_iter, sink);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  int h[1024]; CHECK(hipMemcpy(h, ctl, 4096, hipMemcpyDeviceToHost));
  printf("%-64s %6.2f us per exchange, wrong values %d%s, XCD masks:", name, ms * 1e3 / n_iter, h[513], h[512] ? " (TIMEOUT)" : "");
  for (int c = 0; c < NCL; ++c) printf(" %x", h[520 + c]);
  printf("\n");
}

int main() {
  float* xch; int* ctl; uint4* w; float* sink;
  CHECK(hipMalloc(&xch, 2L * NCL * R * D * 4)); CHECK(hipMemset(xch, 0, 2L * NCL * R * D * 4));
  CHECK(hipMalloc(&ctl, 4096));
  CHECK(hipMalloc(&w, 12L * 96 * 18 * 1024 * 16)); CHECK(hipMemset(w, 1, 12L * 96 * 18 * 1024 * 16));
  CHECK(hipMalloc(&sink, 4096));
  run<0, 0>("0: agent atomic stores / loads", xch, ctl, w, sink);
  run<1, 0>("1: plain stores, release + acquire fences (agent)", xch, ctl, w, sink);
  run<2, 0>("2: agent atomic stores, acquire fence + 16-byte loads", xch, ctl, w, sink);
  run<3, 0>("3: plain stores, workgroup fence, sc0 loads (one XCD only)", xch, ctl, w, sink);
  run<0, 6>("0 + 98 KB of weights per member requested before the wait", xch, ctl, w, sink);
  run<2, 6>("2 + 98 KB of weights", xch, ctl, w, sink);
  run<3, 6>("3 + 98 KB of weights", xch, ctl, w, sink);
  run<3, 18>("3 + 295 KB of weights", xch, ctl, w, sink);
  return 0;
}
