// Micro-benchmark: HBM write bandwidth of the GEMM epilogue's store patterns.  184 MB (40032 x 2304 bf16, the encoder's QKV output)
// written by 256-row x 256-column tiles, one 512-thread workgroup per tile as gemm_glds4_kernel does (8 waves of 128 x 64):
//   0: today's pattern -- lane (l15, lg) stores 8 bytes at (row l15 + 16 i, col jj*16 + lg*4): 16 rows x 32 B per instruction
//   1: 16 bytes per lane, 64 B contiguous per row and instruction (what a permlane16 swap of two column blocks gives)
//   2: full 128-byte row segments: 8 lanes x 16 B per row, 8 rows per instruction
//   3: as 2 with non-temporal stores
// hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/micro/store_pattern_probe.hip && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* C, int M, int N, int nbn, unsigned seed) {
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 2, wc = wid & 3;
  const int l15 = lane & 15, lg = lane >> 4;
  const int m0 = bm * 256 + wr * 128, n0 = bn * 256 + wc * 64;
  uint4 v = make_uint4(seed + lane, seed ^ lane, seed * 3u, seed + 7u);
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int row = m0 + i * 16 + l15, col = n0 + jj * 16 + lg * 4;
        if (row < M && col < N) *reinterpret_cast<uint2*>(C + (long)row * N + col) = make_uint2(v.x + i, v.y + jj);
      }
  } else if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        // lanes lg = 0 / 2: block 2 jp, columns 0-7 / 8-15; lg = 1 / 3: block 2 jp + 1
        const int row = m0 + i * 16 + l15, col = n0 + (2 * jp + (lg & 1)) * 16 + (lg >> 1) * 8;
        if (row < M && col < N) *reinterpret_cast<uint4*>(C + (long)row * N + col) = make_uint4(v.x + i, v.y + jp, v.z, v.w);
      }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + i * 8 + (lane >> 3), col = n0 + (lane & 7) * 8;
      if (row < M && col < N) {
        uint4* dst = reinterpret_cast<uint4*>(C + (long)row * N + col);
        const uint4 x = make_uint4(v.x + i, v.y, v.z, v.w);
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 xv = {x.x, x.y, x.z, x.w};
        if (MODE == 3) __builtin_nontemporal_store(xv, reinterpret_cast<u32x4*>(dst));
        else *dst = x;
      }
    }
  }
}

template <int MODE>
static void run(unsigned short* C, int M, int N, const char* what) {
  const int nbm = (M + 255) / 256, nbn = (N + 255) / 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(store_kernel<MODE>, dim3(nbm * nbn), dim3(512), 0, 0, C, M, N, nbn, 1u + w);
  CHECK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(store_kernel<MODE>, dim3(nbm * nbn), dim3(512), 0, 0, C, M, N, nbn, 11u + r);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = (double)M * N * 2;
  printf("  %-58s %8.1f us  %6.2f TB/s\n", what, us, bytes / us / 1e6);
}

int main() {
  const int shapes[3][2] = {{40032, 2304}, {40032, 2048}, {40032, 768}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1];
    unsigned short* C;
    CHECK(hipMalloc(&C, (size_t)M * N * 2));
    printf("M %d N %d bf16 (%.0f MB)\n", M, N, (double)M * N * 2 / 1e6);
    run<0>(C, M, N, "8 B per lane, 16 rows x 32 B per instruction (today)");
    run<1>(C, M, N, "16 B per lane, 16 rows x 64 B per instruction");
    run<2>(C, M, N, "16 B per lane, 8 rows x 128 B per instruction");
    run<3>(C, M, N, "16 B per lane, 8 rows x 128 B, non-temporal");
    CHECK(hipFree(C));
  }
  return 0;
}
