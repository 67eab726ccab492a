// Seed of an MX-fp8 GEMM (BASELINE configs[4]): C[M][N] = sum_k A[m][k] sA[m][k/32] * W[n][k] sW[n][k/32] with e4m3
// elements and one E8M0 scale per (row, 32 consecutive k), on v_mfma_scale_f32_16x16x128_f8f6f4 with the operand / scale layout
// that tools/micro/mfma_mx_probe*.hip measured.  Deliberately naive (one wave per 16 x 16 tile, fragments straight from global
// memory): it pins the layout end to end -- row-major e4m3 matrices + [rows][K / 32] scale bytes in, fp32 out -- and gives the
// first throughput number; the product kernel would put this fragment shape behind the three-stage LDS-DMA structure of
// gemm_glds3_kernel (a K step = 128 bytes per row = ONE MFMA k, half the fragment reads per MAC of the bf16 form).
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/mx_gemm_probe.hip -o tools/micro/mx_gemm_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// 4 waves per workgroup, wave w -> tile (blockIdx.x * 4 + w) of the 16 x 16 tile grid (row-major over N tiles)
__global__ __launch_bounds__(256) void mx_gemm_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ SA,
                                                      const uint8_t* __restrict__ W, const uint8_t* __restrict__ SW,
                                                      float* __restrict__ C, int M, int N, int K) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wid, ntn = N / 16;
  if (tile >= (M / 16) * ntn) return;
  const int m0 = (tile / ntn) * 16, n0 = (tile % ntn) * 16;
  const int r = lane & 15, lg = lane >> 4;
  const uint8_t* arow = A + (long)(m0 + r) * K + lg * 16;       // bytes 0..15 of the fragment: k = lg*16 .. +15 of the 128-k step
  const uint8_t* wrow = W + (long)(n0 + r) * K + lg * 16;       // bytes 16..31: k = 64 + lg*16 .. +15
  const uint8_t* sarow = SA + (long)(m0 + r) * (K / 32) + lg;   // the scale of k block lg of this lane's row
  const uint8_t* swrow = SW + (long)(n0 + r) * (K / 32) + lg;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 128) {
    const uint4 a0 = *reinterpret_cast<const uint4*>(arow + k0), a1 = *reinterpret_cast<const uint4*>(arow + k0 + 64);
    const uint4 w0 = *reinterpret_cast<const uint4*>(wrow + k0), w1 = *reinterpret_cast<const uint4*>(wrow + k0 + 64);
    const v8i af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    const v8i wf = {(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, wf, acc, 0, 0, 0, (int)sarow[k0 / 32], 0, (int)swrow[k0 / 32]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) C[(long)(m0 + lg * 4 + q) * N + n0 + r] = acc[q];      // D[i = lg*4 + q][j = r]
}

static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  const float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main(int argc, char** argv) {
  const int M = 512, N = 384, K = 1024;
  std::vector<uint8_t> A((size_t)M * K), W((size_t)N * K), SA((size_t)M * K / 32), SW((size_t)N * K / 32);
  srand(11);
  for (auto& v : A) v = (uint8_t)(((rand() & 1) << 7) | ((3 + rand() % 7) << 3) | (rand() & 7));
  for (auto& v : W) v = (uint8_t)(((rand() & 1) << 7) | ((3 + rand() % 7) << 3) | (rand() & 7));
  for (auto& v : SA) v = (uint8_t)(122 + rand() % 8);
  for (auto& v : SW) v = (uint8_t)(123 + rand() % 6);
  uint8_t *dA, *dW, *dSA, *dSW; float* dC;
  CHECK(hipMalloc(&dA, A.size())); CHECK(hipMalloc(&dW, W.size())); CHECK(hipMalloc(&dSA, SA.size())); CHECK(hipMalloc(&dSW, SW.size()));
  CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  CHECK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dW, W.data(), W.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dSA, SA.data(), SA.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dSW, SW.data(), SW.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mx_gemm_kernel, dim3((M / 16) * (N / 16) / 4), dim3(256), 0, 0, dA, dSA, dW, dSW, dC, M, N, K);
  CHECK(hipDeviceSynchronize());
  std::vector<float> C((size_t)M * N);
  CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, mag = 0;
  for (int m = 0; m < M; m += 7)
    for (int n = 0; n < N; n += 5) {
      double ref = 0;
      for (int k = 0; k < K; ++k)
        ref += (double)e4m3(A[(size_t)m * K + k]) * ldexp(1.0, SA[(size_t)m * (K / 32) + k / 32] - 127) *
               (double)e4m3(W[(size_t)n * K + k]) * ldexp(1.0, SW[(size_t)n * (K / 32) + k / 32] - 127);
      worst = fmax(worst, fabs(C[(size_t)m * N + n] - ref));
      mag = fmax(mag, fabs(ref));
    }
  printf("MX-fp8 GEMM %d x %d x %d: max |C - ref| = %.3e at magnitude %.3e -> %s\n", M, N, K, worst, mag, worst < 1e-4 * mag ? "MATCH" : "MISMATCH");
  // throughput of the naive form at 4096^3 (fragments from global memory / L2; no LDS, no pipelining)
  const int B = 4096;
  uint8_t *bA, *bW, *bSA, *bSW; float* bC;
  CHECK(hipMalloc(&bA, (size_t)B * B)); CHECK(hipMalloc(&bW, (size_t)B * B)); CHECK(hipMalloc(&bSA, (size_t)B * B / 32)); CHECK(hipMalloc(&bSW, (size_t)B * B / 32));
  CHECK(hipMalloc(&bC, (size_t)B * B * 4));
  CHECK(hipMemset(bA, 0x38, (size_t)B * B)); CHECK(hipMemset(bW, 0x38, (size_t)B * B)); CHECK(hipMemset(bSA, 127, (size_t)B * B / 32)); CHECK(hipMemset(bSW, 127, (size_t)B * B / 32));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mx_gemm_kernel, dim3((B / 16) * (B / 16) / 4), dim3(256), 0, 0, bA, bSA, bW, bSW, bC, B, B, B);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mx_gemm_kernel, dim3((B / 16) * (B / 16) / 4), dim3(256), 0, 0, bA, bSA, bW, bSW, bC, B, B, B);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("naive MX-fp8 GEMM 4096^3: %.2f ms per launch = %.0f TFLOP/s (dense MX-fp8 peak ~5000)\n", ms / 3, 2.0 * B * B * B / (ms / 3 * 1e-3) / 1e12);
  return 0;
}
