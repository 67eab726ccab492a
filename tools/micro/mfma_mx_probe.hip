// Probe: operand layout and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950, fp8 e4m3 x fp8 e4m3).
// Groundwork for an MX-fp8 GEMM (BASELINE configs[4] asks for fp8 MFMA).  The guides name the builtin but not
// which k indices a lane's 32 operand bytes hold, nor what the scale operand scales.  One wave multiplies random e4m3
// matrices A [16][128], B [16][128] (D = A B^T) under several layout hypotheses and compares with a host product:
//   H0: lane l holds row l & 15, k = (l >> 4) * 32 .. + 32 (32 consecutive bytes)
//   H1: lane l holds row l & 15, k = j * 32 + (l >> 4) * 8 .. + 8 for j = 0..3 (four 16x16x32-style blocks)
//   H2: lane l holds row l & 15, bytes 0..15 = k (l >> 4) * 16 .. + 16, bytes 16..31 = k 64 + (l >> 4) * 16 .. + 16
//       (what mfma_mx_probe2 found: the scale of lane (row, lg) applies to k block lg of that row, and k block q lives in
//       bytes 16 (q / 2) .. + 16 of the two lanes lg = 2 (q % 2), 2 (q % 2) + 1)
// scale byte (E8M0, value 2^(s - 127)) taken from bits 0..7 of the scale register (op_sel 0):
//   S0: the scale of lane l multiplies that lane's 32 elements (MX block of 32 along k)
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_mx_probe.hip -o tools/micro/mfma_mx_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void mx_kernel(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* d) {
  const int l = threadIdx.x;
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0 /* A: fp8 e4m3 */, 0 /* B: fp8 e4m3 */, 0, sa[l], 0, sb[l]);
  d[l] = c;
}

static float e4m3(uint8_t v) {   // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main() {
  uint8_t A[16][128], B[16][128];
  srand(7);
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 128; ++k) {
      A[i][k] = (uint8_t)(((rand() & 1) << 7) | ((4 + rand() % 6) << 3) | (rand() & 7));   // |x| in [2^-3, 2^2)
      B[i][k] = (uint8_t)(((rand() & 1) << 7) | ((4 + rand() % 6) << 3) | (rand() & 7));
    }
  for (int use_scales = 0; use_scales < 2; ++use_scales)
    for (int hyp = 0; hyp < 3; ++hyp) {
      uint8_t ha[64][32], hb[64][32];
      int sa[64], sb[64];
      uint8_t sca[16][4], scb[16][4];                       // E8M0 per (row, 32-k block)
      for (int i = 0; i < 16; ++i)
        for (int q = 0; q < 4; ++q) { sca[i][q] = use_scales ? 124 + (rand() % 6) : 127; scb[i][q] = use_scales ? 125 + (rand() % 5) : 127; }
      for (int l = 0; l < 64; ++l) {
        const int row = l & 15, lg = l >> 4;
        for (int j = 0; j < 32; ++j) {
          const int k = hyp == 0 ? lg * 32 + j : hyp == 1 ? (j / 8) * 32 + lg * 8 + (j % 8) : (j / 16) * 64 + lg * 16 + (j % 16);
          ha[l][j] = A[row][k]; hb[l][j] = B[row][k];
        }
        sa[l] = sca[row][lg]; sb[l] = scb[row][lg];          // lane (row, lg) carries the scale of k block lg of its row
      }
      v8i *da, *db; int *dsa, *dsb; v4f* dd;
      CHECK(hipMalloc(&da, 64 * 32)); CHECK(hipMalloc(&db, 64 * 32)); CHECK(hipMalloc(&dsa, 256)); CHECK(hipMalloc(&dsb, 256)); CHECK(hipMalloc(&dd, 64 * 16));
      CHECK(hipMemcpy(da, ha, 64 * 32, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 64 * 32, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
      CHECK(hipDeviceSynchronize());
      float out[64][4];
      CHECK(hipMemcpy(out, dd, 64 * 16, hipMemcpyDeviceToHost));
      // reference: D[i][j] = sum_k A[i][k] sA[i][k/32] * B[j][k] sB[j][k/32]; C/D layout col = l & 15, row = (l >> 4) * 4 + r
      double worst = 0, worst_t = 0, mag = 0;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
          const int i = (l >> 4) * 4 + r, j = l & 15;
          double ref = 0, ref_t = 0;
          for (int k = 0; k < 128; ++k) {
            const double sa_ = ldexp(1.0, sca[i][k / 32] - 127), sb_ = ldexp(1.0, scb[j][k / 32] - 127);
            ref += (double)e4m3(A[i][k]) * sa_ * (double)e4m3(B[j][k]) * sb_;
            const double sat = ldexp(1.0, sca[j][k / 32] - 127), sbt = ldexp(1.0, scb[i][k / 32] - 127);
            ref_t += (double)e4m3(A[j][k]) * sat * (double)e4m3(B[i][k]) * sbt;       // (transposed C/D layout, for completeness)
          }
          worst = fmax(worst, fabs(out[l][r] - ref));
          mag = fmax(mag, fabs(ref));
          worst_t = fmax(worst_t, fabs(out[l][r] - ref_t));
        }
      printf("scales %s, operand layout H%d: max |D - ref| = %.3e (C/D col = lane & 15), %.3e (transposed)  -> %s\n",
             use_scales ? "random 2^-3..2^2" : "all 1", hyp, worst, worst_t, worst < 1e-5 * mag ? "MATCH" : (worst_t < 1e-5 * mag ? "MATCH (transposed D)" : "no"));   // (fp32 accumulation of 128 products)
      (void)hipFree(da); (void)hipFree(db); (void)hipFree(dsa); (void)hipFree(dsb); (void)hipFree(dd);
    }
  return 0;
}
