// Probe 2: WHICH operand bytes does a lane's scale register (op_sel 0, bits 0..7) of v_mfma_scale_f32_16x16x128_f8f6f4 scale?
// (probe 1: with unit scales any lane / byte -> k placement that is the same for A and B multiplies correctly, so only the
// scale grouping is open.)  A and B are e4m3 ones; one lane L gets scale 2 (E8M0 128), everything else 1.
//   table 1: D[i][0] - 128 for every row i  -> which rows lane L's scale touches and how many of their 128 products
//   table 2: for the touched row, A is 1.0 only in (lane l2 of that row, byte octet o): D = 8 or 16 -> which bytes are scaled
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_mx_probe2.hip -o tools/micro/mfma_mx_probe2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void mx_kernel(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* d) {
  const int l = threadIdx.x;
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 0, sb[l]);
  d[l] = c;
}

static v8i *da, *db; static int *dsa, *dsb; static v4f* dd;
static void run(const uint8_t (*ha)[32], const uint8_t (*hb)[32], const int* sa, const int* sb, float D[16][16]) {
  CHECK(hipMemcpy(da, ha, 64 * 32, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 64 * 32, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  CHECK(hipDeviceSynchronize());
  float out[64][4];
  CHECK(hipMemcpy(out, dd, 64 * 16, hipMemcpyDeviceToHost));
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) D[(l >> 4) * 4 + r][l & 15] = out[l][r];
}

int main() {
  CHECK(hipMalloc(&da, 64 * 32)); CHECK(hipMalloc(&db, 64 * 32)); CHECK(hipMalloc(&dsa, 256)); CHECK(hipMalloc(&dsb, 256)); CHECK(hipMalloc(&dd, 64 * 16));
  uint8_t ones[64][32], ha[64][32];
  memset(ones, 0x38, sizeof(ones));       // e4m3 1.0
  int s1[64], sa[64];
  for (int l = 0; l < 64; ++l) s1[l] = 127;
  float D[16][16];
  const int Ls[] = {0, 1, 5, 16, 17, 32, 48, 63};
  for (int which = 0; which < 2; ++which) {       // 0: the A-side scale register, 1: the B-side one
    printf("== scale register of operand %c ==\n", which ? 'B' : 'A');
    for (int li = 0; li < 8; ++li) {
      const int L = Ls[li];
      memcpy(sa, s1, sizeof(sa));
      sa[L] = 128;
      run(ones, ones, which ? s1 : sa, which ? sa : s1, D);
      printf("lane %2d scale x2: extra products per %s:", L, which ? "COLUMN j (row 0)" : "ROW i (column 0)");
      int touched = -1;
      for (int i = 0; i < 16; ++i) {
        const float v = (which ? D[0][i] : D[i][0]) - 128.f;
        printf(" %g", v);
        if (v != 0.f) touched = i;
      }
      printf("\n");
      if (touched < 0) continue;
      printf("         which bytes of row/col %d it scales (lane of that row : octets 0-3, x = scaled):", touched);
      for (int lg = 0; lg < 4; ++lg) {
        const int l2 = lg * 16 + touched;
        printf("  lane %2d:", l2);
        for (int o = 0; o < 4; ++o) {
          memset(ha, 0, sizeof(ha));
          for (int j = 0; j < 8; ++j) ha[l2][o * 8 + j] = 0x38;
          run(which ? ones : ha, which ? ha : ones, which ? s1 : sa, which ? sa : s1, D);
          const float v = which ? D[0][touched] : D[touched][0];
          printf("%c", v == 16.f ? 'x' : (v == 8.f ? '.' : '?'));
        }
      }
      printf("\n");
    }
  }
  return 0;
}
