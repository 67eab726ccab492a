// Cross-file internal entry points of libmapperhip (C++ side of the C ABI in include/mapperhip.h).
#pragma once
#include "common.hpp"

namespace mh {

int gemm_prepare();  // one-time kernel attribute setup; call before any stream capture
int gemm(const MhGemm& g, hipStream_t s);
int rmsnorm(const float* x, int ldx, const float* w, void* y, int ldy, int rows, int d, float eps, int out_dtype,
            hipStream_t s);
int ln_modulate(const float* x, int ldx, const float* shift, const float* scale, int mod_ld, int rows_per_batch,
                float* y, int ldy, int rows, int d, float eps, hipStream_t s);
int attention(const void* qk, int ld_qk, int k_col0, const void* vt, int Lpad, const float* bias, void* out,
              int ld_out, int B, int L, int H, float scale, int band, int dtype, hipStream_t s);

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// bump allocator over a caller-provided workspace
struct Arena {
  char* base;
  int64_t size, off;
  Arena(void* p, int64_t n) : base((char*)p), size(n), off(0) {}
  void* take(int64_t bytes) {
    int64_t o = off;
    off = align256(off + bytes);
    return (base && off <= size) ? base + o : nullptr;
  }
  bool ok() const { return off <= size; }
};

}  // namespace mh
