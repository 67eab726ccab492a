// Cross-file internal entry points of libmapperhip (C++ side of the C ABI in include/mapperhip.h).
#pragma once
#include "common.hpp"

namespace mh {

int gemm_prepare();  // one-time kernel attribute setup; call before any stream capture
// ascending_k: never pick the 16x16 split-K tile, so the fp32 summation order (k ascending, one accumulator) and with it
// every bit of the result is independent of M (the tile choice otherwise follows the grid size)
int gemm(const MhGemm& g, hipStream_t s, bool ascending_k = false);
int rmsnorm(const float* x, int ldx, const float* w, void* y, int ldy, int rows, int d, float eps, int out_dtype,
            hipStream_t s);
int layernorm(const float* x, int ldx, const float* w, const float* b, void* y, int ldy, int rows, int d, float eps, int out_dtype,
              hipStream_t s);
// MX-fp8 operands (mx8.hip): scale bytes per row, producers that write the quantised operand directly
int quantize_mx8(const void* x, int ldx, int rows, int K, int in_dtype, uint8_t* q, int ldq, uint8_t* scales, hipStream_t s);
int rmsnorm_mx8(const float* x, int ldx, const float* w, int rows, int d, float eps, int round_dtype, uint8_t* q, int ldq,
                uint8_t* scales, hipStream_t s);
int ln_modulate_mx8(const float* x, int ldx, const float* shift, const float* scale, int mod_ld, int rows_per_batch, int rows, int d,
                    float eps, uint8_t* q, int ldq, uint8_t* scales, hipStream_t s);
constexpr int MH_LN_SPLIT3 = 2;   // ln_modulate out_dtype: fp32 values stored pre-split for gemm_s3g_kernel
int ln_modulate(const float* x, int ldx, const float* shift, const float* scale, int mod_ld, int rows_per_batch,
                void* y, int ldy, int rows, int d, float eps, int out_dtype, hipStream_t s);
int attention(const void* qk, int ld_qk, int k_col0, const void* vt, int Lpad, const float* bias, void* out,
              int ld_out, int B, int L, int H, float scale, int band, int dtype, hipStream_t s, int open_from = 0,
              int out_split3 = 0);

// strided description of one attention problem (all byte strides; see attention.hip)
struct AttnArgs {
  const void* q; long q_rs, q_bs;            // query rows: + b*q_bs + row*q_rs + h*64*sizeof(T)
  const void* k; long k_rs, k_bs, k_hs;      // key rows:   + b*k_bs + h*k_hs + key*k_rs
  const void* vt; long vt_bs, vt_hs; int Lkpad;   // V^T [..][64][Lkpad]: + b*vt_bs + h*vt_hs + d*Lkpad*sizeof(T)
  const float* bias; long bias_hs; int bias_center, bias_sign, bias_min, bias_max;   // bias[h*hs + center + clamp(sign*(k-q))]
  const uint8_t* key_mask; int mask_ld, mask_len;   // [B][mask_ld], 1 = attend, keys >= mask_len always attend
  void* out; long out_rs, out_bs;
  int Lq, Lk;
  float scale;
  int open_from = 0;   // with a band: positions >= open_from (padding) attend and are attended by everything (0 = none)
  int out_split3 = 0;  // fp32 only: the output is written pre-split ([32 x bf16 hi | 32 x bf16 lo] per 32 values), MhGemm.w_split3 & 2
  int band;        // > 0: attend iff -(band-1) <= k - q <= band
  int causal;      // attend iff key position <= q_pos0 + query index
  int q_pos0;
};
int attention_general(const AttnArgs& a, int B, int H, int dtype, hipStream_t s);
int transpose_v(const void* v, long v_bs_el, long v_hs_el, int Lk, void* vt, int Lkpad, int B, int H, int dtype,
                hipStream_t s);

int slider_project(float* x0, const uint8_t* imask, const float* iref, int N, int T, const MhSliderSet& ss, hipStream_t s);
int check_slider_set(const MhSliderSet* ss, int N);

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
