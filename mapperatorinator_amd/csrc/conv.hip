// K2  Whisper-style audio front-end: conv1d(k3, p1) + GELU -> conv1d(k3, s2, p1) + GELU (+ fixed positions).
//   HF WhisperEncoder.forward (transformers 4.57.3 models/whisper/modeling_whisper.py, "inputs_embeds =
//   nn.functional.gelu(self.conv1(input_features))" ...), identical in the reference's forks:
//   osuT5/osuT5/model/custom_transformers/modeling_varwhisper.py:779-780,813-816.
// Each k=3 convolution is an im2col gather (HBM-bound, time-major activations so that the three taps of a
// position are three contiguous rows) followed by the MFMA GEMM with a fused bias + exact-erf GELU (+ position
// table) epilogue.
#include "internal.hpp"

namespace mh {
namespace {

#define MH_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != MH_OK) return _rc; \
  } while (0)

// A[(b, t), k*C + c] = x[b][t*stride + k - 1][c]  (zero outside [0, Lin)), columns [3C, Kpad) zero
template <typename T>
__global__ __launch_bounds__(256) void im2col3_kernel(const T* __restrict__ x, int Lin, int C, int Lout, int stride,
                                                     T* __restrict__ A, int Kpad) {
  const int row = blockIdx.x;              // b*Lout + t
  const int b = row / Lout, t = row - b * Lout;
  T* ar = A + (long)row * Kpad;
  for (int j = threadIdx.x; j < Kpad; j += 256) {
    T v = T(0);
    if (j < 3 * C) {
      const int k = j / C, c = j - k * C;
      const int ti = t * stride + k - 1;
      if (ti >= 0 && ti < Lin) v = x[((long)b * Lin + ti) * C + c];
    }
    ar[j] = v;
  }
}

// frames[(b, t)][col0 + j] = cond[b][j]: the wrapper's conditioning vectors repeated over the frames of their chunk and
// concatenated to the mel channels (modeling_mapperatorinator.py:201-202) -- into the K-padded frame buffer mh_mel wrote
template <typename T>
__global__ __launch_bounds__(256) void cond_channels_kernel(T* __restrict__ frames, int L, int ld, int col0, const float* __restrict__ cond, int n_cond) {
  const long row = blockIdx.x;
  const float* c = cond + (row / L) * n_cond;
  T* dst = frames + row * ld + col0;
  for (int j = threadIdx.x; j < n_cond; j += 256) dst[j] = Elem<T>::from_f32(c[j]);
}

}  // namespace
}  // namespace mh

using namespace mh;

extern "C" int mh_cond_channels(void* frames, int B, int L, int ld, int col0, const float* cond, int n_cond, int dtype, void* stream) {
  MH_REQUIRE(frames && cond && B > 0 && L > 0 && n_cond > 0 && col0 >= 0 && col0 + n_cond <= ld, "mh_cond_channels: bad arguments");
  MH_REQUIRE(dtype == MH_F32 || dtype == MH_BF16, "mh_cond_channels: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MH_BF16) hipLaunchKernelGGL(cond_channels_kernel<bf16_t>, dim3(B * L), dim3(256), 0, s, (bf16_t*)frames, L, ld, col0, cond, n_cond);
  else hipLaunchKernelGGL(cond_channels_kernel<float>, dim3(B * L), dim3(256), 0, s, (float*)frames, L, ld, col0, cond, n_cond);
  return check_launch("cond_channels_kernel");
}

extern "C" int64_t mh_whisper_frontend_workspace_bytes(int B, int Lin, int C, int d, int dtype) {
  if (B <= 0 || Lin <= 0 || C <= 0 || d <= 0) return -1;
  const int64_t es = dtype == MH_BF16 ? 2 : 4;
  const int Lout = (Lin - 1) / 2 + 1;
  return align256((int64_t)B * Lin * round_up(3 * C, 32) * es) + align256((int64_t)B * Lin * d * es) +
         align256((int64_t)B * Lout * round_up(3 * d, 32) * es);
}

extern "C" int mh_whisper_frontend(const void* x, int B, int Lin, int C, const void* w1, const float* b1, const void* w2,
                                   const float* b2, const float* pos, int d, void* out, void* workspace,
                                   int64_t workspace_bytes, int dtype, void* stream) {
  MH_REQUIRE(x && w1 && b1 && w2 && b2 && out && workspace, "mh_whisper_frontend: null argument");
  MH_REQUIRE(dtype == MH_F32 || dtype == MH_BF16, "mh_whisper_frontend: bad dtype");
  MH_REQUIRE(B > 0 && Lin > 0 && C > 0 && d > 0 && d % 8 == 0, "mh_whisper_frontend: bad shape");
  MH_REQUIRE(workspace_bytes >= mh_whisper_frontend_workspace_bytes(B, Lin, C, d, dtype), "mh_whisper_frontend: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int es = dtype == MH_BF16 ? 2 : 4;
  const int Lout = (Lin - 1) / 2 + 1, K1 = round_up(3 * C, 32), K2 = round_up(3 * d, 32);
  Arena ar(workspace, workspace_bytes);
  void* A1 = ar.take((int64_t)B * Lin * K1 * es);
  void* y1 = ar.take((int64_t)B * Lin * d * es);
  void* A2 = ar.take((int64_t)B * Lout * K2 * es);
  if (dtype == MH_BF16)
    hipLaunchKernelGGL(im2col3_kernel<bf16_t>, dim3(B * Lin), dim3(256), 0, s, (const bf16_t*)x, Lin, C, Lin, 1, (bf16_t*)A1, K1);
  else
    hipLaunchKernelGGL(im2col3_kernel<float>, dim3(B * Lin), dim3(256), 0, s, (const float*)x, Lin, C, Lin, 1, (float*)A1, K1);
  MH_TRY(check_launch("im2col3_kernel"));
  MhGemm g = MhGemm{};
  g.A = A1; g.lda = K1; g.W = w1; g.ldw = K1; g.C = y1; g.ldc = d; g.M = B * Lin; g.N = d; g.K = K1; g.bias = b1;
  g.dtype = dtype; g.epilogue = MH_EPI_BIAS_GELU_ERF;
  MH_TRY(gemm(g, s));
  if (dtype == MH_BF16)
    hipLaunchKernelGGL(im2col3_kernel<bf16_t>, dim3(B * Lout), dim3(256), 0, s, (const bf16_t*)y1, Lin, d, Lout, 2, (bf16_t*)A2, K2);
  else
    hipLaunchKernelGGL(im2col3_kernel<float>, dim3(B * Lout), dim3(256), 0, s, (const float*)y1, Lin, d, Lout, 2, (float*)A2, K2);
  MH_TRY(check_launch("im2col3_kernel"));
  g = MhGemm{};
  g.A = A2; g.lda = K2; g.W = w2; g.ldw = K2; g.C = out; g.ldc = d; g.M = B * Lout; g.N = d; g.K = K2; g.bias = b2;
  g.gate = pos; g.gate_ld = d; g.rows_per_batch = Lout; g.dtype = dtype; g.epilogue = MH_EPI_BIAS_GELU_ERF;
  return gemm(g, s);
}
