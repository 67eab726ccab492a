// osuT5 on MI355X: encoder stack, cross-attention K/V projection and the KV-cached AR decode loop.
// Orchestration lives here (C++), so that the per-token loop never returns to Python: every decode
// step is one hipGraph replay; the host only polls a 4-byte "rows still running" word.
//
// Reference semantics reproduced (see include/mapperhip.h for the per-entry citations):
//   HF T5Stack / T5Block / T5Attention / T5LayerFF (pre-norm residual blocks, shared layer-0 relative
//   bias, no 1/sqrt(d) scaling, gated gelu_new FFN), HF GenerationMixin._sample under
//   model_generate (osuT5/osuT5/inference/server.py:83-156) with the reference logits processors.
#include <stdlib.h>

#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "decode_kernels.hpp"

namespace mh {
namespace {

inline int es_of(int dtype) { return dtype == MH_BF16 ? 2 : 4; }

int check_cfg(const MhT5Config* c, const char* who) {
  MH_REQUIRE(c, "%s: null config", who);
  MH_REQUIRE(c->d_kv == 64, "%s: d_kv must be 64 (got %d)", who, c->d_kv);
  MH_REQUIRE(c->n_enc_layers <= MH_MAX_LAYERS && c->n_dec_layers <= MH_MAX_LAYERS, "%s: too many layers", who);
  MH_REQUIRE(c->d_model % 32 == 0 && c->d_ff % 32 == 0, "%s: d_model/d_ff must be multiples of 32", who);
  MH_REQUIRE(c->n_mels_pad % 32 == 0 && c->n_mels_pad >= c->n_mels, "%s: bad n_mels_pad", who);
  MH_REQUIRE(c->dtype == MH_F32 || c->dtype == MH_BF16, "%s: bad dtype", who);
  MH_REQUIRE(c->arch >= 0 && c->arch <= 2, "%s: arch %d is none of 0 (T5), 1 (VarWhisper / RoPEWhisper), 2 (HF Whisper)", who, c->arch);
  MH_REQUIRE(c->dec_pos_from_mask == 0 || c->arch == 2, "%s: dec_pos_from_mask belongs to arch 2 (absolute decoder positions)", who);
  if (c->arch >= 1) {
    MH_REQUIRE(c->d_model == c->n_heads * 64, "%s: the Whisper family needs d_model = 64 heads", who);
    MH_REQUIRE(c->in_frames >= 1 && c->src_len == (c->in_frames - 1) / 2 + 1, "%s: src_len must be the conv-strided in_frames", who);
    MH_REQUIRE(c->attn_scale > 0.f, "%s: attn_scale must be positive", who);
  }
  MH_REQUIRE(c->enc_operand_dtype == 0 || c->enc_operand_dtype == MH_MX8, "%s: enc_operand_dtype must be 0 or MH_MX8", who);
  if (c->enc_operand_dtype == MH_MX8)
    MH_REQUIRE(c->dtype == MH_BF16 && c->arch == 0 && c->d_model % 128 == 0 && c->d_ff % 128 == 0,
               "%s: MX-fp8 encoder operands need bf16 storage, the T5 backbone and d_model / d_ff multiples of 128", who);
  return MH_OK;
}
inline bool enc_mx(const MhT5Config* c) { return c->enc_operand_dtype == MH_MX8; }
inline int64_t mx_operand_bytes(int64_t rows, int K) { return align256(rows * K) + align256(rows * mx8_scale_row_bytes(K)); }
// the pre-norm of a block: T5LayerNorm / nn.RMSNorm (arch 0 / 1) or the affine nn.LayerNorm of HF Whisper (arch 2, bias != NULL there)
inline int pre_norm(const MhT5Config* c, const float* x, const float* w, const float* b, void* y, int rows, int out_dtype, hipStream_t s) {
  if (c->arch == 2) {
    MH_REQUIRE(b, "arch 2 needs the LayerNorm biases (MhT5Weights.*_ln*_b)");
    return layernorm(x, c->d_model, w, b, y, c->d_model, rows, c->d_model, c->eps, out_dtype, s);
  }
  return rmsnorm(x, c->d_model, w, y, c->d_model, rows, c->d_model, c->eps, out_dtype, s);
}
inline bool is_local_layer(const MhT5Config* c, int l) { return c->arch == 1 && c->local_every > 1 && c->local_window > 0 && l % c->local_every != 0; }

#define MH_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != MH_OK) return _rc; \
  } while (0)

}  // namespace
}  // namespace mh

using namespace mh;

// ------------------------------------------------------------------------------------------------
// encoder
// ------------------------------------------------------------------------------------------------
extern "C" int64_t mh_t5_encode_workspace_bytes(const MhT5Config* c, int B) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  if (!c || B <= 0) return -1;
  const int64_t rows = (int64_t)B * c->src_len, es = es_of(c->dtype);
  const int inner = c->n_heads * 64, Lpad = round_up(c->src_len, 64);
  int64_t t = 0;
  if (c->arch >= 1)   // front-end scratch + its output (storage type) in front of the layer buffers
    t += align256(mh_whisper_frontend_workspace_bytes(B, c->in_frames, c->arch == 2 ? c->d_model : c->n_mels_pad, c->d_model, c->dtype)) +
         align256(rows * c->d_model * es);
  if (c->arch == 2)   // encoder_embedder output in front of conv1: fp32 accumulator rows + their storage-typed copy
    t += align256((int64_t)B * c->in_frames * c->d_model * 4) + align256((int64_t)B * c->in_frames * c->d_model * es);
  t += align256(rows * c->d_model * 4);                    // h
  t += align256(rows * c->d_model * es);                   // n
  t += align256(rows * 2 * inner * es);                    // qk
  t += align256((int64_t)B * inner * Lpad * es);           // vt
  t += align256(rows * inner * es);                        // attn
  t += align256(rows * c->d_ff * es);                      // ff
  if (c->enc_operand_dtype == MH_MX8)                      // MX-fp8 copy of the current GEMM's A operand (widest: d_ff columns)
    t += mx_operand_bytes(rows, c->d_ff > c->d_model ? c->d_ff : c->d_model);
  return t;
}

namespace mh {
namespace {
// fp32 residual stream <- storage-typed rows (the conv front-end's output)
template <typename T>
__global__ __launch_bounds__(256) void rows_to_f32_kernel(const T* __restrict__ x, float* __restrict__ h, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    *reinterpret_cast<float4*>(h + i) = make_float4(Elem<T>::to_f32(x[i]), Elem<T>::to_f32(x[i + 1]), Elem<T>::to_f32(x[i + 2]), Elem<T>::to_f32(x[i + 3]));
  } else {
    for (long j = i; j < n; ++j) h[j] = Elem<T>::to_f32(x[j]);
  }
}
// storage-typed rows <- fp32 rows (arch 2: encoder_embedder's output becomes conv1's operand)
template <typename T>
__global__ __launch_bounds__(256) void f32_to_rows_kernel(const float* __restrict__ h, T* __restrict__ x, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(h + i);
    x[i] = Elem<T>::from_f32(v.x); x[i + 1] = Elem<T>::from_f32(v.y); x[i + 2] = Elem<T>::from_f32(v.z); x[i + 3] = Elem<T>::from_f32(v.w);
  } else {
    for (long j = i; j < n; ++j) x[j] = Elem<T>::from_f32(h[j]);
  }
}
// rotate-half RoPE in place on the q | k block of a QKV GEMM output (apply_rotary_pos_emb,
// osuT5/osuT5/model/custom_transformers/modeling_varwhisper.py:229-258): row r = b * L + position, 2H head slots of 64
// columns; one thread rotates 8 pairs (i, i + 32): two 16-byte (bf16) / four (fp32) accesses each way.
// rope fp32 [L][64] = cos(32) | sin(32).
template <typename T>
__global__ __launch_bounds__(256) void rope_qk_kernel(T* __restrict__ qk, int ld, long rows, int L, int slots, const float* __restrict__ rope) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // (row, slot, chunk of 8 pairs)
  const long total = rows * slots * 4;
  if (idx >= total) return;
  const int c = (int)(idx & 3);
  const long rs = idx >> 2;
  const int slot = (int)(rs % slots);
  const long row = rs / slots;
  const int pos = (int)(row % L);
  T* x1 = qk + row * ld + slot * 64 + c * 8;
  T* x2 = x1 + 32;
  const float* cs = rope + (long)pos * 64 + c * 8;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = Elem<T>::to_f32(x1[i]); b[i] = Elem<T>::to_f32(x2[i]); }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float co = cs[i], si = cs[32 + i];
    x1[i] = Elem<T>::from_f32(a[i] * co - b[i] * si);
    x2[i] = Elem<T>::from_f32(b[i] * co + a[i] * si);
  }
}

// rotate-half RoPE in place on the cached keys of positions 0 .. np-1: kc [BH][tgt][64], one thread per (bh, pos, 8 pairs)
template <typename T>
__global__ __launch_bounds__(256) void rope_cache_kernel(T* __restrict__ kc, long BH, int tgt, int np, const float* __restrict__ rope) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= BH * np * 4) return;
  const int c = (int)(idx & 3);
  const long rp = idx >> 2;
  const int pos = (int)(rp % np);
  const long bh = rp / np;
  T* x1 = kc + (bh * tgt + pos) * 64 + c * 8;
  T* x2 = x1 + 32;
  const float* cs = rope + (long)pos * 64 + c * 8;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = Elem<T>::to_f32(x1[i]); b[i] = Elem<T>::to_f32(x2[i]); }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float co = cs[i], si = cs[32 + i];
    x1[i] = Elem<T>::from_f32(a[i] * co - b[i] * si);
    x2[i] = Elem<T>::from_f32(b[i] * co + a[i] * si);
  }
}

// h[b * L + t][:] = row_bias[b][:]  (the conditioning embedders' contribution, constant along a chunk's frames)
__global__ __launch_bounds__(256) void fill_rows_kernel(float* __restrict__ h, const float* __restrict__ row_bias, int L, int d) {
  const long row = blockIdx.x;
  const float4* src = reinterpret_cast<const float4*>(row_bias + (row / L) * d);
  float4* dst = reinterpret_cast<float4*>(h + row * d);
  for (int i = threadIdx.x; i < d / 4; i += 256) dst[i] = src[i];
}
}  // namespace
}  // namespace mh

extern "C" int mh_t5_encode(const MhT5Config* c, const MhT5Weights* w, const void* mel, int B, void* enc_out,
                            float* enc_out_f32, void* workspace, int64_t workspace_bytes, void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  return mh_t5_encode_cond(c, w, mel, B, nullptr, enc_out, enc_out_f32, workspace, workspace_bytes, stream);
}

extern "C" int mh_t5_encode_cond(const MhT5Config* c, const MhT5Weights* w, const void* mel, int B, const float* row_bias,
                                 void* enc_out, float* enc_out_f32, void* workspace, int64_t workspace_bytes, void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_cfg(c, "mh_t5_encode"));
  MH_REQUIRE(w && mel && enc_out && workspace && B > 0, "mh_t5_encode: null argument");
  MH_REQUIRE(workspace_bytes >= mh_t5_encode_workspace_bytes(c, B), "mh_t5_encode: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int L = c->src_len, d = c->d_model, H = c->n_heads, inner = H * 64, dff = c->d_ff;
  const int rows = B * L, es = es_of(c->dtype), Lpad = round_up(L, 64);
  Arena ar(workspace, workspace_bytes);
  if (c->arch >= 1) {
    // ---- VarWhisperEncoder.forward (modeling_varwhisper.py:779-852) / RoPEWhisperEncoder.forward (modeling_ropewhisper.py:
    // 1167-1278); arch 2: HF WhisperEncoder.forward behind the wrapper's encoder_embedder ---------------------------------
    const bool hf = c->arch == 2;
    MH_REQUIRE(hf || !row_bias, "mh_t5_encode: arch 1 takes its conditioning as conv1 input channels (mh_cond_channels), not as a row bias");
    MH_REQUIRE(w->conv1_w && w->conv1_b && w->conv2_w && w->conv2_b, "mh_t5_encode: arch 1 / 2 need the conv front-end weights");
    MH_REQUIRE(hf || w->enc_rope, "mh_t5_encode: arch 1 needs the rotary table");
    MH_REQUIRE(!hf || (w->enc_embed_w && w->enc_pos && w->enc_final_ln_b), "mh_t5_encode: arch 2 needs encoder_embedder, embed_positions and the LayerNorm biases");
    const int fe_C = hf ? d : c->n_mels_pad;
    const int64_t fe_bytes = mh_whisper_frontend_workspace_bytes(B, c->in_frames, fe_C, d, c->dtype);
    void* fe_ws = ar.take(fe_bytes);
    void* x0 = ar.take((int64_t)rows * d * es);
    const int64_t rows_in = (int64_t)B * c->in_frames;
    float* hin = hf ? (float*)ar.take(rows_in * d * 4) : nullptr;
    void* xin = hf ? ar.take(rows_in * d * es) : nullptr;
    float* h = (float*)ar.take((int64_t)rows * d * 4);
    void* n = ar.take((int64_t)rows * d * es);
    void* qk = ar.take((int64_t)rows * 2 * inner * es);
    void* vt = ar.take((int64_t)B * inner * Lpad * es);
    void* attn = ar.take((int64_t)rows * inner * es);
    void* ff = ar.take((int64_t)rows * dff * es);
    MH_REQUIRE(ar.ok() && ff, "mh_t5_encode: arena overflow");
    const void* fe_in = mel;
    if (hf) {
      // input_features = encoder_embedder([mel | cond]) (modeling_mapperatorinator.py:204-205,211): the T5 path's projection
      // (row_bias = the conditioning columns' contribution incl. the bias), rounded to the storage type = conv1's operand
      MH_REQUIRE(xin != nullptr, "mh_t5_encode: arena overflow");
      MhGemm ge = MhGemm{};
      ge.A = mel; ge.lda = c->n_mels_pad; ge.W = w->enc_embed_w; ge.ldw = c->n_mels_pad; ge.C = hin; ge.ldc = d;
      ge.M = (int)rows_in; ge.N = d; ge.K = c->n_mels_pad; ge.bias = w->enc_embed_b; ge.dtype = c->dtype; ge.epilogue = MH_EPI_STORE_F32;
      if (row_bias) {
        hipLaunchKernelGGL(mh::fill_rows_kernel, dim3((unsigned)rows_in), dim3(256), 0, s, hin, row_bias, c->in_frames, d);
        MH_TRY(check_launch("fill_rows_kernel"));
        ge.bias = nullptr; ge.epilogue = MH_EPI_RESID;
      }
      MH_TRY(gemm(ge, s));
      const long n_in = (long)rows_in * d;
      if (c->dtype == MH_BF16) hipLaunchKernelGGL(mh::f32_to_rows_kernel<bf16_t>, dim3((unsigned)((n_in / 4 + 255) / 256 + 1)), dim3(256), 0, s, hin, (bf16_t*)xin, n_in);
      else hipLaunchKernelGGL(mh::f32_to_rows_kernel<float>, dim3((unsigned)((n_in / 4 + 255) / 256 + 1)), dim3(256), 0, s, hin, (float*)xin, n_in);
      MH_TRY(check_launch("f32_to_rows_kernel"));
      fe_in = xin;
    }
    MH_TRY(mh_whisper_frontend(fe_in, B, c->in_frames, fe_C, w->conv1_w, w->conv1_b, w->conv2_w, w->conv2_b, hf ? w->enc_pos : nullptr, d, x0,
                               fe_ws, fe_bytes, c->dtype, stream));
    const long nel = (long)rows * d;
    if (c->dtype == MH_BF16) hipLaunchKernelGGL(mh::rows_to_f32_kernel<bf16_t>, dim3((unsigned)((nel / 4 + 255) / 256 + 1)), dim3(256), 0, s, (const bf16_t*)x0, h, nel);
    else hipLaunchKernelGGL(mh::rows_to_f32_kernel<float>, dim3((unsigned)((nel / 4 + 255) / 256 + 1)), dim3(256), 0, s, (const float*)x0, h, nel);
    MH_TRY(check_launch("rows_to_f32_kernel"));
    if (hipMemsetAsync(vt, 0, (size_t)B * inner * Lpad * es, s) != hipSuccess) return check_launch("memset vt");
    MhGemm g;
    for (int l = 0; l < c->n_enc_layers; ++l) {
      const bool local = is_local_layer(c, l);
      MH_TRY(pre_norm(c, h, w->enc_ln1[l], w->enc_ln1_b[l], n, rows, c->dtype, s));
      g = MhGemm{};
      g.A = n; g.lda = d; g.W = w->enc_qkv[l]; g.ldw = d; g.C = qk; g.ldc = 2 * inner; g.M = rows; g.N = 3 * inner;
      g.K = d; g.bias = w->enc_qkv_b[l]; g.dtype = c->dtype; g.epilogue = MH_EPI_QKV_VT; g.C2 = vt; g.n_split = 2 * inner;
      g.kv_B = B; g.kv_H = H; g.kv_L = L; g.kv_Lpad = Lpad;
      MH_TRY(gemm(g, s));
      if (!hf) {   // (HF Whisper: absolute positions were added by the front-end, no rotation)
        const float* rope = (local && w->enc_rope_local) ? w->enc_rope_local : w->enc_rope;
        const long work = (long)rows * 2 * H * 4;
        if (c->dtype == MH_BF16) hipLaunchKernelGGL(mh::rope_qk_kernel<bf16_t>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, (bf16_t*)qk, 2 * inner, (long)rows, L, 2 * H, rope);
        else hipLaunchKernelGGL(mh::rope_qk_kernel<float>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, (float*)qk, 2 * inner, (long)rows, L, 2 * H, rope);
        MH_TRY(check_launch("rope_qk_kernel"));
      }
      MH_TRY(attention(qk, 2 * inner, inner, vt, Lpad, nullptr, attn, inner, B, L, H, c->attn_scale, local ? -c->local_window : 0, c->dtype, s));
      g = MhGemm{};
      g.A = attn; g.lda = inner; g.W = w->enc_o[l]; g.ldw = inner; g.C = h; g.ldc = d; g.M = rows; g.N = d; g.K = inner;
      g.bias = w->enc_o_b[l]; g.dtype = c->dtype; g.epilogue = MH_EPI_RESID;
      MH_TRY(gemm(g, s));
      MH_TRY(pre_norm(c, h, w->enc_ln2[l], w->enc_ln2_b[l], n, rows, c->dtype, s));
      MH_REQUIRE(w->enc_fc1_b[l] && w->enc_fc2_b[l], "mh_t5_encode: fc1 / fc2 carry biases in the Whisper family");
      g = MhGemm{};
      g.A = n; g.lda = d; g.W = w->enc_wi[l]; g.ldw = d; g.C = ff; g.ldc = dff; g.M = rows; g.N = dff; g.K = d;
      g.bias = w->enc_fc1_b[l]; g.dtype = c->dtype; g.epilogue = MH_EPI_BIAS_GELU_ERF;
      MH_TRY(gemm(g, s));
      g = MhGemm{};
      g.A = ff; g.lda = dff; g.W = w->enc_wo[l]; g.ldw = dff; g.C = h; g.ldc = d; g.M = rows; g.N = d; g.K = dff;
      g.bias = w->enc_fc2_b[l]; g.dtype = c->dtype; g.epilogue = MH_EPI_RESID;
      MH_TRY(gemm(g, s));
    }
    MH_TRY(pre_norm(c, h, w->enc_final_ln, w->enc_final_ln_b, enc_out, rows, c->dtype, s));
    if (enc_out_f32) MH_TRY(pre_norm(c, h, w->enc_final_ln, w->enc_final_ln_b, enc_out_f32, rows, MH_F32, s));
    return MH_OK;
  }
  float* h = (float*)ar.take((int64_t)rows * d * 4);
  void* n = ar.take((int64_t)rows * d * es);
  void* qk = ar.take((int64_t)rows * 2 * inner * es);
  void* vt = ar.take((int64_t)B * inner * Lpad * es);
  void* attn = ar.take((int64_t)rows * inner * es);
  void* ff = ar.take((int64_t)rows * dff * es);
  const bool mx = enc_mx(c);
  const int kmax = dff > d ? dff : d;
  uint8_t* xq = mx ? (uint8_t*)ar.take((int64_t)rows * kmax) : nullptr;                       // e4m3 bytes of the current A operand
  uint8_t* xs = mx ? (uint8_t*)ar.take((int64_t)rows * mx8_scale_row_bytes(kmax)) : nullptr;  // its E8M0 scales
  MH_REQUIRE(ar.ok() && ff && (!mx || xs), "mh_t5_encode: arena overflow");
  if (mx) {
    MH_REQUIRE(w->dec_ckv_all_mx && w->dec_ckv_all_mxs, "mh_t5_encode: enc_operand_dtype = MH_MX8 needs the MX-fp8 weight copies");
    for (int l = 0; l < c->n_enc_layers; ++l)
      MH_REQUIRE(w->enc_qkv_mx[l] && w->enc_qkv_mxs[l] && w->enc_o_mx[l] && w->enc_o_mxs[l] && w->enc_wi_mx[l] && w->enc_wi_mxs[l] &&
                     w->enc_wo_mx[l] && w->enc_wo_mxs[l], "mh_t5_encode: enc_operand_dtype = MH_MX8 needs the MX-fp8 weight copies (layer %d)", l);
  }
  if (hipMemsetAsync(vt, 0, (size_t)B * inner * Lpad * es, s) != hipSuccess) return check_launch("memset vt");

  MhGemm g;
  // h = encoder_embedder(mel)      (modeling_mapperatorinator.py:195-196)
  g = MhGemm{};
  g.A = mel; g.lda = c->n_mels_pad; g.W = w->enc_embed_w; g.ldw = c->n_mels_pad; g.C = h; g.ldc = d;
  g.M = rows; g.N = d; g.K = c->n_mels_pad; g.bias = w->enc_embed_b; g.dtype = c->dtype; g.epilogue = MH_EPI_STORE_F32;
  if (row_bias) {
    // conditioning: [mel | cond] @ W^T + b = mel @ W[:, :n_mels]^T + (cond @ W[:, n_mels:]^T + b); the bracket is the
    // caller's per-chunk row_bias [B, d] (it already contains b), laid under the mel product
    MH_REQUIRE(d % 4 == 0, "mh_t5_encode_cond: d_model must be a multiple of 4");
    hipLaunchKernelGGL(mh::fill_rows_kernel, dim3(rows), dim3(256), 0, s, h, row_bias, L, d);
    MH_TRY(check_launch("fill_rows_kernel"));
    g.bias = nullptr; g.epilogue = MH_EPI_RESID;
  }
  MH_TRY(gemm(g, s));

  // MX-fp8 operands (enc_operand_dtype = MH_MX8): the A operand of each projection is the e4m3 + E8M0 copy in xq / xs -- written
  // by the RMSNorm itself (the bf16 rounding of the storage mode applied first, so the quantiser sees what the bf16 path multiplies),
  // or by a pass over the bf16 buffer the attention / the gated GELU wrote
  auto operand = [&](MhGemm& gg, const void* a_plain, int K, const void* w_plain, const uint8_t* w_mx, const uint8_t* w_mxs) {
    if (mx) { gg.A = xq; gg.lda = K; gg.a_scale = xs; gg.W = w_mx; gg.ldw = K; gg.w_scale = w_mxs; gg.dtype = MH_MX8; }
    else { gg.A = a_plain; gg.lda = K; gg.W = w_plain; gg.ldw = K; gg.dtype = c->dtype; }
  };
  for (int l = 0; l < c->n_enc_layers; ++l) {
    if (mx) MH_TRY(rmsnorm_mx8(h, d, w->enc_ln1[l], rows, d, c->eps, MH_BF16, xq, d, xs, s));
    else MH_TRY(rmsnorm(h, d, w->enc_ln1[l], n, d, rows, d, c->eps, c->dtype, s));
    g = MhGemm{};
    operand(g, n, d, w->enc_qkv[l], w->enc_qkv_mx[l], w->enc_qkv_mxs[l]);
    g.C = qk; g.ldc = 2 * inner; g.M = rows; g.N = 3 * inner;
    g.K = d; g.epilogue = MH_EPI_QKV_VT; g.C2 = vt; g.n_split = 2 * inner; g.kv_B = B; g.kv_H = H;
    g.kv_L = L; g.kv_Lpad = Lpad;
    MH_TRY(gemm(g, s));
    MH_TRY(attention(qk, 2 * inner, inner, vt, Lpad, w->enc_rel_bias, attn, inner, B, L, H, 1.0f, 0, c->dtype, s));
    if (mx) MH_TRY(quantize_mx8(attn, inner, rows, inner, MH_BF16, xq, inner, xs, s));
    g = MhGemm{};
    operand(g, attn, inner, w->enc_o[l], w->enc_o_mx[l], w->enc_o_mxs[l]);
    g.C = h; g.ldc = d; g.M = rows; g.N = d; g.K = inner; g.epilogue = MH_EPI_RESID;
    MH_TRY(gemm(g, s));
    if (mx) MH_TRY(rmsnorm_mx8(h, d, w->enc_ln2[l], rows, d, c->eps, MH_BF16, xq, d, xs, s));
    else MH_TRY(rmsnorm(h, d, w->enc_ln2[l], n, d, rows, d, c->eps, c->dtype, s));
    g = MhGemm{};
    operand(g, n, d, w->enc_wi[l], w->enc_wi_mx[l], w->enc_wi_mxs[l]);
    g.C = ff; g.ldc = dff; g.M = rows; g.N = 2 * dff; g.K = d; g.epilogue = MH_EPI_GEGLU;
    // MX mode: the gated GELU leaves the GEMM as the MX-fp8 operand of wo (the bytes mh_quantize_mx8 would make of the bf16
    // hidden) -- into the second operand buffer: xq / xs still hold this GEMM's own A operand
    const bool mx_fused = mx && dff % 128 == 0;
    uint8_t* xq2 = reinterpret_cast<uint8_t*>(ff);                 // (the bf16 hidden is not written then: its buffer holds the MX image)
    uint8_t* xs2 = xq2 + (int64_t)rows * dff;
    if (mx_fused) {
      if (dff % 512) MH_REQUIRE(hipMemsetAsync(xs2, 0, (size_t)rows * mx8_scale_row_bytes(dff), s) == hipSuccess, "mh_t5_encode: scale reset failed");
      g.mx_out = xq2; g.mx_out_scales = xs2; g.ldc = dff;
    }
    MH_TRY(gemm(g, s));
    if (mx && !mx_fused) MH_TRY(quantize_mx8(ff, dff, rows, dff, MH_BF16, xq, dff, xs, s));
    g = MhGemm{};
    operand(g, ff, dff, w->enc_wo[l], w->enc_wo_mx[l], w->enc_wo_mxs[l]);
    if (mx_fused) { g.A = xq2; g.a_scale = xs2; }
    g.C = h; g.ldc = d; g.M = rows; g.N = d; g.K = dff; g.epilogue = MH_EPI_RESID;
    MH_TRY(gemm(g, s));
  }
  MH_TRY(rmsnorm(h, d, w->enc_final_ln, enc_out, d, rows, d, c->eps, c->dtype, s));
  if (enc_out_f32) MH_TRY(rmsnorm(h, d, w->enc_final_ln, enc_out_f32, d, rows, d, c->eps, MH_F32, s));
  return MH_OK;
}

extern "C" int64_t mh_t5_cross_kv_workspace_bytes(const MhT5Config* c, int B) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  if (!c || B <= 0) return -1;
  return c->enc_operand_dtype == MH_MX8 ? mx_operand_bytes((int64_t)B * c->src_len, c->d_model) : 0;
}

extern "C" int mh_t5_cross_kv_ws(const MhT5Config* c, const MhT5Weights* w, const void* enc_out, int B, void* cross_kv, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_cfg(c, "mh_t5_cross_kv"));
  MH_REQUIRE(w && enc_out && cross_kv && B > 0, "mh_t5_cross_kv: null argument");
  if (!enc_mx(c)) return mh_t5_cross_kv(c, w, enc_out, B, cross_kv, stream);
  MH_REQUIRE(workspace && workspace_bytes >= mh_t5_cross_kv_workspace_bytes(c, B), "mh_t5_cross_kv_ws: workspace too small");
  MH_REQUIRE(w->dec_ckv_all_mx && w->dec_ckv_all_mxs, "mh_t5_cross_kv_ws: enc_operand_dtype = MH_MX8 needs dec_ckv_all_mx / _mxs");
  hipStream_t s = (hipStream_t)stream;
  const int inner = c->n_heads * 64, d = c->d_model, rows = B * c->src_len;
  Arena ar(workspace, workspace_bytes);
  uint8_t* xq = (uint8_t*)ar.take((int64_t)rows * d);
  uint8_t* xs = (uint8_t*)ar.take((int64_t)rows * mx8_scale_row_bytes(d));
  MH_REQUIRE(ar.ok() && xs, "mh_t5_cross_kv_ws: arena overflow");
  MH_TRY(quantize_mx8(enc_out, d, rows, d, MH_BF16, xq, d, xs, s));
  MhGemm g = MhGemm{};
  g.A = xq; g.lda = d; g.a_scale = xs; g.W = w->dec_ckv_all_mx; g.ldw = d; g.w_scale = w->dec_ckv_all_mxs; g.C = cross_kv; g.ldc = 0;
  g.M = rows; g.N = c->n_dec_layers * 2 * inner; g.K = d; g.dtype = MH_MX8;
  g.epilogue = MH_EPI_KV_SCATTER; g.kv_B = B; g.kv_H = c->n_heads; g.kv_L = c->src_len;
  return gemm(g, s);
}

extern "C" int mh_t5_cross_kv(const MhT5Config* c, const MhT5Weights* w, const void* enc_out, int B, void* cross_kv,
                              void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_cfg(c, "mh_t5_cross_kv"));
  MH_REQUIRE(w && enc_out && cross_kv && B > 0, "mh_t5_cross_kv: null argument");
  MH_REQUIRE(!enc_mx(c), "mh_t5_cross_kv: enc_operand_dtype = MH_MX8 needs scratch for the quantised encoder output: call mh_t5_cross_kv_ws");
  const int inner = c->n_heads * 64;
  MhGemm g = MhGemm{};
  g.A = enc_out; g.lda = c->d_model; g.W = w->dec_ckv_all; g.ldw = c->d_model; g.C = cross_kv; g.ldc = 0;
  g.M = B * c->src_len; g.N = c->n_dec_layers * 2 * inner; g.K = c->d_model; g.dtype = c->dtype;
  g.bias = c->arch >= 1 ? w->dec_ckv_b_all : nullptr;
  g.epilogue = MH_EPI_KV_SCATTER; g.kv_B = B; g.kv_H = c->n_heads; g.kv_L = c->src_len;
  return gemm(g, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
namespace mh {
namespace {

struct DecState {       // device-resident control block
  int pos;              // index of the token being fed this step
  int n_running;        // rows not yet finished (written by the sampler)
  int ticket;           // arrival counter of the sampler's workgroups (the last one advances `pos`)
  unsigned rng_row0;    // MhSampling.rng_row0 and .seed of the running call: written by dec_init_kernel and read from here by
  unsigned long long seed;   // the sampler, so that the replayed step graph does not carry them (they change call by call)
};

struct SampleP {
  const float* logits; int ldl; int V;
  int32_t* tokens; int max_length;    // [B, max_length]
  const int32_t* forced;
  const uint8_t* eos_table;
  uint8_t* finished;                  // [B]
  int32_t* finish_col;                // [B]
  int32_t* last_ts_val;               // [B] value of the last TIME_SHIFT after the last SOS, -1 if none
  float* logits_dump;                 // [max_length][R][V] or null   (R = returned rows: B, or B/2 with CFG)
  float* proc;                        // [R][V] processed scores of this step (read back by the sampling passes)
  float* hist_scores;                 // [2][R][V] scores entering LookbackBias at this / the previous step
  const void* dec_embed; float* h; int d;
  // arch 2 (HF Whisper): decoder.embed_positions fp32 [tgt_len][d] added to the token embedding (NULL otherwise); pos_off [B] =
  // masked prompt columns of each row when MhT5Config.dec_pos_from_mask (written by dec_init_kernel), NULL = cache positions
  const float* dec_pos; int32_t* pos_off; const uint8_t* prompt_mask;
  MhSampling sp;
  DecState* st;
  int B, P;                           // B = rows of the WHOLE batch
  int b0;                             // first global row of this chain; logits / h / ss are chain-local
  int pair;                           // CFG: distance between the negative-prompt row g and its prompt row g + pair
                                      //      (= B/2, single chain); 0 = no guidance
  int chain_rows;                     // rows of this chain (both halves under CFG)
};

__device__ inline int32_t next_ts_state(const MhSampling& sp, int tok, int32_t cur) {
  // incremental form of MonotonicTimeShiftLogitsProcessor's "last TIME_SHIFT after the last SOS-type
  // token" scan (osuT5/osuT5/inference/logit_processors.py:150-172): the state after `tok`
  if (tok >= sp.ts_start && tok < sp.ts_end) return tok - sp.ts_start;
  for (int i = 0; i < sp.n_sos; ++i)
    if (tok == sp.sos_ids[i]) return -1;
  return cur;
}

// row of decoder.embed_positions for column `col` of global row b (arch 2)
__device__ inline int dec_pos_row(const SampleP& p, int b, int col) {
  const int off = p.pos_off ? p.pos_off[b] : 0;
  return col - off > 0 ? col - off : 0;
}

// input_ids[row][i] as the reference's processors see it: the prompt, then the ids that were FED (the forced ids
// under teacher forcing, the emitted ones otherwise)
__device__ inline int history_id(const SampleP& p, int row, int i) {
  return (p.forced && i >= p.P) ? p.forced[(long)row * p.max_length + i] : p.tokens[(long)row * p.max_length + i];
}

// counter-based RNG (Philox-like mixing is overkill here; splitmix64 on (seed, row, step))
__device__ inline float uniform01(uint64_t seed, uint32_t row, uint32_t step) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)row * 0x100000001ull + step + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)((z >> 40) + 0.5) * (1.0f / 16777216.0f);
}

// init: consume prompt columns 0..start_pos (processor state machine) and embed the token at start_pos, the first
// position the per-token loop feeds (0 without prefill, P-1 after the batched prompt prefill)
template <typename T>
__global__ __launch_bounds__(256) void dec_init_kernel(SampleP p, int chain_rows, int start_pos) {
  const int lb = blockIdx.x, b = p.b0 + lb;
  if (threadIdx.x == 0) {
    int32_t v = -1;
    for (int i = 0; i <= start_pos; ++i) v = next_ts_state(p.sp, p.tokens[(long)b * p.max_length + i], v);
    p.last_ts_val[b] = v;
    p.finished[b] = 0;
    p.finish_col[b] = p.max_length - 1;
    if (lb == 0) { p.st->pos = start_pos; p.st->n_running = chain_rows; p.st->ticket = 0; p.st->rng_row0 = p.sp.rng_row0; p.st->seed = p.sp.seed; }
  }
  __shared__ int s_off;
  if (threadIdx.x == 0) {
    int off = 0;
    if (p.pos_off) {   // transformers 4.57's Whisper decoder_position_ids = cumsum(mask) - 1, clamped at 0: a left-padded row counts from its first real token
      if (p.prompt_mask)
        for (int i = 0; i < p.P; ++i) off += p.prompt_mask[(long)b * p.P + i] == 0;
      p.pos_off[b] = off;
    }
    s_off = off;
  }
  __syncthreads();
  const int tok = p.tokens[(long)b * p.max_length + start_pos];
  const T* e = reinterpret_cast<const T*>(p.dec_embed) + (long)tok * p.d;
  const float* pe = p.dec_pos ? p.dec_pos + (long)(start_pos - s_off > 0 ? start_pos - s_off : 0) * p.d : nullptr;
  for (int i = threadIdx.x; i < p.d; i += 256) p.h[(long)lb * p.d + i] = Elem<T>::to_f32(e[i]) + (pe ? pe[i] : 0.f);
}

constexpr int kSampleRegs = 16;   // the register sampling path holds up to 16 ids per thread: vocabularies of <= 4096 ids

// One workgroup per returned row (per CFG pair): processors -> selection -> bookkeeping -> next-token embedding.
// Processor order = server.py:106-134: CFG -> MonotonicTimeShift -> TimeshiftBias -> (Conditional)Temperature ->
// LookbackBias, then HF's own top-k / top-p warpers and the multinomial draw.
template <typename T>
__global__ __launch_bounds__(256) void dec_sample_kernel(SampleP p) {
  __shared__ float sf[8];
  __shared__ int si[8];
  __shared__ float s3[12];
  __shared__ float s_e[256 * kSampleRegs];
  __shared__ int s_is_last;
  __shared__ float s_sum, s_temp;
  __shared__ int s_timed;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool cfg = p.pair > 0;
  const int nrow = cfg ? 2 : 1;
  const int lneg = blockIdx.x;                 // chain-local row of the negative prompt (CFG only)
  const int lb = cfg ? lneg + p.pair : lneg;   // chain-local returned row (the row whose ids are `input_ids`)
  const int b = p.b0 + lb;                     // its global row
  const int bneg = p.b0 + lneg;
  const int gr = cfg ? lneg : b;               // index among the returned rows
  const int R = cfg ? p.pair : p.B;
  // Requested before anything else, all in ONE round trip: this row's logits (vocabularies of <= 4096 ids sit in registers,
  // thread t holds ids t, t + 256, ...), its finished flag and time-shift state -- none of them depends on the position.
  // (The loop form `for (v = tid; v < V; v += 256) lg[v]` is a chain of V / 256 dependent round trips: the trip count is
  // dynamic, so hipcc neither unrolls it nor hoists the loads.)
  const float* lg = p.logits + (long)lb * p.ldl;
  const float* lgn = p.logits + (long)lneg * p.ldl;
  const bool in_regs = p.V <= 256 * kSampleRegs;
  float raw[kSampleRegs], rawn[kSampleRegs];
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < kSampleRegs; ++i) {
      int v = tid + 256 * i;
      v = v < p.V ? v : p.V - 1;
      raw[i] = lg[v];
      rawn[i] = cfg ? lgn[v] : 0.f;
    }
  }
  const bool was_finished = p.finished[b] != 0;
  const int ltv = p.last_ts_val[b];
  const int pos = p.st->pos;
  const int col = pos + 1;  // column being produced
  const MhSampling& sp = p.sp;
  if (col >= p.max_length) return;
  int nxt[2] = {0, 0};      // the ids the next step is fed (row b, and under guidance its negative-prompt row)

  if (col < p.P) {
    // still inside the prompt: the token is given; only advance the processor state + embedding
    nxt[0] = p.tokens[(long)b * p.max_length + col];
    if (cfg) nxt[1] = p.tokens[(long)bneg * p.max_length + col];
    if (tid == 0) dec::store_wt(&p.last_ts_val[b], next_ts_state(sp, nxt[0], ltv));
  } else {
    if (tid == 0) {
      // ConditionalTemperatureLogitsWarper: the lookback is ROW 0's history for the whole batch
      // (logit_processors.py:75-80 `input_ids[0, -max_offset:]`), first matching rule wins; cond_per_row: the row's
      // own history (= the reference called with batch 1 per row: sequential songs, shards)
      float temp = sp.temperature;
      const int row0 = sp.cond_per_row ? b : (cfg ? p.pair : 0);
      for (int j = 0; j < sp.n_cond; ++j) {
        const int off = sp.cond_offset[j];
        if (col >= off && (sp.tok_flags[history_id(p, row0, col - off)] & (2 << j))) { temp = sp.cond_temp[j]; break; }
      }
      s_temp = temp;
      // LookbackBiasLogitsWarper types_first: "the scores are for a timeshift event" when the last id is a timed event
      s_timed = (sp.lookback_types_first && col > p.P) ? (sp.tok_flags[history_id(p, b, col - 1)] & 1) : 0;
    }
    __syncthreads();
    const float temp = s_temp;
    const bool lb_range_on = sp.lookback_mask_end > sp.ts_start;
    // CFG -> monotonic -> bias -> temperature
    auto warped_of = [&](int v, float x, float xn) -> float {
      // HF ClassifierFreeGuidanceLogitsProcessor on the reference's row order (first half = negative prompt):
      // uncond + (cond - uncond) * scale with cond = first half, no fused multiply-add
      if (cfg) x = __fadd_rn(x, __fmul_rn(__fsub_rn(xn, x), sp.cfg_scale));
      // MonotonicTimeShiftLogitsProcessor: ids [ts_start, ts_start + value) -> -inf
      if (ltv >= 0 && v >= sp.ts_start && v < sp.ts_start + ltv) x = -INFINITY;
      // TimeshiftBias
      if (sp.timeshift_bias != 0.f && v >= sp.ts_start && v < sp.ts_end) x += sp.timeshift_bias;
      // TemperatureLogitsWarper / ConditionalTemperatureLogitsWarper (scores / temperature)
      return x / temp;
    };
    auto warped = [&](int v) -> float { return warped_of(v, lg[v], cfg ? lgn[v] : 0.f); };
    float* dump = p.logits_dump ? p.logits_dump + ((long)col * R + gr) * p.V : nullptr;
    float* fin = p.proc + (long)gr * p.V;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    const bool keep_scores = sp.do_sample != 0;   // only the sampling passes below read the processed scores back
    auto consider = [&](int v, float x) {
      if (keep_scores) fin[v] = x;
      if (dump) dump[v] = x;
      if (x > best) { best = x; besti = v; }   // strided scan keeps the smallest index per thread
    };
    if (sp.lookback_types_first && lb_range_on) {
      // LookbackBiasLogitsWarper, types_first == True (logit_processors.py:116-133).  `last_scores` = what entered
      // this processor at the previous step: two row buffers indexed by the parity of the column.
      float* cur = p.hist_scores + ((long)(col & 1) * R + gr) * p.V;
      const float* last = p.hist_scores + ((long)((col & 1) ^ 1) * R + gr) * p.V;
      const bool renorm = s_timed != 0;
      float mc = -INFINITY, ml = -INFINITY;
      for (int v = tid; v < p.V; v += 256) {
        const float x = warped(v);
        cur[v] = x;
        mc = fmaxf(mc, x);
        if (renorm) ml = fmaxf(ml, last[v]);
      }
      if (!renorm) {
        for (int v = tid; v < p.V; v += 256) consider(v, cur[v]);
      } else {
        mc = block_max(mc, sf);
        ml = block_max(ml, sf);
        float s_cur = 0.f, o_cur = 0.f, s_last = 0.f, e_last = 0.f;
        for (int v = tid; v < p.V; v += 256) {
          const float ec = expf(cur[v] - mc), el = expf(last[v] - ml);
          s_cur += ec;
          if (!(v >= sp.ts_start && v < sp.lookback_mask_end)) o_cur += ec;
          s_last += el;
          if (sp.tok_flags[v] & 16) e_last += el;
        }
        s_cur = block_sum(s_cur, sf);
        o_cur = block_sum(o_cur, sf);
        s_last = block_sum(s_last, sf);
        e_last = block_sum(e_last, sf);
        const float prob_eos = e_last / s_last, prob_event = 1.f - prob_eos;
        const float sc = 1.f / ((o_cur / s_cur) * prob_event + prob_eos);
        const float extra = fminf(fmaxf((sc - 1.f) * prob_eos / prob_event, 0.f), 1.f);
        const float log_norm = logf(s_cur) - logf(sc);
        for (int v = tid; v < p.V; v += 256) {
          float x;
          if (v >= sp.ts_start && v < sp.lookback_mask_end) x = (v == sp.ts_start) ? logf(extra) : -INFINITY;
          else x = (cur[v] - mc) - log_norm;   // log(softmax(cur)[v] * sc), kept in the log domain (no underflow)
          consider(v, x);
        }
      }
    } else if (in_regs) {
#pragma unroll
      for (int i = 0; i < kSampleRegs; ++i) {
        const int v = tid + 256 * i;
        if (v < p.V) {
          float x = warped_of(v, raw[i], rawn[i]);
          // LookbackBiasLogitsWarper, types_first == False branch
          if (lb_range_on && v >= sp.ts_start && v < sp.lookback_mask_end) x = -INFINITY;
          consider(v, x);
        }
      }
    } else {
      for (int v = tid; v < p.V; v += 256) {
        float x = warped(v);
        if (lb_range_on && v >= sp.ts_start && v < sp.lookback_mask_end) x = -INFINITY;
        consider(v, x);
      }
    }
    // argmax with first-index tie-break (torch.argmax semantics on ties = first maximal index)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(besti, o, 64);
      if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __syncthreads();   // the reductions above may still be reading sf
    if (lane == 0) { sf[wid] = best; si[wid] = besti; }
    __syncthreads();
    if (tid == 0) {
      for (int w2 = 1; w2 < 4; ++w2)
        if (sf[w2] > best || (sf[w2] == best && si[w2] < besti)) { best = sf[w2]; besti = si[w2]; }
      sf[4] = best; si[4] = besti;
    }
    __syncthreads();
    int tok = si[4];
    if (sp.do_sample && p.V <= 256 * kSampleRegs) {
      // softmax sampling with optional top-k / top-p truncation, the processed scores of this row in REGISTERS (thread t
      // holds ids t, t + 256, ...): both thresholds by 4-ary search (three candidate thresholds per block reduction),
      // the draw by a block-wide prefix sum in id order.  The reference's default user settings sample (top_p 0.9,
      // configs/inference/v32.yaml:12-13): the memory-bound bisection + serial scan below cost 350 us per token step.
      const float mx = sf[4];
      __syncthreads();   // sf / s3 are reduction scratch below
      float xr[kSampleRegs];
#pragma unroll
      for (int i = 0; i < kSampleRegs; ++i) {
        const int v = tid + 256 * i;
        xr[i] = v < p.V ? fin[v] - mx : -INFINITY;     // this thread's own stores (consider)
      }
      auto reduce3 = [&](float a, float b, float c, float (&out)[3]) {
        a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
        __syncthreads();
        if (lane == 0) { s3[wid * 3] = a; s3[wid * 3 + 1] = b; s3[wid * 3 + 2] = c; }
        __syncthreads();
        out[0] = s3[0] + s3[3] + s3[6] + s3[9];
        out[1] = s3[1] + s3[4] + s3[7] + s3[10];
        out[2] = s3[2] + s3[5] + s3[8] + s3[11];
      };
      float thr = -INFINITY;
      if (sp.top_k > 0 && sp.top_k < p.V) {   // largest threshold (to 80 / 4^14) that still keeps >= top_k ids
        float lo = -80.f, hi = 0.f;
        for (int it = 0; it < 14; ++it) {
          const float q = 0.25f * (hi - lo), t1 = lo + q, t2 = lo + 2.f * q, t3 = lo + 3.f * q;
          float c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
          for (int i = 0; i < kSampleRegs; ++i) { c1 += xr[i] >= t1 ? 1.f : 0.f; c2 += xr[i] >= t2 ? 1.f : 0.f; c3 += xr[i] >= t3 ? 1.f : 0.f; }
          float r[3];
          reduce3(c1, c2, c3, r);
          const float k = (float)sp.top_k;
          if (r[2] >= k) lo = t3; else if (r[1] >= k) { lo = t2; hi = t3; } else if (r[0] >= k) { lo = t1; hi = t2; } else hi = t1;
        }
        thr = lo;
      }
      float er[kSampleRegs];
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < kSampleRegs; ++i) { er[i] = xr[i] >= thr ? __expf(xr[i]) : 0.f; part += er[i]; }
      const float total = block_sum(part, sf);
      if (sp.top_p < 1.0f) {   // largest probability threshold whose kept mass still reaches top_p
        float lo = 0.f, hi = 1.f;
        const float inv = 1.0f / total;
        for (int it = 0; it < 12; ++it) {
          const float q = 0.25f * (hi - lo), t1 = lo + q, t2 = lo + 2.f * q, t3 = lo + 3.f * q;
          float m1 = 0.f, m2 = 0.f, m3 = 0.f;
#pragma unroll
          for (int i = 0; i < kSampleRegs; ++i) {
            const float pr = er[i] * inv;
            m1 += pr >= t1 ? pr : 0.f; m2 += pr >= t2 ? pr : 0.f; m3 += pr >= t3 ? pr : 0.f;
          }
          float r[3];
          reduce3(m1, m2, m3, r);
          if (r[2] >= sp.top_p) lo = t3; else if (r[1] >= sp.top_p) { lo = t2; hi = t3; } else if (r[0] >= sp.top_p) { lo = t1; hi = t2; } else hi = t1;
        }
#pragma unroll
        for (int i = 0; i < kSampleRegs; ++i) er[i] = (er[i] * inv >= lo) ? er[i] : 0.f;
      }
      // kept weights in id order -> contiguous ownership (thread t: ids [t C, t C + C)) -> inclusive prefix -> first id
      // whose running sum reaches u (else the last kept id), as a sequential scan in id order would pick
#pragma unroll
      for (int i = 0; i < kSampleRegs; ++i) {
        const int v = tid + 256 * i;
        if (v < p.V) s_e[v] = er[i];
      }
      if (tid == 0) { si[5] = 0x7fffffff; si[6] = -1; }
      __syncthreads();
      const int Cn = (p.V + 255) / 256;
      float loc[kSampleRegs];
      float run = 0.f;
#pragma unroll
      for (int i = 0; i < kSampleRegs; ++i) {
        const int v = tid * Cn + i;
        const float e = (i < Cn && v < p.V) ? s_e[v] : 0.f;
        run += e;
        loc[i] = run;
      }
      float incl = run;                       // inclusive scan of the thread totals over the block
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      if (lane == 63) sf[wid] = incl;
      __syncthreads();
      float base = incl - run;
      for (int w2 = 0; w2 < wid; ++w2) base += sf[w2];
      const float mass = sf[0] + sf[1] + sf[2] + sf[3];
      const float u = uniform01(p.st->seed, (uint32_t)gr + p.st->rng_row0, (uint32_t)col) * mass;
      int first = 0x7fffffff, last = -1;
      float prev = 0.f;
#pragma unroll
      for (int i = 0; i < kSampleRegs; ++i) {
        const int v = tid * Cn + i;
        const bool kept = i < Cn && v < p.V && loc[i] > prev;     // a kept id carries positive weight
        if (kept) { last = v; if (base + loc[i] >= u && first == 0x7fffffff) first = v; }
        prev = loc[i];
      }
      if (first != 0x7fffffff) atomicMin(&si[5], first);
      if (last >= 0) atomicMax(&si[6], last);
      __syncthreads();
      tok = si[5] != 0x7fffffff ? si[5] : (si[6] >= 0 ? si[6] : tok);
    } else if (sp.do_sample) {
      // (vocabularies beyond 256 * kSampleRegs ids: the same search through memory)
      const float mx = sf[4];
      __syncthreads();   // sf[4] is reused as reduction scratch below
      float thr = -INFINITY;
      if (sp.top_k > 0 && sp.top_k < p.V) {
        float lo = -80.f, hi = 0.f;  // on x - mx
        for (int it = 0; it < 40; ++it) {
          const float mid = 0.5f * (lo + hi);
          int cnt = 0;
          for (int v = tid; v < p.V; v += 256) cnt += (fin[v] - mx >= mid) ? 1 : 0;
          cnt = (int)block_sum((float)cnt, sf);
          if (cnt >= sp.top_k) lo = mid; else hi = mid;
        }
        thr = lo;
      }
      // probabilities of the kept set
      float part = 0.f;
      for (int v = tid; v < p.V; v += 256) {
        const float x = fin[v];
        if (x - mx >= thr) part += __expf(x - mx);
      }
      float total = block_sum(part, sf);
      float pthr = 0.f;
      if (sp.top_p < 1.0f) {
        // keep the smallest set of most-probable ids whose mass reaches top_p
        float lo = 0.f, hi = 1.f;
        for (int it = 0; it < 30; ++it) {
          const float mid = 0.5f * (lo + hi);
          float mass = 0.f;
          for (int v = tid; v < p.V; v += 256) {
            const float x = fin[v];
            if (x - mx >= thr) {
              const float pr = __expf(x - mx) / total;
              if (pr >= mid) mass += pr;
            }
          }
          mass = block_sum(mass, sf);
          if (mass >= sp.top_p) lo = mid; else hi = mid;
        }
        pthr = lo;
        float part2 = 0.f;
        for (int v = tid; v < p.V; v += 256) {
          const float x = fin[v];
          if (x - mx >= thr && __expf(x - mx) / total >= pthr) part2 += __expf(x - mx);
        }
        const float t2 = block_sum(part2, sf);
        if (tid == 0) s_sum = t2;
        __syncthreads();
      } else {
        if (tid == 0) s_sum = total;
        __syncthreads();
      }
      if (tid == 0) {
        const float u = uniform01(p.st->seed, (uint32_t)gr + p.st->rng_row0, (uint32_t)col) * s_sum;
        float cum = 0.f;
        int pick = tok;
        for (int v = 0; v < p.V; ++v) {
          const float x = fin[v];
          if (x - mx >= thr) {
            const float e = __expf(x - mx);
            if (sp.top_p < 1.0f && e / total < pthr) continue;
            cum += e;
            pick = v;
            if (cum >= u) break;
          }
        }
        si[5] = pick;
      }
      __syncthreads();
      tok = si[5];
    }
    // every thread knows the emitted id (the finished flag came with the first round trip), so the embedding row below is
    // requested at once -- beside thread 0's EOS lookup and bookkeeping stores, not behind them
    const bool forced = p.forced != nullptr;
    const int emit = was_finished ? sp.pad_id : tok;      // HF: finished rows receive pad_token_id
    nxt[0] = forced ? p.forced[(long)b * p.max_length + col] : emit;
    if (cfg) nxt[1] = forced ? p.forced[(long)bneg * p.max_length + col] : emit;
    if (tid == 0) {
      const bool done = !forced && !was_finished && (p.eos_table[emit] || col + 1 >= sp.max_length);
      for (int j = 0; j < nrow; ++j) {   // the rows of a CFG pair receive the same id (decoder_input_ids.repeat)
        const int row = j == 0 ? b : bneg;
        dec::store_wt(p.tokens + (long)row * p.max_length + col, (int32_t)emit);
        if (done) {
          __hip_atomic_store(&p.finished[row], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by the last-arriving workgroup below
          dec::store_wt(p.finish_col + row, (int32_t)col);
        }
      }
      dec::store_wt(&p.last_ts_val[b], next_ts_state(sp, nxt[0], ltv));
    }
  }
  // embedding of the token that the next step consumes (decoder_embedder, modeling_mapperatorinator.py:205-206)
  for (int j = 0; j < nrow; ++j) {
    const int lrow = j == 0 ? lb : lneg;
    const T* e = reinterpret_cast<const T*>(p.dec_embed) + (long)nxt[j] * p.d;
    const float* pe = p.dec_pos ? p.dec_pos + (long)dec_pos_row(p, j == 0 ? b : bneg, col) * p.d : nullptr;   // (arch 2: + embed_positions[col])
    for (int i = tid * 4; i < p.d; i += 1024) {   // 16-byte write-through pieces (d is a multiple of 4)
      float v4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v4[k] = Elem<T>::to_f32(e[i + k]) + (pe ? pe[i + k] : 0.f);
      dec::store_piece_wt<float>(p.h + (long)lrow * p.d + i, v4, 4, 4);
    }
  }
  // Step bookkeeping without a launch of its own: every workgroup read `pos` when it started, so the one that arrives
  // last may advance it; it also recounts the running rows for the host's early-stop poll.  The `finished` flags are
  // agent-scope stores drained before the ticket is taken and agent-scope loads here (write-through hand-off, no
  // fence); a stale flag could only delay the early stop by a poll, never change a token.
  // The recount is ONE round trip of the last workgroup's first wave (lane i reads row i's flag), not chain_rows serial ones.
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = __hip_atomic_fetch_add(&p.st->ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_is_last = t == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (s_is_last && wid == 0) {
    int run = 0;
    for (int i0 = 0; i0 < p.chain_rows; i0 += 64) {
      const int i = i0 + lane;
      const bool running = i < p.chain_rows &&
                           __hip_atomic_load(&p.finished[p.b0 + (i < p.chain_rows ? i : 0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
      run += __popcll(__ballot(running));
    }
    if (lane == 0) {
      __hip_atomic_store(&p.st->ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dec::store_wt(&p.st->n_running, run);
      dec::store_wt(&p.st->pos, pos + 1);
    }
  }
}

__global__ void dec_finalize_kernel(const int32_t* finish_col, int B, int32_t* n_steps_out) {
  if (threadIdx.x == 0) {
    int mx = 0;
    for (int b = 0; b < B; ++b) mx = finish_col[b] > mx ? finish_col[b] : mx;
    *n_steps_out = mx + 1;
  }
}

// real columns per 16-column tile of a decode GEMV: 16 when that still yields >= 192 workgroups, else 8 down to 96
// workgroups, else 4 (every workgroup re-reads ALL activations through L2, the resource the two decode chains fight
// over: at N = 768, 96 workgroups of 8 columns measured 42.9 k tok/s against 42.4 k for 192 of 4 and 42.1 k for 48 of
// 16).  Option decode_gemv_cols overrides; a result never depends on it: every column is an independent dot product.
int gemv_cols(int N) {
  const long o = option(OPT_DECODE_GEMV_COLS);
  if (o == 4 || o == 8 || o == 16) return (int)o;
  if (N / 16 >= 192) return 16;
  if (N / 8 >= 96) return 8;
  return 4;
}

template <typename T, int MF, int PRO, int EPI, bool BIAS = false>
int launch_skinny(dec::SkinnyP p, hipStream_t s) {
  const int kb = 4 * (16 / (int)sizeof(T));
  MH_REQUIRE(p.K % kb == 0 && p.K >= kb, "decode: K=%d must be a positive multiple of %d", p.K, kb);
  const int nkb = p.K / kb;
  MH_REQUIRE(PRO != dec::PRO_PLAIN || nkb % 2 == 0, "decode: K=%d must be a multiple of %d", p.K, 2 * kb);
  // waves per workgroup: a function of K ONLY (batch invariance of the summation order)
  const bool wide = nkb > 4 * dec::kGemvCH;
  MH_REQUIRE(PRO == dec::PRO_PLAIN || nkb <= 8 * dec::kGemvCH, "decode: the norm prologue needs K <= %d", 8 * dec::kGemvCH * kb);
  MH_REQUIRE(PRO != dec::PRO_LAYERNORM || p.ln_b, "decode: LayerNorm prologue without its bias");
  MH_REQUIRE(p.lda == p.K && p.ldw == p.K, "decode: the GEMV operands must be dense (lda = ldw = K)");
  MH_REQUIRE(EPI != dec::SK_RESID || (p.N % 4 == 0 && p.ldh == p.N), "decode: residual GEMV needs a dense [B, N] residual stream, N a multiple of 4");
  int tiles;
  if (EPI == dec::SK_GEGLU) { p.nv = 8; tiles = ceil_div(p.N / 2, 8); }
  else { p.nv = gemv_cols(p.N); tiles = ceil_div(p.N, p.nv); }
  if (wide) {
    if constexpr (PRO != dec::PRO_PLAIN && sizeof(T) == 2) {   // d_model > 1024 in bf16: not a shape of this model family
      set_error("decode: RMSNorm GEMV needs d_model <= 1024 in bf16 storage");
      return MH_ERR_ARG;
    } else {
      hipLaunchKernelGGL((dec::gemv_kernel<T, MF, 8, PRO, EPI, BIAS>), dim3(tiles), dim3(512), 0, s, MH_GEMV_LEAD_ARGS(p), p);
    }
  } else {
    hipLaunchKernelGGL((dec::gemv_kernel<T, MF, 4, PRO, EPI, BIAS>), dim3(tiles), dim3(256), 0, s, MH_GEMV_LEAD_ARGS(p), p);
  }
  return check_launch("gemv_kernel");
}

template <typename T, int PRO, int EPI, bool BIAS = false>
int skinny(const dec::SkinnyP& p, hipStream_t s) {
  MH_REQUIRE(!BIAS || p.bias, "decode: biased GEMV without a bias");
  if (p.B <= 16) return launch_skinny<T, 1, PRO, EPI, BIAS>(p, s);
  if (p.B <= 32) return launch_skinny<T, 2, PRO, EPI, BIAS>(p, s);
  return launch_skinny<T, 4, PRO, EPI, BIAS>(p, s);
}
// residual GEMV with an optional bias (the Whisper family's Wo / fc2)
template <typename T>
int skinny_resid(const dec::SkinnyP& p, hipStream_t s) {
  return p.bias ? skinny<T, dec::PRO_PLAIN, dec::SK_RESID, true>(p, s) : skinny<T, dec::PRO_PLAIN, dec::SK_RESID>(p, s);
}

constexpr int kMaxChains = 8;

struct DecBuffers {
  float* h; void* q; void* attn; void* ff; float* logits; int chain;
  void* self_k; void* self_v;  // [n_dec][B][H][tgt][64]
  uint8_t* finished; int32_t* finish_col; int32_t* last_ts; DecState* st;
};

template <typename T>
int launch_cross(const dec::CrossAttnP& ca, hipStream_t s) {
  // two keys in flight per 8-lane group (measured stand-alone, B=32 base bf16: U=2 5.08, U=4 4.64, U=8 3.37 TB/s)
  hipLaunchKernelGGL((dec::dec_cross_attn_kernel<T, 2>), dim3(ca.B * ca.H), dim3(1024), 0, s, ca);
  return check_launch("dec_cross_attn_kernel");
}

// attention kernels that project their own q (/ k / v): see decode_kernels.hpp.  d_model = 128 KC, KC in 1..8; option
// decode_fused_proj = 0 runs the stand-alone QKV and cross-Q GEMV launches instead (same results up to fp32 summation
// order of the projections; both forms are covered by the GPU tests).
bool fused_proj_enabled(int d) { return option(OPT_DECODE_FUSED_PROJ) != 0 && d % 128 == 0 && d >= 128 && d <= 1024; }

#define MH_SELF_LEAD_ARGS hp.h, hp.ln_w, hp.W, sa.pos, sa.kc, sa.vc, sa.H, hp.d
#define MH_CROSS_LEAD_ARGS hp.h, hp.ln_w, hp.W, ca.k, ca.v, ca.H, ca.L, hp.d, ca.kv_B
template <typename T, int KC>
int launch_self_qkv(const dec::SelfAttnP& sa, const dec::HeadProjP& hp, int inner, hipStream_t s) {
  MH_REQUIRE(hp.ldh == hp.d && hp.ldw == hp.d && inner == sa.H * 64, "decode: dense residual rows / projection weights expected");
  if (sa.rope && hp.ln_b)   // HF Whisper (arch 2): the same behind an affine LayerNorm, identity rotary table
    hipLaunchKernelGGL((dec::dec_self_attn_qkv_kernel<T, KC, true, true>), dim3(sa.B * sa.H), dim3(1024), 0, s, MH_SELF_LEAD_ARGS, sa, hp);
  else if (sa.rope)   // the Whisper family: biased fused Wqkv, RoPE, scaled scores, optional window
    hipLaunchKernelGGL((dec::dec_self_attn_qkv_kernel<T, KC, true>), dim3(sa.B * sa.H), dim3(1024), 0, s, MH_SELF_LEAD_ARGS, sa, hp);
  else
    hipLaunchKernelGGL((dec::dec_self_attn_qkv_kernel<T, KC>), dim3(sa.B * sa.H), dim3(1024), 0, s, MH_SELF_LEAD_ARGS, sa, hp);
  return check_launch("dec_self_attn_qkv_kernel");
}
template <typename T>
int launch_self_qkv_d(const dec::SelfAttnP& sa, const dec::HeadProjP& hp, int inner, hipStream_t s) {
  switch (hp.d) {
    case 128: return launch_self_qkv<T, 1>(sa, hp, inner, s);
    case 256: return launch_self_qkv<T, 2>(sa, hp, inner, s);
    case 384: return launch_self_qkv<T, 3>(sa, hp, inner, s);
    case 512: return launch_self_qkv<T, 4>(sa, hp, inner, s);
    case 640: return launch_self_qkv<T, 5>(sa, hp, inner, s);
    case 768: return launch_self_qkv<T, 6>(sa, hp, inner, s);
    case 896: return launch_self_qkv<T, 7>(sa, hp, inner, s);
    default: return launch_self_qkv<T, 8>(sa, hp, inner, s);
  }
}
#ifndef MH_CROSS_U
#define MH_CROSS_U 1     // keys in flight per 8-lane group of the fused cross-attention kernel (A/B builds: 2, 4)
#endif
template <typename T, int KC>
int launch_cross_q(const dec::CrossAttnP& ca, const dec::HeadProjP& hp, hipStream_t s) {
  MH_REQUIRE(hp.ldh == hp.d && hp.ldw == hp.d, "decode: dense residual rows / projection weights expected");
  // one key in flight per 8-lane group: 64 VGPRs without spills (two 16-wave workgroups per CU); U = 2 measured the
  // same bandwidth in the stand-alone kernel
  if (ca.scale != 0.f) {   // the Whisper family: biased Wq, scaled scores
    MH_REQUIRE(ca.kscale == nullptr, "decode: the fp8 cross K/V copy is not wired for the Whisper family");
    if (hp.ln_b) hipLaunchKernelGGL((dec::dec_cross_attn_q_kernel<T, KC, MH_CROSS_U, false, true, true>), dim3(ca.B * ca.H), dim3(1024), 0, s, MH_CROSS_LEAD_ARGS, ca, hp);
    else hipLaunchKernelGGL((dec::dec_cross_attn_q_kernel<T, KC, MH_CROSS_U, false, true>), dim3(ca.B * ca.H), dim3(1024), 0, s, MH_CROSS_LEAD_ARGS, ca, hp);
  } else if (ca.kscale != nullptr) {
    if constexpr (sizeof(T) == 2)
      hipLaunchKernelGGL((dec::dec_cross_attn_q_kernel<T, KC, MH_CROSS_U, true>), dim3(ca.B * ca.H), dim3(1024), 0, s, MH_CROSS_LEAD_ARGS, ca, hp);
    else { set_error("decode: the fp8 cross K/V copy needs bf16 storage"); return MH_ERR_ARG; }
  } else {
    hipLaunchKernelGGL((dec::dec_cross_attn_q_kernel<T, KC, MH_CROSS_U>), dim3(ca.B * ca.H), dim3(1024), 0, s, MH_CROSS_LEAD_ARGS, ca, hp);
  }
  return check_launch("dec_cross_attn_q_kernel");
}
template <typename T>
int launch_cross_q_d(const dec::CrossAttnP& ca, const dec::HeadProjP& hp, hipStream_t s) {
  switch (hp.d) {
    case 128: return launch_cross_q<T, 1>(ca, hp, s);
    case 256: return launch_cross_q<T, 2>(ca, hp, s);
    case 384: return launch_cross_q<T, 3>(ca, hp, s);
    case 512: return launch_cross_q<T, 4>(ca, hp, s);
    case 640: return launch_cross_q<T, 5>(ca, hp, s);
    case 768: return launch_cross_q<T, 6>(ca, hp, s);
    case 896: return launch_cross_q<T, 7>(ca, hp, s);
    default: return launch_cross_q<T, 8>(ca, hp, s);
  }
}

// measurement hook (mh_t5_decode_timing): device buffer that receives per-launch timestamps of the dominant kernel
struct DecodeTiming { unsigned long long* buf = nullptr; int ring = 0; };
DecodeTiming g_timing;

template <typename T>
int enqueue_step(const MhT5Config* c, const MhT5Weights* w, const void* cross_kv, int B, int Bfull, int kvB,
                 const uint8_t* prompt_mask, int P, const DecBuffers& bf, const SampleP& smp, hipStream_t s,
                 const void* kv8 = nullptr, const float* kv8_scales = nullptr, bool with_sampler = true, int kv_group = 0) {
  // with_sampler = false: the step ends with the logits (mh_t5_step: the host selects); kv_group > 1: rows are (chunk, beam)
  // pairs and row b reads cross K/V row b / kv_group
  // kv8 / kv8_scales: the chain's first row of the e4m3 copy of cross_kv and of its scales (mh_t5_quantize_cross_kv)
  // B rows of one chain; every pointer in `bf` / `cross_kv` / `prompt_mask` already points at the chain's first
  // row, only the per-layer strides of the caches use the full batch size.  kvB = rows of cross_kv (B/2 under CFG:
  // a pair shares its encoder output, row b reads K/V row b % kvB).
  const int d = c->d_model, H = c->n_heads, inner = H * 64, dff = c->d_ff, L = c->src_len, tgt = c->tgt_len;
  const int es = (int)sizeof(T);
  const int* posp = &bf.st->pos;
  const bool wh = c->arch >= 1, hf = c->arch == 2;   // (arch 2: affine LayerNorm prologues, the identity rotary table the host packs)
  if (wh) MH_REQUIRE(fused_proj_enabled(d) && option(OPT_DECODE_FUSED_PROJ) == 1 && w->dec_rope,
                     "decode: the Whisper family runs on the fused attention kernels (d_model a multiple of 128 <= 1024) and needs its rotary table");
  for (int l = 0; l < c->n_dec_layers; ++l) {
    const long cache_off = (long)l * Bfull * H * tgt * 64 * es;
    dec::SkinnyP sk{};
    if (wh) {
      // ---- VarWhisperDecoderLayer (modeling_varwhisper.py:633-741), one token: the same six dependent kernels -------
      const bool local = is_local_layer(c, l);
      dec::SelfAttnP sa{};
      sa.kc = (char*)bf.self_k + cache_off; sa.vc = (char*)bf.self_v + cache_off; sa.prompt_mask = prompt_mask; sa.P = P;
      sa.out = bf.attn; sa.ldo = inner; sa.B = B; sa.H = H; sa.tgt_len = tgt; sa.pos = posp;
      sa.qkv_bias = w->dec_qkv_b[l]; sa.rope = (local && w->dec_rope_local) ? w->dec_rope_local : w->dec_rope;
      sa.scale = c->attn_scale; sa.window = local ? c->local_window : 0;
      dec::HeadProjP hp{};
      hp.h = bf.h; hp.ldh = d; hp.ln_w = w->dec_ln1[l]; hp.eps = c->eps; hp.W = w->dec_qkv[l]; hp.ldw = d; hp.d = d;
      hp.ln_b = hf ? w->dec_ln1_b[l] : nullptr;
      MH_REQUIRE(!hf || (w->dec_ln1_b[l] && w->dec_ln2_b[l] && w->dec_ln3_b[l]), "decode: arch 2 needs the LayerNorm biases of layer %d", l);
      MH_TRY(launch_self_qkv_d<T>(sa, hp, inner, s));
      sk = dec::SkinnyP{};
      sk.A = bf.attn; sk.lda = inner; sk.W = w->dec_o[l]; sk.ldw = inner; sk.B = B; sk.N = d; sk.K = inner; sk.h = bf.h; sk.ldh = d;
      sk.bias = w->dec_o_b[l];
      MH_TRY(skinny_resid<T>(sk, s));
      dec::CrossAttnP ca{};
      const long kv_layer = (long)kvB * H * L * 64 * es;
      ca.k = (const char*)cross_kv + (long)(l * 2 + 0) * kv_layer; ca.v = (const char*)cross_kv + (long)(l * 2 + 1) * kv_layer;
      ca.out = bf.attn; ca.ldo = inner; ca.B = B; ca.H = H; ca.L = L; ca.kv_B = kv_group > 1 ? -kv_group : (kvB < Bfull ? kvB : 0);
      ca.q_bias = w->dec_cq_b[l]; ca.scale = c->attn_scale;
      MH_REQUIRE(!kv8, "decode: the fp8 cross K/V copy is not wired for the Whisper family");
      if (g_timing.buf) {
        ca.tstamp = g_timing.buf + 2L * bf.chain * g_timing.ring * c->n_dec_layers;
        ca.pos = posp; ca.ts_ring = g_timing.ring; ca.ts_layers = c->n_dec_layers; ca.ts_layer = l;
      }
      hp = dec::HeadProjP{};
      hp.h = bf.h; hp.ldh = d; hp.ln_w = w->dec_ln2[l]; hp.eps = c->eps; hp.W = w->dec_cq[l]; hp.ldw = d; hp.d = d;
      hp.ln_b = hf ? w->dec_ln2_b[l] : nullptr;
      MH_TRY(launch_cross_q_d<T>(ca, hp, s));
      sk = dec::SkinnyP{};
      sk.A = bf.attn; sk.lda = inner; sk.W = w->dec_co[l]; sk.ldw = inner; sk.B = B; sk.N = d; sk.K = inner; sk.h = bf.h; sk.ldh = d;
      sk.bias = w->dec_co_b[l];
      MH_TRY(skinny_resid<T>(sk, s));
      sk = dec::SkinnyP{};
      sk.A = bf.h; sk.lda = d; sk.ln_w = w->dec_ln3[l]; sk.eps = c->eps; sk.W = w->dec_wi[l]; sk.ldw = d; sk.B = B;
      sk.N = dff; sk.K = d; sk.out = bf.ff; sk.ldo = dff; sk.bias = w->dec_fc1_b[l];
      if (hf) { sk.ln_b = w->dec_ln3_b[l]; MH_TRY((skinny<T, dec::PRO_LAYERNORM, dec::SK_GELU_ERF, true>(sk, s))); }
      else MH_TRY((skinny<T, dec::PRO_RMSNORM, dec::SK_GELU_ERF, true>(sk, s)));
      sk = dec::SkinnyP{};
      sk.A = bf.ff; sk.lda = dff; sk.W = w->dec_wo[l]; sk.ldw = dff; sk.B = B; sk.N = d; sk.K = dff; sk.h = bf.h; sk.ldh = d;
      sk.bias = w->dec_fc2_b[l];
      MH_TRY(skinny_resid<T>(sk, s));
      continue;
    }
    // self attention
    const bool fused = fused_proj_enabled(d);                                  // cross-attention projects its own query
    const bool fused_self = fused && option(OPT_DECODE_FUSED_PROJ) == 1;       // (2: stand-alone QKV GEMV, fused cross-attention)
    dec::SelfAttnP sa{};
    sa.q = bf.q; sa.ldq = inner; sa.kc = (char*)bf.self_k + cache_off; sa.vc = (char*)bf.self_v + cache_off;
    sa.bias = w->dec_rel_bias; sa.prompt_mask = prompt_mask;
    sa.P = P; sa.out = bf.attn; sa.ldo = inner; sa.B = B; sa.H = H; sa.tgt_len = tgt; sa.pos = posp;
    if (fused_self) {
      dec::HeadProjP hp{};
      hp.h = bf.h; hp.ldh = d; hp.ln_w = w->dec_ln1[l]; hp.eps = c->eps; hp.W = w->dec_qkv[l]; hp.ldw = d; hp.d = d;
      MH_TRY(launch_self_qkv_d<T>(sa, hp, inner, s));
    } else {
      sk = dec::SkinnyP{};
      sk.A = bf.h; sk.lda = d; sk.ln_w = w->dec_ln1[l]; sk.eps = c->eps; sk.W = w->dec_qkv[l]; sk.ldw = d; sk.B = B;
      sk.N = 3 * inner; sk.K = d; sk.out = bf.q; sk.ldo = inner; sk.kc = (char*)bf.self_k + cache_off;
      sk.vc = (char*)bf.self_v + cache_off; sk.H = H; sk.tgt_len = tgt; sk.inner = inner; sk.pos = posp;
      MH_TRY((skinny<T, dec::PRO_RMSNORM, dec::SK_QKV>(sk, s)));
      hipLaunchKernelGGL(dec::dec_self_attn_kernel<T>, dim3(B * H), dim3(256), 0, s, sa);
      MH_TRY(check_launch("dec_self_attn_kernel"));
    }
    sk = dec::SkinnyP{};
    sk.A = bf.attn; sk.lda = inner; sk.W = w->dec_o[l]; sk.ldw = inner; sk.B = B; sk.N = d; sk.K = inner; sk.h = bf.h;
    sk.ldh = d;
    MH_TRY((skinny<T, dec::PRO_PLAIN, dec::SK_RESID>(sk, s)));
    // cross attention
    dec::CrossAttnP ca{};
    const long kv_layer = (long)kvB * H * L * 64 * es;
    ca.q = bf.q; ca.ldq = inner; ca.k = (const char*)cross_kv + (long)(l * 2 + 0) * kv_layer;
    ca.v = (const char*)cross_kv + (long)(l * 2 + 1) * kv_layer; ca.out = bf.attn; ca.ldo = inner;
    ca.B = B; ca.H = H; ca.L = L; ca.kv_B = kv_group > 1 ? -kv_group : (kvB < Bfull ? kvB : 0);
    if (kv8) {
      MH_REQUIRE(fused, "decode: the fp8 cross K/V copy needs decode_fused_proj = 1 and d_model a multiple of 128 <= 1024");
      const long slab = (long)kvB * H * L * 64;
      ca.k = (const char*)kv8 + (long)(l * 2 + 0) * slab; ca.v = (const char*)kv8 + (long)(l * 2 + 1) * slab;
      ca.kscale = kv8_scales + (long)(l * 2 + 0) * kvB * H; ca.vscale = kv8_scales + (long)(l * 2 + 1) * kvB * H;
    }
    if (g_timing.buf) {   // one region of ring x layers slots per chain (chain index = first row / rows of a full chain)
      ca.tstamp = g_timing.buf + 2L * bf.chain * g_timing.ring * c->n_dec_layers;
      ca.pos = posp; ca.ts_ring = g_timing.ring; ca.ts_layers = c->n_dec_layers; ca.ts_layer = l;
    }
    if (fused) {
      dec::HeadProjP hp{};
      hp.h = bf.h; hp.ldh = d; hp.ln_w = w->dec_ln2[l]; hp.eps = c->eps; hp.W = w->dec_cq[l]; hp.ldw = d; hp.d = d;
      MH_TRY(launch_cross_q_d<T>(ca, hp, s));
    } else {
      sk = dec::SkinnyP{};
      sk.A = bf.h; sk.lda = d; sk.ln_w = w->dec_ln2[l]; sk.eps = c->eps; sk.W = w->dec_cq[l]; sk.ldw = d; sk.B = B;
      sk.N = inner; sk.K = d; sk.out = bf.q; sk.ldo = inner;
      MH_TRY((skinny<T, dec::PRO_RMSNORM, dec::SK_STORE>(sk, s)));
      MH_TRY(launch_cross<T>(ca, s));
    }
    sk = dec::SkinnyP{};
    sk.A = bf.attn; sk.lda = inner; sk.W = w->dec_co[l]; sk.ldw = inner; sk.B = B; sk.N = d; sk.K = inner; sk.h = bf.h;
    sk.ldh = d;
    MH_TRY((skinny<T, dec::PRO_PLAIN, dec::SK_RESID>(sk, s)));
    // feed forward
    sk = dec::SkinnyP{};
    sk.A = bf.h; sk.lda = d; sk.ln_w = w->dec_ln3[l]; sk.eps = c->eps; sk.W = w->dec_wi[l]; sk.ldw = d; sk.B = B;
    sk.N = 2 * dff; sk.K = d; sk.out = bf.ff; sk.ldo = dff;
    MH_TRY((skinny<T, dec::PRO_RMSNORM, dec::SK_GEGLU>(sk, s)));
    sk = dec::SkinnyP{};
    sk.A = bf.ff; sk.lda = dff; sk.W = w->dec_wo[l]; sk.ldw = dff; sk.B = B; sk.N = d; sk.K = dff; sk.h = bf.h;
    sk.ldh = d;
    MH_TRY((skinny<T, dec::PRO_PLAIN, dec::SK_RESID>(sk, s)));
  }
  dec::SkinnyP sk{};
  sk.A = bf.h; sk.lda = d; sk.ln_w = w->dec_final_ln; sk.eps = c->eps; sk.W = w->lm_head; sk.ldw = d; sk.B = B;
  sk.N = c->vocab_out; sk.K = d; sk.out = bf.logits; sk.ldo = c->vocab_out;
  if (hf) { sk.ln_b = w->dec_final_ln_b; MH_TRY((skinny<T, dec::PRO_LAYERNORM, dec::SK_LOGITS>(sk, s))); }
  else MH_TRY((skinny<T, dec::PRO_RMSNORM, dec::SK_LOGITS>(sk, s)));
  if (!with_sampler) return MH_OK;
  hipLaunchKernelGGL(dec_sample_kernel<T>, dim3(smp.pair > 0 ? smp.pair : B), dim3(256), 0, s, smp);
  return check_launch("dec_sample_kernel");
}

// ---- step-wise decode (beam search) ----------------------------------------------------------------------------------
// h[b][:] = dec_embed[ids[b]] and the position word of the step
template <typename T>
__global__ __launch_bounds__(256) void step_embed_kernel(const int32_t* ids, const T* emb, int d, float* h, DecState* st, int pos,
                                                        const float* dec_pos, const uint8_t* prompt_mask, int P) {
  // dec_pos (arch 2): + decoder.embed_positions[position]; prompt_mask != NULL here means dec_pos_from_mask: the position is the
  // column minus this row's masked prompt columns
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0) { st->pos = pos; st->n_running = 0; st->ticket = 0; }
  __shared__ int s_off;
  if (threadIdx.x == 0) {
    int off = 0;
    if (dec_pos && prompt_mask)
      for (int i = 0; i < P; ++i) off += prompt_mask[(long)b * P + i] == 0;
    s_off = off;
  }
  __syncthreads();
  const T* e = emb + (long)ids[b] * d;
  const float* pe = dec_pos ? dec_pos + (long)(pos - s_off > 0 ? pos - s_off : 0) * d : nullptr;
  for (int i = threadIdx.x; i < d; i += 256) h[(long)b * d + i] = Elem<T>::to_f32(e[i]) + (pe ? pe[i] : 0.f);
}
// cache rows [l][b][h][0 .. n_pos) <- [l][src[b]][h][..]: gather into `tmp`, then copy back (in place would read rows that
// were already overwritten); one workgroup per (k|v, layer, row, head)
template <typename T>
__global__ __launch_bounds__(256) void cache_gather_kernel(const T* kc, const T* vc, T* tmp, const int32_t* src, int B, int H, int tgt,
                                                          int n_pos, long layer_stride) {
  const int h = blockIdx.x % H, b = (blockIdx.x / H) % B, l = (blockIdx.x / H / B) % (int)gridDim.y, kv = blockIdx.z;
  const T* from = (kv ? vc : kc) + (long)blockIdx.y * layer_stride + ((long)src[b] * H + h) * tgt * 64;
  T* to = tmp + ((((long)kv * gridDim.y + blockIdx.y) * B + b) * H + h) * (long)n_pos * 64;
  (void)l;
  const uint4* f4 = reinterpret_cast<const uint4*>(from);
  uint4* t4 = reinterpret_cast<uint4*>(to);
  const int n16 = n_pos * 64 * (int)sizeof(T) / 16;
  for (int i = threadIdx.x; i < n16; i += 256) t4[i] = f4[i];
}
template <typename T>
__global__ __launch_bounds__(256) void cache_scatter_kernel(T* kc, T* vc, const T* tmp, int B, int H, int tgt, int n_pos, long layer_stride) {
  const int h = blockIdx.x % H, b = (blockIdx.x / H) % B, kv = blockIdx.z;
  T* to = (kv ? vc : kc) + (long)blockIdx.y * layer_stride + ((long)b * H + h) * tgt * 64;
  const T* from = tmp + ((((long)kv * gridDim.y + blockIdx.y) * B + b) * H + h) * (long)n_pos * 64;
  const uint4* f4 = reinterpret_cast<const uint4*>(from);
  uint4* t4 = reinterpret_cast<uint4*>(to);
  const int n16 = n_pos * 64 * (int)sizeof(T) / 16;
  for (int i = threadIdx.x; i < n16; i += 256) t4[i] = f4[i];
}

}  // namespace
}  // namespace mh

namespace mh {
namespace {
struct PrefillBuf;
int64_t prefill_layout(const MhT5Config* c, int B, int np_max, void* base, int64_t size, PrefillBuf* out);
}  // namespace
}  // namespace mh

extern "C" int64_t mh_t5_decode_workspace_bytes(const MhT5Config* c, int B) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  if (!c || B <= 0) return -1;
  const int64_t es = es_of(c->dtype);
  const int inner = c->n_heads * 64;
  int64_t t = 0;
  t += align256((int64_t)B * c->d_model * 4);                                     // h
  t += align256((int64_t)B * inner * es) * 2;                                     // q, attn
  t += align256((int64_t)B * c->d_ff * es);                                       // ff
  t += align256((int64_t)B * c->vocab_out * 4);                                   // logits
  t += align256((int64_t)c->n_dec_layers * B * inner * c->tgt_len * es) * 2;      // self K, V caches
  t += align256(B) + align256((int64_t)B * 4) * 3 + align256(sizeof(DecState)) * kMaxChains;   // flags / state (+ pos_off)
  t += align256((int64_t)B * c->vocab_out * 4) * 3;                               // processed scores + LookbackBias history
  t += prefill_layout(c, B, c->tgt_len - 1, nullptr, 0, nullptr);                  // batched prompt prefill
  return t;
}

namespace mh {
namespace {

// rows r = b*np + i  <-  dec_embed[prompt[b][i]]   (fp32 residual stream of the prompt prefill)
template <typename T>
__global__ __launch_bounds__(256) void prefill_embed_kernel(const int32_t* __restrict__ prompt, int P, int np,
                                                           const T* __restrict__ emb, int d, float* __restrict__ h,
                                                           const float* __restrict__ dec_pos, const uint8_t* __restrict__ pos_mask) {
  // dec_pos (arch 2): + decoder.embed_positions[position of column i]; pos_mask != NULL (dec_pos_from_mask): minus the row's masked columns
  const int r = blockIdx.x, b = r / np, i = r - b * np;
  __shared__ int s_off;
  if (threadIdx.x == 0) {
    int off = 0;
    if (dec_pos && pos_mask)
      for (int k = 0; k < P; ++k) off += pos_mask[(long)b * P + k] == 0;
    s_off = off;
  }
  __syncthreads();
  const T* e = emb + (long)prompt[(long)b * P + i] * d;
  const float* pe = dec_pos ? dec_pos + (long)(i - s_off > 0 ? i - s_off : 0) * d : nullptr;
  for (int k = threadIdx.x; k < d; k += 256) h[(long)r * d + k] = Elem<T>::to_f32(e[k]) + (pe ? pe[k] : 0.f);
}

struct PrefillBuf {
  float* h; void* n; void* q; void* attn; void* ff; void* vt; void* cross_vt;
  int np_pad, Lpad;
};

int64_t prefill_layout(const MhT5Config* c, int B, int np_max, void* base, int64_t size, PrefillBuf* out) {
  Arena ar(base, size);
  const int64_t es = es_of(c->dtype), rows = (int64_t)B * np_max, inner = c->n_heads * 64;
  PrefillBuf t;
  t.np_pad = round_up(np_max, 64);
  t.Lpad = round_up(c->src_len, 64);
  t.h = (float*)ar.take(rows * c->d_model * 4);
  t.n = ar.take(rows * c->d_model * es);
  t.q = ar.take(rows * inner * es);
  t.attn = ar.take(rows * inner * es);
  t.ff = ar.take(rows * c->d_ff * es);
  t.vt = ar.take((int64_t)B * inner * t.np_pad * es);
  t.cross_vt = ar.take((int64_t)c->n_dec_layers * B * inner * t.Lpad * es);
  if (out) *out = t;
  return ar.off;
}

// Batched prompt prefill: positions 0 .. P-2 of every row go through the decoder stack at once (MFMA GEMMs +
// flash attention), filling the self-attention K/V caches exactly as P-1 single-token steps would
// (HF prefill: SURVEY.md appendix A.1).  Position P-1 is left to the per-token loop, which also produces the
// first sampled token.  Left-pad keys are masked, left-pad query rows produce unused garbage, as in HF.
int prefill_prompt(const MhT5Config* c, const MhT5Weights* w, const void* cross_kv, int B, int kvB, const int32_t* prompt,
                   const uint8_t* prompt_mask, int P, int np, void* self_k, void* self_v, const PrefillBuf& pb, hipStream_t s) {
  // P = row stride of `prompt` / `prompt_mask`, np <= P = positions that go through the stack
  const int rows = B * np;
  const int d = c->d_model, H = c->n_heads, inner = H * 64, dff = c->d_ff, L = c->src_len, tgt = c->tgt_len;
  const int es = es_of(c->dtype);
  const int np_pad = round_up(np, 64);
  // arch 1 (VarWhisperDecoderLayer, modeling_varwhisper.py:633-741) through the same batched form: biased projections,
  // rotate-half RoPE on q and on the cached keys, scores / 8 without a relative bias, fc1 -> gelu(erf) -> fc2.  A local
  // (windowed) layer takes its own rotary table and the causal attention gets the band |k - q| <= local_window on top -- the
  // keys the token-by-token step attends (decode_kernels.hpp `window`).
  // arch 2 (HF WhisperDecoderLayer): the same with affine LayerNorms, absolute positions added to the embedding, no rotation
  const bool wh = c->arch >= 1, hf = c->arch == 2;
  if (wh && !hf) MH_REQUIRE(w->dec_rope != nullptr, "prefill: the Whisper family needs its rotary table");
  if (hf) MH_REQUIRE(w->dec_pos && w->dec_final_ln_b, "prefill: arch 2 needs decoder.embed_positions and the LayerNorm biases");
  const float* dpos = hf ? w->dec_pos : nullptr;
  const uint8_t* pmask = (hf && c->dec_pos_from_mask) ? prompt_mask : nullptr;
  if (c->dtype == MH_BF16)
    hipLaunchKernelGGL(prefill_embed_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, prompt, P, np, (const bf16_t*)w->dec_embed, d, pb.h, dpos, pmask);
  else
    hipLaunchKernelGGL(prefill_embed_kernel<float>, dim3(rows), dim3(256), 0, s, prompt, P, np, (const float*)w->dec_embed, d, pb.h, dpos, pmask);
  MH_TRY(check_launch("prefill_embed_kernel"));
  if (hipMemsetAsync(pb.vt, 0, (size_t)B * inner * np_pad * es, s) != hipSuccess) return check_launch("memset prefill vt");
  if (hipMemsetAsync(pb.cross_vt, 0, (size_t)c->n_dec_layers * kvB * inner * pb.Lpad * es, s) != hipSuccess)
    return check_launch("memset cross vt");
  const long kv_layer = (long)kvB * H * L * 64;   // elements per (layer, k|v) slab of cross_kv (kvB = B/2 under CFG)
  for (int l = 0; l < c->n_dec_layers; ++l)
    MH_TRY(transpose_v((const char*)cross_kv + (long)(l * 2 + 1) * kv_layer * es, (long)H * L * 64, (long)L * 64, L,
                       (char*)pb.cross_vt + (long)l * kvB * inner * pb.Lpad * es, pb.Lpad, kvB, H, c->dtype, s));
  MhGemm g;
  for (int l = 0; l < c->n_dec_layers; ++l) {
    char* kc = (char*)self_k + (long)l * B * H * tgt * 64 * es;
    char* vc = (char*)self_v + (long)l * B * H * tgt * 64 * es;
    // self attention over the prompt
    MH_TRY(pre_norm(c, pb.h, w->dec_ln1[l], w->dec_ln1_b[l], pb.n, rows, c->dtype, s));
    g = MhGemm{};
    g.A = pb.n; g.lda = d; g.W = w->dec_qkv[l]; g.ldw = d; g.C = pb.q; g.ldc = inner; g.M = rows; g.N = 3 * inner; g.K = d;
    g.dtype = c->dtype; g.epilogue = MH_EPI_QKV_CACHE; g.n_split = inner; g.C2 = kc; g.C3 = vc; g.C4 = pb.vt; g.kv_B = B;
    g.kv_H = H; g.kv_L = np; g.kv_Lpad = np_pad; g.cache_len = tgt;
    if (wh) g.bias = w->dec_qkv_b[l];
    MH_TRY(gemm(g, s));
    const bool local = is_local_layer(c, l);
    if (wh && !hf) {   // rotate-half RoPE on q (all prompt positions) and on the keys just cached; position = column of the padded prompt
      const long wq = (long)rows * H * 4, wk = (long)B * H * np * 4;
      const float* rope = (local && w->dec_rope_local) ? w->dec_rope_local : w->dec_rope;
      if (c->dtype == MH_BF16) {
        hipLaunchKernelGGL(mh::rope_qk_kernel<bf16_t>, dim3((unsigned)((wq + 255) / 256)), dim3(256), 0, s, (bf16_t*)pb.q, inner, (long)rows, np, H, rope);
        hipLaunchKernelGGL(mh::rope_cache_kernel<bf16_t>, dim3((unsigned)((wk + 255) / 256)), dim3(256), 0, s, (bf16_t*)kc, (long)B * H, tgt, np, rope);
      } else {
        hipLaunchKernelGGL(mh::rope_qk_kernel<float>, dim3((unsigned)((wq + 255) / 256)), dim3(256), 0, s, (float*)pb.q, inner, (long)rows, np, H, rope);
        hipLaunchKernelGGL(mh::rope_cache_kernel<float>, dim3((unsigned)((wk + 255) / 256)), dim3(256), 0, s, (float*)kc, (long)B * H, tgt, np, rope);
      }
      MH_TRY(check_launch("rope (prefill)"));
    }
    AttnArgs a{};
    a.q = pb.q; a.q_rs = (long)inner * es; a.q_bs = (long)np * inner * es;
    a.k = kc; a.k_rs = 64L * es; a.k_hs = (long)tgt * 64 * es; a.k_bs = (long)H * tgt * 64 * es;
    a.vt = pb.vt; a.Lkpad = np_pad; a.vt_hs = 64L * np_pad * es; a.vt_bs = (long)H * 64 * np_pad * es;
    if (!wh) { a.bias = w->dec_rel_bias; a.bias_hs = tgt; a.bias_center = 0; a.bias_sign = -1; a.bias_min = 0; a.bias_max = tgt - 1; }
    a.key_mask = prompt_mask; a.mask_ld = P; a.mask_len = np;
    a.out = pb.attn; a.out_rs = (long)inner * es; a.out_bs = (long)np * inner * es;
    a.Lq = np; a.Lk = np; a.scale = wh ? c->attn_scale : 1.0f; a.band = local ? -c->local_window : 0; a.causal = 1; a.q_pos0 = 0;
    MH_TRY(attention_general(a, B, H, c->dtype, s));
    g = MhGemm{};
    g.A = pb.attn; g.lda = inner; g.W = w->dec_o[l]; g.ldw = inner; g.C = pb.h; g.ldc = d; g.M = rows; g.N = d; g.K = inner;
    g.dtype = c->dtype; g.epilogue = MH_EPI_RESID;
    if (wh) g.bias = w->dec_o_b[l];
    MH_TRY(gemm(g, s));
    // cross attention of every prompt position over the encoder keys
    MH_TRY(pre_norm(c, pb.h, w->dec_ln2[l], w->dec_ln2_b[l], pb.n, rows, c->dtype, s));
    g = MhGemm{};
    g.A = pb.n; g.lda = d; g.W = w->dec_cq[l]; g.ldw = d; g.C = pb.q; g.ldc = inner; g.M = rows; g.N = inner; g.K = d;
    g.dtype = c->dtype; g.epilogue = MH_EPI_STORE;
    if (wh) g.bias = w->dec_cq_b[l];
    MH_TRY(gemm(g, s));
    for (int b0 = 0; b0 < B; b0 += kvB) {   // under CFG both halves of the batch attend to the same kvB encoder rows
      a = AttnArgs{};
      a.q = (char*)pb.q + (long)b0 * np * inner * es; a.q_rs = (long)inner * es; a.q_bs = (long)np * inner * es;
      a.k = (const char*)cross_kv + (long)(l * 2 + 0) * kv_layer * es; a.k_rs = 64L * es; a.k_hs = (long)L * 64 * es;
      a.k_bs = (long)H * L * 64 * es;
      a.vt = (char*)pb.cross_vt + (long)l * kvB * inner * pb.Lpad * es; a.Lkpad = pb.Lpad; a.vt_hs = 64L * pb.Lpad * es;
      a.vt_bs = (long)H * 64 * pb.Lpad * es;
      a.out = (char*)pb.attn + (long)b0 * np * inner * es; a.out_rs = (long)inner * es; a.out_bs = (long)np * inner * es;
      a.Lq = np; a.Lk = L; a.scale = wh ? c->attn_scale : 1.0f;
      MH_TRY(attention_general(a, kvB, H, c->dtype, s));
    }
    g = MhGemm{};
    g.A = pb.attn; g.lda = inner; g.W = w->dec_co[l]; g.ldw = inner; g.C = pb.h; g.ldc = d; g.M = rows; g.N = d; g.K = inner;
    g.dtype = c->dtype; g.epilogue = MH_EPI_RESID;
    if (wh) g.bias = w->dec_co_b[l];
    MH_TRY(gemm(g, s));
    // feed forward
    MH_TRY(pre_norm(c, pb.h, w->dec_ln3[l], w->dec_ln3_b[l], pb.n, rows, c->dtype, s));
    g = MhGemm{};
    g.A = pb.n; g.lda = d; g.W = w->dec_wi[l]; g.ldw = d; g.C = pb.ff; g.ldc = dff; g.M = rows; g.K = d;
    g.dtype = c->dtype;
    if (wh) { g.N = dff; g.epilogue = MH_EPI_BIAS_GELU_ERF; g.bias = w->dec_fc1_b[l]; }     // fc1 -> gelu(erf) (modeling_varwhisper.py:633-741)
    else { g.N = 2 * dff; g.epilogue = MH_EPI_GEGLU; }
    MH_TRY(gemm(g, s));
    g = MhGemm{};
    g.A = pb.ff; g.lda = dff; g.W = w->dec_wo[l]; g.ldw = dff; g.C = pb.h; g.ldc = d; g.M = rows; g.N = d; g.K = dff;
    g.dtype = c->dtype; g.epilogue = MH_EPI_RESID;
    if (wh) g.bias = w->dec_fc2_b[l];
    MH_TRY(gemm(g, s));
  }
  return MH_OK;
}

// Independent rows => independent "chains": the batch is cut into contiguous blocks of rows and every
// block runs its own captured decode step on its own stream.  A decode step is ~110 dependent, mostly
// tiny kernels (each costs a few microseconds of dispatch + drain whatever its size), so one chain leaves
// the GPU idle most of the time; several chains overlap one another's dispatch gaps and let the
// HBM-bound cross-attention of one chain run under the latency-bound GEMVs of the others.  Results do not
// depend on the chain count (every kernel is batch-invariant by construction).
int pick_chains(int B) {
  int n = B >= 16 ? 2 : 1;   // measured on MI355X (B=32, base): 1 -> 26.7k, 2 -> 28.0k, 4 -> 17.6k tok/s (host graph-launch bound)
  if (option(OPT_DECODE_CHAINS) >= 1) n = (int)option(OPT_DECODE_CHAINS);
  if (n > kMaxChains) n = kMaxChains;
  if (n > B) n = B;
  return n;
}

struct ChainPool {   // extra streams + fork/join events of ONE device, created on first use under that device
  hipStream_t streams[kMaxChains] = {};
  hipEvent_t fork = nullptr, join[kMaxChains] = {};
  bool ready = false;
  int init() {
    if (ready) return MH_OK;
    if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return check_launch("event create");
    for (int i = 0; i < kMaxChains; ++i) {
      if (hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&join[i], hipEventDisableTiming) != hipSuccess)
        return check_launch("chain stream create");
    }
    ready = true;
    return MH_OK;
  }
};
// one pool per device; mh_t5_generate holds the pool's mutex for the whole call, so two engines (or two host
// threads) decoding on the same GPU take turns instead of sharing fork / join events
struct DevicePool { ChainPool pool; std::mutex mu; };
DevicePool* device_pool(int dev) {
  static std::mutex table_mu;
  static std::map<int, DevicePool*> table;
  std::lock_guard<std::mutex> g(table_mu);
  auto it = table.find(dev);
  if (it == table.end()) it = table.emplace(dev, new DevicePool()).first;
  return it->second;
}

}  // namespace
}  // namespace mh

// ---- e4m3 copy of the cross-attention K / V -------------------------------------------------------------------------
// one workgroup per (layer, k|v, row, head) slab of L x 64 bf16: absolute maximum, then x / scale -> OCP e4m3
__global__ __launch_bounds__(256) void kv_quant_fp8_kernel(const bf16_t* src, uint8_t* dst, float* scales, int L) {
  __shared__ float red[4];
  const long slab = (long)blockIdx.x * L * 64;
  const uint4* s16 = reinterpret_cast<const uint4*>(src + slab);
  const int n16 = L * 8;   // 16-byte vectors (8 elements) in the slab
  float mx = 0.f;
  for (int i = threadIdx.x; i < n16; i += 256) {
    const uint4 v = s16[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mx = fmaxf(mx, fabsf(__uint_as_float(w[j] << 16)));
      mx = fmaxf(mx, fabsf(__uint_as_float(w[j] & 0xffff0000u)));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float scale = mx > 0.f ? mx / 448.0f : 1.0f;   // 448 = largest finite e4m3 value
  const float inv = 1.0f / scale;
  if (threadIdx.x == 0) scales[blockIdx.x] = scale;
  uint2* d8 = reinterpret_cast<uint2*>(dst + slab);
  for (int i = threadIdx.x; i < n16; i += 256) {
    const uint4 v = s16[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(w[j] << 16) * inv; f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u) * inv; }
    int o0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    o0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], o0, true);
    int o1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    o1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], o1, true);
    d8[i] = make_uint2((uint32_t)o0, (uint32_t)o1);
  }
}

extern "C" int64_t mh_t5_cross_kv_fp8_bytes(const MhT5Config* c, int B) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  if (!c || B <= 0) return -1;
  const int64_t data = (int64_t)c->n_dec_layers * 2 * B * c->n_heads * c->src_len * 64;
  return align256(data) + align256((int64_t)c->n_dec_layers * 2 * B * c->n_heads * 4);
}

extern "C" int mh_t5_quantize_cross_kv(const MhT5Config* c, const void* cross_kv, int B, void* out, void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_cfg(c, "mh_t5_quantize_cross_kv"));
  MH_REQUIRE(cross_kv && out && B > 0, "mh_t5_quantize_cross_kv: null argument");
  MH_REQUIRE(c->dtype == MH_BF16, "mh_t5_quantize_cross_kv: needs bf16 storage");
  const int64_t data = (int64_t)c->n_dec_layers * 2 * B * c->n_heads * c->src_len * 64;
  const int slabs = c->n_dec_layers * 2 * B * c->n_heads;
  hipLaunchKernelGGL(kv_quant_fp8_kernel, dim3(slabs), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)cross_kv,
                     (uint8_t*)out, reinterpret_cast<float*>((char*)out + align256(data)), c->src_len);
  return check_launch("kv_quant_fp8_kernel");
}

namespace mh {
// ------------------------------------------------------------------------------------------------
// Instantiated step graphs kept ACROSS mh_t5_generate calls.  A chain's step graph bakes in nothing but addresses, sizes, the
// sampling struct and the kernel choice (the position and every per-call state live in device memory), so a later call
// whose inputs are byte-for-byte the same description -- the normal case of an engine that decodes window after window out
// of the same workspace with the same prompt length -- replays the graph of the earlier one instead of capturing and
// instantiating ~75 nodes again.  The key is the exact byte string of everything enqueue_step() receives plus every option
// value and the timing hook (exact compare: a spurious difference costs a capture, never a wrong graph).  Small LRU; an entry in
// use by a concurrent call is never shared (hipGraphExec objects are single-flight) nor evicted.  Option decode_graph_cache = 0
// switches it off.
struct StepGraphEntry {
  std::vector<unsigned char> key;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  uint64_t stamp = 0;
  bool in_use = false;
};
constexpr size_t kStepGraphCacheMax = 16;
static std::mutex g_step_graph_mu;
static std::vector<StepGraphEntry*> g_step_graphs;
static uint64_t g_step_graph_clock = 0;
static std::atomic<long> g_step_graph_hits{0}, g_step_graph_misses{0};

static StepGraphEntry* step_graph_acquire(const std::vector<unsigned char>& key) {
  std::lock_guard<std::mutex> lk(g_step_graph_mu);
  for (StepGraphEntry* e : g_step_graphs)
    if (!e->in_use && e->key == key) {
      e->in_use = true;
      e->stamp = ++g_step_graph_clock;
      return e;
    }
  return nullptr;
}

// takes ownership of graph / exec; returns the entry (in use) -- or nullptr when the cache is full of entries in use
static StepGraphEntry* step_graph_insert(std::vector<unsigned char>&& key, hipGraph_t graph, hipGraphExec_t exec) {
  std::lock_guard<std::mutex> lk(g_step_graph_mu);
  while (g_step_graphs.size() >= kStepGraphCacheMax) {
    int victim = -1;
    for (size_t i = 0; i < g_step_graphs.size(); ++i)
      if (!g_step_graphs[i]->in_use && (victim < 0 || g_step_graphs[i]->stamp < g_step_graphs[victim]->stamp)) victim = (int)i;
    if (victim < 0) return nullptr;
    StepGraphEntry* v = g_step_graphs[victim];
    (void)hipGraphExecDestroy(v->exec);
    (void)hipGraphDestroy(v->graph);
    delete v;
    g_step_graphs.erase(g_step_graphs.begin() + victim);
  }
  StepGraphEntry* e = new StepGraphEntry();
  e->key = std::move(key);
  e->graph = graph;
  e->exec = exec;
  e->in_use = true;
  e->stamp = ++g_step_graph_clock;
  g_step_graphs.push_back(e);
  return e;
}

static void step_graph_release(StepGraphEntry* e) {
  std::lock_guard<std::mutex> lk(g_step_graph_mu);
  e->in_use = false;
}
}  // namespace mh

extern "C" int mh_t5_step_graph_cache_stats(long* hits, long* misses, int reset) {
  if (hits) *hits = mh::g_step_graph_hits.load();
  if (misses) *misses = mh::g_step_graph_misses.load();
  if (reset) { mh::g_step_graph_hits.store(0); mh::g_step_graph_misses.store(0); }
  return MH_OK;
}

extern "C" int mh_t5_generate(const MhT5Config* c, const MhT5Weights* w, const void* cross_kv, int B,
                              const int32_t* prompt, const uint8_t* prompt_mask, int P, const uint8_t* eos_table,
                              const MhSampling* sp, int32_t* tokens, int32_t* n_steps_out, float* logits_dump,
                              const int32_t* forced, void* workspace, int64_t workspace_bytes, int poll_every,
                              void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_cfg(c, "mh_t5_generate"));
  MH_REQUIRE(w && cross_kv && prompt && eos_table && sp && tokens && n_steps_out && workspace,
             "mh_t5_generate: null argument");
  MH_REQUIRE(stream != nullptr, "mh_t5_generate: needs a non-default stream (hipGraph capture)");
  MH_REQUIRE(B > 0 && B <= 64, "mh_t5_generate: batch %d not in [1, 64] (shard larger batches on the host)", B);
  MH_REQUIRE(P >= 1 && P < sp->max_length, "mh_t5_generate: prompt length %d must be in [1, max_length)", P);
  MH_REQUIRE(sp->max_length <= c->tgt_len, "mh_t5_generate: max_length %d exceeds tgt_len %d", sp->max_length, c->tgt_len);
  MH_REQUIRE(sp->temperature > 0.f, "mh_t5_generate: temperature must be > 0");
  MH_REQUIRE(sp->n_sos >= 0 && sp->n_sos <= 16, "mh_t5_generate: too many sos ids");
  const bool cfg = sp->cfg_scale > 1.0f;
  MH_REQUIRE(!cfg || B % 2 == 0, "mh_t5_generate: classifier-free guidance needs an even batch (negative rows, then prompt rows)");
  MH_REQUIRE(sp->n_cond >= 0 && sp->n_cond <= 3, "mh_t5_generate: n_cond %d not in [0, 3]", sp->n_cond);
  for (int j = 0; j < sp->n_cond; ++j)
    MH_REQUIRE(sp->cond_temp[j] > 0.f && sp->cond_offset[j] >= 1, "mh_t5_generate: bad conditional temperature rule %d", j);
  MH_REQUIRE(sp->tok_flags || (sp->n_cond == 0 && !sp->lookback_types_first),
             "mh_t5_generate: tok_flags is required by the conditional temperature / types_first lookback processors");
  const int kvB = cfg ? B / 2 : B;
  MH_REQUIRE(!sp->cross_kv_fp8 || c->dtype == MH_BF16, "mh_t5_generate: cross_kv_fp8 needs bf16 storage");
  MH_REQUIRE(workspace_bytes >= mh_t5_decode_workspace_bytes(c, B), "mh_t5_generate: workspace too small");
  MH_REQUIRE(c->arch != 2 || (w->dec_pos && w->dec_final_ln_b), "mh_t5_generate: arch 2 needs decoder.embed_positions and the LayerNorm biases");
  hipStream_t s = (hipStream_t)stream;
  const int es = es_of(c->dtype), H = c->n_heads, inner = H * 64, d = c->d_model, V = c->vocab_out;

  Arena ar(workspace, workspace_bytes);
  DecBuffers all;
  all.h = (float*)ar.take((int64_t)B * d * 4);
  all.q = ar.take((int64_t)B * inner * es);
  all.attn = ar.take((int64_t)B * inner * es);
  all.ff = ar.take((int64_t)B * c->d_ff * es);
  all.logits = (float*)ar.take((int64_t)B * V * 4);
  all.self_k = ar.take((int64_t)c->n_dec_layers * B * inner * c->tgt_len * es);
  all.self_v = ar.take((int64_t)c->n_dec_layers * B * inner * c->tgt_len * es);
  all.finished = (uint8_t*)ar.take(B);
  all.finish_col = (int32_t*)ar.take((int64_t)B * 4);
  all.last_ts = (int32_t*)ar.take((int64_t)B * 4);
  int32_t* pos_off = (int32_t*)ar.take((int64_t)B * 4);
  DecState* st_all = (DecState*)ar.take((int64_t)align256(sizeof(DecState)) * kMaxChains);
  float* proc = (float*)ar.take((int64_t)B * V * 4);
  float* hist_scores = (float*)ar.take((int64_t)B * V * 4 * 2);
  MH_REQUIRE(ar.ok() && hist_scores, "mh_t5_generate: arena overflow");
  // a CFG pair spans both halves of the batch and the (batch-wide) conditional temperature reads row 0's history: one chain
  const int n_chains = (cfg || (sp->n_cond > 0 && !sp->cond_per_row)) ? 1 : pick_chains(B);
  const int rows_per = ceil_div(B, n_chains);
  int dev = 0;   // the device that owns the caller's stream (not whatever happens to be current)
  if (hipStreamGetDevice(s, &dev) != hipSuccess) { (void)hipGetLastError(); if (hipGetDevice(&dev) != hipSuccess) return check_launch("hipGetDevice"); }
  struct DeviceGuard {   // streams / events of the pool are created under the stream's device; restored on every return path
    int prev = -1;
    explicit DeviceGuard(int want) { if (hipGetDevice(&prev) == hipSuccess && prev != want) (void)hipSetDevice(want); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  } device_guard(dev);
  DevicePool* dp = device_pool(dev);
  std::lock_guard<std::mutex> pool_guard(dp->mu);
  ChainPool& g_pool = dp->pool;
  MH_TRY(g_pool.init());
  hipStream_t chain_stream[kMaxChains];
  for (int i = 0; i < kMaxChains; ++i) chain_stream[i] = g_pool.streams[i];

  // tokens[:, :P] = prompt; the remainder is produced by the sampler
  if (hipMemcpy2DAsync(tokens, (size_t)sp->max_length * 4, prompt, (size_t)P * 4, (size_t)P * 4, B,
                       hipMemcpyDeviceToDevice, s) != hipSuccess)
    return check_launch("prompt copy");
  // batched prompt prefill (positions 0..P-2); MH_DECODE_PREFILL=0 feeds the prompt token by token instead
  int start_pos = 0;
  {
    if (P > 1 && option(OPT_DECODE_PREFILL) != 0) {
      PrefillBuf pb;
      const int64_t used_dec = ar.off;
      prefill_layout(c, B, P - 1, (char*)workspace + used_dec, workspace_bytes - used_dec, &pb);
      MH_REQUIRE(used_dec + prefill_layout(c, B, P - 1, nullptr, 0, nullptr) <= workspace_bytes,
                 "mh_t5_generate: workspace too small for the prompt prefill");
      MH_TRY(prefill_prompt(c, w, cross_kv, B, kvB, prompt, prompt_mask, P, P - 1, all.self_k, all.self_v, pb, s));
      start_pos = P - 1;
    }
  }
  if (hipEventRecord(g_pool.fork, s) != hipSuccess) return check_launch("fork record");

  const bool bf16 = c->dtype == MH_BF16;
  hipGraph_t graphs[kMaxChains] = {};
  hipGraphExec_t execs[kMaxChains] = {};
  StepGraphEntry* cached[kMaxChains] = {};     // chains whose graph lives in the cross-call cache (not destroyed below)
  DecState* states[kMaxChains] = {};
  int used = 0, rc = MH_OK;
  for (int ci = 0; ci < n_chains && rc == MH_OK; ++ci) {
    const int b0 = ci * rows_per;
    const int Bc = (b0 + rows_per <= B) ? rows_per : B - b0;
    if (Bc <= 0) break;
    hipStream_t cs = chain_stream[ci];
    if (hipStreamWaitEvent(cs, g_pool.fork, 0) != hipSuccess) { rc = check_launch("fork wait"); break; }
    DecBuffers bf = all;
    bf.chain = ci;
    bf.h = all.h + (long)b0 * d;
    bf.q = (char*)all.q + (long)b0 * inner * es;
    bf.attn = (char*)all.attn + (long)b0 * inner * es;
    bf.ff = (char*)all.ff + (long)b0 * c->d_ff * es;
    bf.logits = all.logits + (long)b0 * V;
    bf.self_k = (char*)all.self_k + (long)b0 * inner * c->tgt_len * es;
    bf.self_v = (char*)all.self_v + (long)b0 * inner * c->tgt_len * es;
    bf.finished = all.finished + b0;
    bf.st = (DecState*)((char*)st_all + (long)ci * align256(sizeof(DecState)));
    states[ci] = bf.st;
    const void* ckv = (const char*)cross_kv + (long)b0 * H * c->src_len * 64 * es;
    const uint8_t* pm = prompt_mask ? prompt_mask + (long)b0 * P : nullptr;

    SampleP smp{};
    smp.logits = bf.logits; smp.ldl = V; smp.V = V; smp.tokens = tokens; smp.max_length = sp->max_length;
    smp.forced = forced; smp.eos_table = eos_table; smp.finished = all.finished; smp.finish_col = all.finish_col;
    smp.last_ts_val = all.last_ts; smp.logits_dump = logits_dump; smp.dec_embed = w->dec_embed; smp.h = bf.h;
    smp.d = d; smp.sp = *sp; smp.st = bf.st; smp.B = B; smp.P = P; smp.b0 = b0;
    smp.proc = proc; smp.hist_scores = hist_scores; smp.pair = cfg ? B / 2 : 0; smp.chain_rows = Bc;
    if (c->arch == 2) {
      smp.dec_pos = w->dec_pos;
      smp.pos_off = c->dec_pos_from_mask ? pos_off : nullptr;
      smp.prompt_mask = prompt_mask;
    }

    if (bf16) hipLaunchKernelGGL(dec_init_kernel<bf16_t>, dim3(Bc), dim3(256), 0, cs, smp, Bc, start_pos);
    else hipLaunchKernelGGL(dec_init_kernel<float>, dim3(Bc), dim3(256), 0, cs, smp, Bc, start_pos);
    rc = check_launch("dec_init_kernel");
    if (rc != MH_OK) break;
    const void* kv8 = nullptr;
    const float* kv8_scales = nullptr;
    if (sp->cross_kv_fp8) {   // packed e4m3 copy: data, then (256-byte aligned) the scales
      const int64_t data_bytes = (int64_t)c->n_dec_layers * 2 * kvB * H * c->src_len * 64;
      kv8 = (const char*)sp->cross_kv_fp8 + (long)b0 * H * c->src_len * 64;
      kv8_scales = reinterpret_cast<const float*>((const char*)sp->cross_kv_fp8 + align256(data_bytes)) + (long)b0 * H;
    }
    // one step of this chain (every kernel reads the position from device memory) as a graph for replay: an earlier call's, if
    // its description is byte for byte this one's, else captured now
    std::vector<unsigned char> key;
    if (option(OPT_DECODE_GRAPH_CACHE) != 0) {
      auto put = [&key](const void* p, size_t n) { key.insert(key.end(), (const unsigned char*)p, (const unsigned char*)p + n); };
      MhT5Config cc = *c;
      cc.options = nullptr;
      long opts[OPT_COUNT];
      for (int o = 0; o < OPT_COUNT; ++o) opts[o] = option(o);
      int dev_id = 0;
      (void)hipGetDevice(&dev_id);
      const void* ptrs[] = {ckv, pm, kv8, kv8_scales, (const void*)g_timing.buf};
      const int ints[] = {Bc, B, kvB, P, g_timing.ring, dev_id, bf16 ? 1 : 0};
      put(&cc, sizeof(cc)); put(opts, sizeof(opts)); put(w, sizeof(*w)); put(ptrs, sizeof(ptrs)); put(ints, sizeof(ints));
      SampleP smp_key = smp;          // (seed and rng_row0 reach the sampler through DecState, not through the graph)
      smp_key.sp.seed = 0;
      smp_key.sp.rng_row0 = 0;
      put(&bf, sizeof(bf)); put(&smp_key, sizeof(smp_key));
      if (StepGraphEntry* e = step_graph_acquire(key)) {
        cached[ci] = e;
        execs[ci] = e->exec;
        g_step_graph_hits.fetch_add(1);
        ++used;
        continue;
      }
      g_step_graph_misses.fetch_add(1);
    }
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) { rc = check_launch("begin capture"); break; }
    int rce = bf16 ? enqueue_step<bf16_t>(c, w, ckv, Bc, B, kvB, pm, P, bf, smp, cs, kv8, kv8_scales)
                   : enqueue_step<float>(c, w, ckv, Bc, B, kvB, pm, P, bf, smp, cs, kv8, kv8_scales);
    hipError_t ce = hipStreamEndCapture(cs, &graphs[ci]);
    ++used;
    if (rce != MH_OK) { rc = rce; break; }
    if (ce != hipSuccess || !graphs[ci]) { set_error("mh_t5_generate: stream capture failed: %s", hipGetErrorString(ce)); rc = MH_ERR_LAUNCH; break; }
    if (hipGraphInstantiate(&execs[ci], graphs[ci], nullptr, nullptr, 0) != hipSuccess) { rc = check_launch("graph instantiate"); break; }
    if (!key.empty()) {
      if (StepGraphEntry* e = step_graph_insert(std::move(key), graphs[ci], execs[ci])) {
        cached[ci] = e;          // the cache owns graph and exec now
        graphs[ci] = nullptr;
      }
    }
  }

  const int total_steps = sp->max_length - 1 - start_pos;   // positions start_pos .. max_length-2 are fed
  if (poll_every <= 0) poll_every = 16;
  // One launcher per chain.  Replaying a ~110-node graph costs ~0.4 ms of HOST time on ROCm 7.2, so with more
  // than one chain the launches are issued from one host thread per chain (the chains are independent: each
  // polls only its own "rows still running" word and stops launching when its rows are done).
  auto run_chain = [&](int ci) -> int {
    int step = 0;
    while (step < total_steps) {
      const int burst = total_steps - step < poll_every ? total_steps - step : poll_every;
      for (int i = 0; i < burst; ++i) {
        if (hipGraphLaunch(execs[ci], chain_stream[ci]) != hipSuccess) return MH_ERR_LAUNCH;
      }
      step += burst;
      if (step < total_steps && !forced) {
        int running = 1;
        if (hipMemcpyAsync(&running, &states[ci]->n_running, 4, hipMemcpyDeviceToHost, chain_stream[ci]) != hipSuccess ||
            hipStreamSynchronize(chain_stream[ci]) != hipSuccess)
          return MH_ERR_LAUNCH;
        if (running == 0) break;
      }
    }
    return MH_OK;
  };
  if (rc == MH_OK) {
    int rcs[kMaxChains] = {};
    if (used <= 1) {
      if (used == 1) rcs[0] = run_chain(0);
    } else if (option(OPT_DECODE_LAUNCH_THREADS) == 0) {
      // one thread, the chains' steps interleaved (slower on the host side; the same kernels with the same arguments)
      bool alive[kMaxChains];
      for (int ci = 0; ci < used; ++ci) alive[ci] = true;
      for (int step = 0; step < total_steps;) {
        const int burst = total_steps - step < poll_every ? total_steps - step : poll_every;
        for (int i = 0; i < burst; ++i)
          for (int ci = 0; ci < used; ++ci) {
            if (!alive[ci] || rcs[ci] != MH_OK) continue;
            if (hipGraphLaunch(execs[ci], chain_stream[ci]) != hipSuccess) rcs[ci] = MH_ERR_LAUNCH;
          }
        step += burst;
        bool any = false;
        for (int ci = 0; ci < used; ++ci) {
          if (!alive[ci] || rcs[ci] != MH_OK) continue;
          if (step < total_steps && !forced) {
            int running = 1;
            if (hipMemcpyAsync(&running, &states[ci]->n_running, 4, hipMemcpyDeviceToHost, chain_stream[ci]) != hipSuccess ||
                hipStreamSynchronize(chain_stream[ci]) != hipSuccess) { rcs[ci] = MH_ERR_LAUNCH; continue; }
            if (running == 0) alive[ci] = false;
          }
          any = any || alive[ci];
        }
        if (!any) break;
      }
    } else {
      std::vector<std::thread> th;
      const MhOptionSet* set = current_option_set();   // thread-local in the caller: re-installed in every launcher
      for (int ci = 1; ci < used; ++ci) th.emplace_back([&, ci, set] { OptionScope sc(set); rcs[ci] = run_chain(ci); });
      rcs[0] = run_chain(0);
      for (auto& t : th) t.join();
    }
    for (int ci = 0; ci < used; ++ci)
      if (rcs[ci] != MH_OK) { set_error("mh_t5_generate: graph launch / poll failed on chain %d: %s", ci, hipGetErrorString(hipGetLastError())); rc = rcs[ci]; }
  }
  // join the chains back into the caller's stream
  for (int ci = 0; ci < used; ++ci) {
    (void)hipEventRecord(g_pool.join[ci], chain_stream[ci]);
    (void)hipStreamWaitEvent(s, g_pool.join[ci], 0);
  }
  if (rc == MH_OK) {
    hipLaunchKernelGGL(dec_finalize_kernel, dim3(1), dim3(64), 0, s, all.finish_col, B, n_steps_out);
    rc = check_launch("dec_finalize_kernel");
  }
  (void)hipStreamSynchronize(s);   // the graph objects must outlive their launches
  for (int ci = 0; ci < kMaxChains; ++ci) {
    if (cached[ci]) { step_graph_release(cached[ci]); continue; }
    if (execs[ci]) (void)hipGraphExecDestroy(execs[ci]);
    if (graphs[ci]) (void)hipGraphDestroy(graphs[ci]);
  }
  return rc;
}

// ------------------------------------------------------------------------------------------------
// Teacher-forced decoder forward over whole sequences: the `Mapperatorinator.forward` seam (B2,
// modeling_mapperatorinator.py:174-228) with `encoder_outputs` given -- every position of `ids` goes through the
// decoder stack at once (the batched prefill path: MFMA GEMMs + flash attention, causal + key mask), then the
// final RMSNorm and lm_head.  logits fp32 [B, T, V].
extern "C" int64_t mh_t5_forward_workspace_bytes(const MhT5Config* c, int B, int T) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  if (!c || B <= 0 || T <= 0) return -1;
  const int64_t es = es_of(c->dtype);
  return mh::prefill_layout(c, B, T, nullptr, 0, nullptr) +
         2 * mh::align256((int64_t)c->n_dec_layers * B * c->n_heads * 64 * c->tgt_len * es);
}

extern "C" int mh_t5_decoder_forward(const MhT5Config* c, const MhT5Weights* w, const void* cross_kv, int B,
                                     const int32_t* ids, const uint8_t* mask, int T, float* logits, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_cfg(c, "mh_t5_decoder_forward"));
  MH_REQUIRE(w && cross_kv && ids && logits && workspace, "mh_t5_decoder_forward: null argument");
  MH_REQUIRE(B > 0 && T >= 1 && T <= c->tgt_len, "mh_t5_decoder_forward: T=%d not in [1, tgt_len=%d]", T, c->tgt_len);
  MH_REQUIRE(workspace_bytes >= mh_t5_forward_workspace_bytes(c, B, T), "mh_t5_decoder_forward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int es = es_of(c->dtype);
  const int64_t cache = align256((int64_t)c->n_dec_layers * B * c->n_heads * 64 * c->tgt_len * es);
  char* self_k = (char*)workspace;
  char* self_v = self_k + cache;
  PrefillBuf pb;
  prefill_layout(c, B, T, self_v + cache, workspace_bytes - 2 * cache, &pb);
  MH_TRY(prefill_prompt(c, w, cross_kv, B, B, ids, mask, T, T, self_k, self_v, pb, s));
  const int rows = B * T, d = c->d_model;
  MH_TRY(pre_norm(c, pb.h, w->dec_final_ln, w->dec_final_ln_b, pb.n, rows, c->dtype, s));
  MhGemm g{};
  g.A = pb.n; g.lda = d; g.W = w->lm_head; g.ldw = d; g.C = logits; g.ldc = c->vocab_out; g.M = rows; g.N = c->vocab_out;
  g.K = d; g.dtype = c->dtype; g.epilogue = MH_EPI_STORE_F32;
  return gemm(g, s);
}

// ------------------------------------------------------------------------------------------------
// In-situ timing of the dominant kernel.  `buf` (device, uint64 [n_chains][ring][n_dec_layers][2], pre-filled by the
// caller with (UINT64_MAX, 0) pairs) makes every cross-attention launch of the following mh_t5_generate calls record its
// earliest workgroup start and latest workgroup end in wall-clock ticks (hipDeviceAttributeWallClockRate kHz); slot =
// decode position % ring.  buf = NULL switches it off.  Costs two atomics per workgroup: bench.py uses an EXTRA decode
// pass for it, never the timed region.
extern "C" int mh_t5_decode_timing(void* buf, int ring) {
  MH_REQUIRE((buf == nullptr) == (ring <= 0), "mh_t5_decode_timing: buf and ring go together");
  mh::g_timing.buf = (unsigned long long*)buf;
  mh::g_timing.ring = ring;
  return MH_OK;
}

#ifdef MH_PHASE_STAMPS
// profiling build only: read (and optionally clear) the phase stamps; out = host uint64 [16][16][2]
extern "C" int mh_debug_phase_stamps(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(mh::dec::g_stamps), sizeof(mh::dec::g_stamps)) != hipSuccess)
    return mh::check_launch("stamps read");
  if (reset) {
    static unsigned long long zero[16 * 16 * 2] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mh::dec::g_stamps), zero, sizeof(zero)) != hipSuccess) return mh::check_launch("stamps reset");
  }
  return MH_OK;
}
#endif

extern "C" int mh_wall_clock_khz(void) {   // rate of the device wall clock the timing hook records (kHz); 0 if unknown
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return khz;
}

extern "C" int mh_t5_decode_chains(int B) { return B > 0 ? mh::pick_chains(B) : 0; }
extern "C" int mh_t5_decode_chains_cfg(const MhT5Config* c, int B) {   // ... under the engine's option set
  mh::OptionScope option_scope(c ? c->options : nullptr);
  return B > 0 ? mh::pick_chains(B) : 0;
}

// ------------------------------------------------------------------------------------------------
// Measurement hook for bench.py's roofline line: the dominant decode kernel (cross-attention over the
// encoder keys) launched `reps` times back to back between two HIP events ON THE GIVEN STREAM, cycling
// through the decoder layers exactly like a decode step does (so every launch streams a different layer's
// K/V: B*H*L*64*2 elements).  ms_out[0] = average milliseconds per launch (split kernel + merge kernel).
extern "C" int mh_t5_cross_attn_probe(const MhT5Config* c, const MhT5Weights* w, const void* cross_kv, int B, int reps,
                                      float* ms_out, void* workspace, int64_t workspace_bytes, void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_cfg(c, "mh_t5_cross_attn_probe"));
  MH_REQUIRE(cross_kv && ms_out && workspace && B > 0 && B <= 64 && reps > 0, "mh_t5_cross_attn_probe: bad argument");
  MH_REQUIRE(workspace_bytes >= mh_t5_decode_workspace_bytes(c, B), "mh_t5_cross_attn_probe: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int es = es_of(c->dtype), H = c->n_heads, inner = H * 64, L = c->src_len, d = c->d_model;
  Arena ar(workspace, workspace_bytes);
  float* h = (float*)ar.take((int64_t)B * d * 4);
  void* q = ar.take((int64_t)B * inner * es);
  void* attn = ar.take((int64_t)B * inner * es);
  if (hipMemsetAsync(q, 0, (size_t)B * inner * es, s) != hipSuccess) return check_launch("probe memset");
  if (hipMemsetAsync(h, 0, (size_t)B * d * 4, s) != hipSuccess) return check_launch("probe memset");
  // the kernel the decode step launches: with weights given and the fused projections on, the cross-attention that
  // also projects its query (dec_cross_attn_q_kernel); otherwise the stand-alone dec_cross_attn_kernel
  const bool with_q = w != nullptr && fused_proj_enabled(d);
  const long kv_layer = (long)B * H * L * 64 * es;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return check_launch("event create");
  int rc = MH_OK;
  for (int pass = 0; pass < 2 && rc == MH_OK; ++pass) {   // pass 0 = warm-up
    if (pass == 1) (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps && rc == MH_OK; ++r) {
      const int l = r % c->n_dec_layers;
      dec::CrossAttnP ca{};
      ca.q = q; ca.ldq = inner; ca.k = (const char*)cross_kv + (long)(l * 2 + 0) * kv_layer;
      ca.v = (const char*)cross_kv + (long)(l * 2 + 1) * kv_layer; ca.out = attn; ca.ldo = inner;
      ca.B = B; ca.H = H; ca.L = L;
      if (with_q) {
        dec::HeadProjP hp{};
        hp.h = h; hp.ldh = d; hp.ln_w = w->dec_ln2[l]; hp.eps = c->eps; hp.W = w->dec_cq[l]; hp.ldw = d; hp.d = d;
        rc = c->dtype == MH_BF16 ? launch_cross_q_d<bf16_t>(ca, hp, s) : launch_cross_q_d<float>(ca, hp, s);
      } else {
        rc = c->dtype == MH_BF16 ? launch_cross<bf16_t>(ca, s) : launch_cross<float>(ca, s);
      }
    }
    if (pass == 1) (void)hipEventRecord(e1, s);
  }
  if (rc == MH_OK) {
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
      rc = check_launch("probe events");
    else
      ms_out[0] = ms / (float)reps;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}


// ------------------------------------------------------------------------------------------------
// Step-wise decode for host-driven search (beam search: HF `GenerationMixin._beam_search` drives the model one position at a
// time and reorders its cache, osuT5/osuT5/inference/cache_utils.py:16-20).  The workspace is the one of mh_t5_generate
// (mh_t5_decode_workspace_bytes) and holds the self-attention K/V caches between calls.
extern "C" int mh_t5_step(const MhT5Config* c, const MhT5Weights* w, const void* cross_kv, int B, int kv_group, const int32_t* ids,
                          int pos, const uint8_t* prompt_mask, int P, float* logits, void* workspace, int64_t workspace_bytes,
                          void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_cfg(c, "mh_t5_step"));
  MH_REQUIRE(w && cross_kv && ids && logits && workspace, "mh_t5_step: null argument");
  MH_REQUIRE(B > 0 && B <= 64, "mh_t5_step: batch %d not in [1, 64]", B);
  MH_REQUIRE(kv_group >= 1 && B % kv_group == 0, "mh_t5_step: %d rows are not whole groups of %d", B, kv_group);
  MH_REQUIRE(pos >= 0 && pos < c->tgt_len, "mh_t5_step: position %d outside the cache (tgt_len %d)", pos, c->tgt_len);
  MH_REQUIRE(workspace_bytes >= mh_t5_decode_workspace_bytes(c, B), "mh_t5_step: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int es = es_of(c->dtype), H = c->n_heads, inner = H * 64, d = c->d_model, V = c->vocab_out;
  Arena ar(workspace, workspace_bytes);
  DecBuffers bf{};
  bf.h = (float*)ar.take((int64_t)B * d * 4);
  bf.q = ar.take((int64_t)B * inner * es);
  bf.attn = ar.take((int64_t)B * inner * es);
  bf.ff = ar.take((int64_t)B * c->d_ff * es);
  float* ws_logits = (float*)ar.take((int64_t)B * V * 4);
  (void)ws_logits;
  bf.logits = logits;
  bf.self_k = ar.take((int64_t)c->n_dec_layers * B * inner * c->tgt_len * es);
  bf.self_v = ar.take((int64_t)c->n_dec_layers * B * inner * c->tgt_len * es);
  bf.finished = (uint8_t*)ar.take(B);
  bf.finish_col = (int32_t*)ar.take((int64_t)B * 4);
  bf.last_ts = (int32_t*)ar.take((int64_t)B * 4);
  bf.st = (DecState*)ar.take((int64_t)align256(sizeof(DecState)) * kMaxChains);
  bf.chain = 0;
  MH_REQUIRE(ar.ok(), "mh_t5_step: arena overflow");
  MH_REQUIRE(c->arch != 2 || (w->dec_pos && w->dec_final_ln_b), "mh_t5_step: arch 2 needs decoder.embed_positions and the LayerNorm biases");
  const float* dpos = c->arch == 2 ? w->dec_pos : nullptr;
  const uint8_t* pmask = (c->arch == 2 && c->dec_pos_from_mask) ? prompt_mask : nullptr;
  if (c->dtype == MH_BF16) hipLaunchKernelGGL(step_embed_kernel<bf16_t>, dim3(B), dim3(256), 0, s, ids, (const bf16_t*)w->dec_embed, d, bf.h, bf.st, pos, dpos, pmask, P);
  else hipLaunchKernelGGL(step_embed_kernel<float>, dim3(B), dim3(256), 0, s, ids, (const float*)w->dec_embed, d, bf.h, bf.st, pos, dpos, pmask, P);
  MH_TRY(check_launch("step_embed_kernel"));
  SampleP smp{};
  const int kvB = B / kv_group;
  return c->dtype == MH_BF16 ? enqueue_step<bf16_t>(c, w, cross_kv, B, B, kvB, prompt_mask, P, bf, smp, s, nullptr, nullptr, false, kv_group)
                             : enqueue_step<float>(c, w, cross_kv, B, B, kvB, prompt_mask, P, bf, smp, s, nullptr, nullptr, false, kv_group);
}

// self-attention cache rows of every layer: row b <- row src[b] for positions 0 .. n_pos-1 (`cache.reorder_cache(beam_idx)`).
// scratch: 2 * n_dec * B * inner * n_pos elements of the storage type (mh_t5_reorder_cache_scratch_bytes).
extern "C" int64_t mh_t5_reorder_cache_scratch_bytes(const MhT5Config* c, int B, int n_pos) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  if (!c || B <= 0 || n_pos <= 0) return -1;
  return align256(2LL * c->n_dec_layers * B * c->n_heads * 64 * n_pos * es_of(c->dtype));
}
extern "C" int mh_t5_reorder_cache(const MhT5Config* c, int B, const int32_t* src, int n_pos, void* workspace, int64_t workspace_bytes,
                                   void* scratch, int64_t scratch_bytes, void* stream) {
  mh::OptionScope option_scope(c ? c->options : nullptr);
  MH_TRY(check_cfg(c, "mh_t5_reorder_cache"));
  MH_REQUIRE(src && workspace && scratch && B > 0 && B <= 64 && n_pos > 0 && n_pos <= c->tgt_len, "mh_t5_reorder_cache: bad argument");
  MH_REQUIRE(workspace_bytes >= mh_t5_decode_workspace_bytes(c, B) && scratch_bytes >= mh_t5_reorder_cache_scratch_bytes(c, B, n_pos),
             "mh_t5_reorder_cache: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int es = es_of(c->dtype), H = c->n_heads, inner = H * 64, d = c->d_model, V = c->vocab_out;
  Arena ar(workspace, workspace_bytes);
  ar.take((int64_t)B * d * 4); ar.take((int64_t)B * inner * es); ar.take((int64_t)B * inner * es); ar.take((int64_t)B * c->d_ff * es);
  ar.take((int64_t)B * V * 4);
  void* self_k = ar.take((int64_t)c->n_dec_layers * B * inner * c->tgt_len * es);
  void* self_v = ar.take((int64_t)c->n_dec_layers * B * inner * c->tgt_len * es);
  const long layer_stride = (long)B * H * c->tgt_len * 64;
  const dim3 grid(B * H, c->n_dec_layers, 2);
  if (c->dtype == MH_BF16) {
    hipLaunchKernelGGL(cache_gather_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)self_k, (const bf16_t*)self_v, (bf16_t*)scratch, src, B, H, c->tgt_len, n_pos, layer_stride);
    hipLaunchKernelGGL(cache_scatter_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)self_k, (bf16_t*)self_v, (const bf16_t*)scratch, B, H, c->tgt_len, n_pos, layer_stride);
  } else {
    hipLaunchKernelGGL(cache_gather_kernel<float>, grid, dim3(256), 0, s, (const float*)self_k, (const float*)self_v, (float*)scratch, src, B, H, c->tgt_len, n_pos, layer_stride);
    hipLaunchKernelGGL(cache_scatter_kernel<float>, grid, dim3(256), 0, s, (float*)self_k, (float*)self_v, (const float*)scratch, B, H, c->tgt_len, n_pos, layer_stride);
  }
  return check_launch("cache reorder");
}
