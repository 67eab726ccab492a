/*
 * libmapperhip -- C ABI of the MI355X-native (gfx950) audio->event hot path of Mapperatorinator.
 *
 * Every entry point below replaces a *Python call site into third-party device code* of the
 * reference (the reference owns no native code); the file:line after each
 * declaration is the reference interface it stands in for (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - plain C: raw DEVICE pointers + sizes, no torch / C++ types;
 *   - asynchronous on the given `hipStream_t` (passed as void*), no hidden allocation, no host sync
 *     (the one exception, mh_t5_generate, documents its polling); caller owns every buffer;
 *   - returns MH_OK (0) or a negative MhStatus; the message is available from mh_last_error();
 *     never throws, never aborts;
 *   - `dtype` is MH_F32 or MH_BF16: the storage type of weights and GEMM operands.  Accumulation,
 *     normalisation, softmax, the residual stream and logits are always fp32.
 */
#ifndef MAPPERHIP_H_
#define MAPPERHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_ABI_VERSION 10
#define MH_MAX_LAYERS 32

typedef enum MhStatus {
  MH_OK = 0,
  MH_ERR_ARG = -1,     /* bad argument (shape / alignment / null) */
  MH_ERR_LAUNCH = -2,  /* HIP launch or runtime error */
  MH_ERR_STATE = -3,   /* object used in the wrong state */
} MhStatus;

/* MH_MX8 (ABI 7): OCP MX-fp8 -- e4m3 elements (one byte each) + one E8M0 scale byte per (row, 32 consecutive k); an OPERAND
 * type of mh_gemm and of the reduced-precision modes built on it (MhT5Config.enc_operand_dtype, MhDiTConfig.operand_dtype),
 * never a storage type of activations that leave a stage.  Layouts: see mh_quantize_mx8. */
typedef enum MhDtype { MH_F32 = 0, MH_BF16 = 1, MH_MX8 = 2 } MhDtype;

/* epilogues of mh_gemm */
typedef enum MhEpilogue {
  MH_EPI_STORE = 0,      /* C[T]   = A W^T (+bias)                                  */
  MH_EPI_STORE_F32 = 1,  /* C[f32] = A W^T (+bias)                                  */
  MH_EPI_RESID = 2,      /* C[f32] += A W^T (+bias)                                 */
  MH_EPI_GEGLU = 3,      /* C[T][M,N/2] = gelu_tanh(acc[:,blk even]) * acc[:,blk odd];
                            W rows interleaved in blocks of 16 (wi_0 blk, wi_1 blk, ...) */
  MH_EPI_BIAS_GELU = 4,  /* C[T]   = gelu_tanh(A W^T + bias)                        */
  MH_EPI_GATE_RESID = 5, /* C[f32][m,n] += gate[m / rows_per_batch, n] * (A W^T + bias)[m,n] */
  MH_EPI_KV_SCATTER = 6, /* C[T] scattered to [(layer,kv)][b][h][key][64]; see mh_t5_cross_kv */
  MH_EPI_QKV_VT = 7,     /* cols < n_split: C[T] = A W^T (+bias) (q|k block, ldc);
                            cols >= n_split (the v block, col' = h*64+dd, row = b*kv_L+key):
                            C2[T][((b*kv_H + h)*64 + dd)*kv_Lpad + key]  (V transposed per head)  */
  MH_EPI_QKV_CACHE = 8,  /* decoder prompt prefill: rows m = b*kv_L + i (prompt position i of chunk b);
                            cols < n_split: C[T] (q, ldc); then the k block -> C2 and the v block -> C3, both
                            [B][H][cache_len][64] at position i; the v block is also written transposed to
                            C4[T][((b*kv_H + h)*64 + dd)*kv_Lpad + i]                                     */
  MH_EPI_BIAS_GELU_ERF = 9, /* C[T] = gelu_erf(A W^T + bias) (+ gate[m % rows_per_batch, n] when gate != NULL:
                               the Whisper position table)                                                 */
} MhEpilogue;

const char* mh_last_error(void);
int mh_abi_version(void);

/* Tuning options (process-wide; per engine through an option set, below): each has a default, an environment override read at first use and this run-time
 * setter.  They choose between kernels that compute the same results (bit-identical unless noted):
 *   "gemm_splitk_tiles"  MH_GEMM_SPLITK_TILES  192  fp32/bf16 GEMMs with fewer 32x32 tiles than this use the 16x16
 *                                                   split-K tile (different fp32 summation order); 0 = never
 *   "decode_chains"      MH_DECODE_CHAINS      0    independent row chains of a decode step (0 = automatic)
 *   "decode_prefill"     MH_DECODE_PREFILL     1    batched prompt prefill (0: token by token)
 *   "decode_gemv_cols"   MH_DECODE_GEMV_COLS   0    valid columns per 16-column tile of the decode GEMVs (0 = automatic)
 *   "decode_fused_proj"  MH_DECODE_FUSED_PROJ  1    decode attention kernels project their own q / k / v (0: stand-alone
 *                                                   GEMV launches; fp32 summation order of the projections differs;
 *                                                   2: stand-alone QKV GEMV, cross-attention keeps its own projection)
 *   "gemm_glds"          MH_GEMM_GLDS          3    bf16 GEMM operands by LDS-DMA: 3 = three-stage kernel (256x128 tiles, 128x128
 *                                                   below half a wave of them), 2 = 256x128 only, 1 = two-stage 128x128,
 *                                                   0 = register staging
 *   "decode_launch_threads" MH_DECODE_LAUNCH_THREADS 1  one host launcher thread per decode chain; 0 = one thread feeds all chains
 *                                                   round robin (for profilers whose counter passes do not survive concurrent
 *                                                   launcher threads)
 *   "decode_graph_cache" MH_DECODE_GRAPH_CACHE 1    step graphs kept across mh_t5_generate calls (0: captured per call)
 *   "gemm_tile256sq_min" MH_GEMM_TILE256SQ_MIN 440  bf16 GEMM: the 256x256 tile (two LDS stages, 128x64 wave tiles, staggered wave
 *                                                   groups) from this many tiles on when its rounds of 256 workgroups are >= 88 %
 *                                                   full (0 = never; bit-identical to the 256x128 three-stage kernel)
 *   "gemm_2stage_max_k"  MH_GEMM_2STAGE_MAX_K  512  bf16 GEMM with K <= this: 128x128 tile on two LDS stages, two workgroups per CU
 *                                                   (0 = never; bit-identical)
 * (gemm_tile128_min and dit_split3_min_rows are documented next to their definitions in csrc/api.hip: 12 options in all.
 * Round 5 removed the measured-slower variants decode_overlap, decode_fold_oproj, decode_cu_split, decode_self_rows,
 * mx8_waves = 4, dit_s3_fused_ln, the debugging aid gemm_lds_pad, the variant switches attn_flash2, dit_s3_presplit and
 * mx8_fused_quant (the faster form is the only one left) and turned the thresholds attn_small_max_wgs, gemm_tile256_min,
 * mx8_tile256_min into constants; their measurements stay in profiles/r02_* .. r04_*.)
 * Unknown names return MH_ERR_ARG (set) / -1 (get). */
int mh_set_option(const char* name, long value);
long mh_get_option(const char* name);
/* ABI 8 -- option sets: overrides owned by ONE engine instead of the process.  A set starts empty (every option falls through to
 * the process-wide value above); MhT5Config.options / MhDiTConfig.options point at it (NULL = process-wide values only) and every
 * entry point that takes such a config resolves its options through the set for the duration of the call, the decode chains'
 * launcher threads included.  Two engines in one process can so run different kernel variants side by side.  The set must
 * outlive the calls that name it; setting an option while a call that uses the set is in flight is a data race the caller
 * excludes.  mh_options_get returns the override, else the process-wide value; unknown names: MH_ERR_ARG / -1. */
typedef struct MhOptionSet MhOptionSet;
MhOptionSet* mh_options_create(void);
void mh_options_destroy(MhOptionSet* set);
int mh_options_set(MhOptionSet* set, const char* name, long value);
int mh_options_clear(MhOptionSet* set, const char* name);      /* drop one override (name = NULL: all of them) */
long mh_options_get(const MhOptionSet* set, const char* name);
/* sizeof() of the ABI structs as this library was compiled, so that a binding can verify its own layout:
 * which = 0 MhGemm, 1 MhT5Config, 2 MhT5Weights, 3 MhSampling, 4 MhDiTConfig, 5 MhDiTWeights,
 * 6 MhSliderSet; -1 otherwise. */
int mh_struct_size(int which);

/* ------------------------------------------------------------------------------------------------
 * K1  mel frontend.  Replaces `nnAudio.features.MelSpectrogram` as constructed and called at
 *     osuT5/osuT5/model/spectrogram.py:50-61 and :63-83 (zero-pad n_fft/2, hann STFT, power
 *     spectrum, Slaney mel filterbank, optional log1p, (B, frames, n_mels) layout).
 * audio   [B, n_samples] fp32.
 * out     [B, n_frames, ld_out] (fp32 if out_dtype==MH_F32 else bf16), n_frames = n_samples/hop + 1;
 *         columns [n_mels, ld_out) are written as zero (K padding for the following GEMM).
 * The sparse filterbank (CSR over mel rows) and the twiddle table are built on the host
 * (mapperatorinator_amd/mel.py) and passed in:
 *   fb_start[n_mels], fb_len[n_mels], fb_off[n_mels] (int32), fb_w[sum(len)] fp32,
 *   window[n_fft] fp32 (periodic hann), twiddle[n_fft] (cos, sin) pairs fp32.
 * log_scale: bit 0 = log1p of the mel energies (spectrogram.py:79-80); bit 1 = reflect instead of zero padding of the
 *   n_fft/2 samples either side (the `torchaudio` parameterisation of the Whisper-family configs:
 *   configs/model/whisper_base_v3.yaml:16-21 -- torch.stft(center=True, pad_mode="reflect"), HTK filterbank without
 *   area normalisation; the filterbank is whatever the host passes in).
 * n_fft must be 1024 (the only size the reference configs use: configs/model/default.yaml:29-37). */
int mh_mel(const float* audio, int B, int n_samples, int n_fft, int hop, int n_mels,
           const float* window, const float* twiddle, const int32_t* fb_start, const int32_t* fb_len,
           const int32_t* fb_off, const float* fb_w, int log_scale, void* out, int ld_out,
           int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense layer.  Replaces `nn.Linear` / `torch.matmul` call sites on the hot path
 * (modeling_mapperatorinator.py:195-196; HF T5Attention q/k/v/o, T5DenseGatedActDense;
 *  osu_diffusion/utils/models.py:111-116,145-155).  C = A[M,K] * W[N,K]^T with a fused epilogue.
 * A, W have element type `dtype`; lda/ldw/ldc in elements; K % 8 == 0 (bf16) or % 4 (f32),
 * row starts 16-byte aligned.  bias fp32[N] or NULL.  gate fp32 [.., gate_ld].
 * For MH_EPI_KV_SCATTER: kv_B, kv_H, kv_L describe the scatter (rows m = b*kv_L + key). */
typedef struct MhGemm {
  const void* A; int lda;
  const void* W; int ldw;
  void* C; int ldc;
  int M, N, K;
  const float* bias;
  const float* gate; int gate_ld; int rows_per_batch;
  int kv_B, kv_H, kv_L;
  void* C2; int n_split; int kv_Lpad;   /* MH_EPI_QKV_VT / MH_EPI_QKV_CACHE */
  void* C3; void* C4; int cache_len;    /* MH_EPI_QKV_CACHE only */
  int dtype; int epilogue;
  /* fp32 only -- adaLN `modulate(LayerNorm(x), shift, scale)` (osu_diffusion/utils/models.py:11-12,145,152) fused
   * around the GEMMs of a DiT block instead of a stand-alone pass over the activations:
   *   stats_out  (MH_EPI_STORE_F32 / MH_EPI_GATE_RESID producers) fp32 [ceil(N/16)][M][2]: per 16-column strip the
   *              sum and sum of squares of every output row, written by the epilogue;
   *   ln_stats   (consumer) the same array for the rows of A (ln_strips strips, K columns): the A operand becomes
   *              (a - mean) * rsqrt(var + ln_eps) * (1 + ln_scale[b][k]) + ln_shift[b][k], b = row / rows_per_batch,
   *              ln_shift / ln_scale fp32 with row stride ln_ld. */
  float* stats_out;
  const float* ln_stats; int ln_strips;
  const float* ln_shift; const float* ln_scale; int ln_ld; float ln_eps;
  /* fp32 only -- "bf16 x 3": W is stored PRE-SPLIT, every 32-float block of a row as [32 x bf16 hi | 32 x bf16 lo]
   * (hi = bf16(w), lo = bf16(w - hi); the same 128 bytes per block, ldw still counted in floats), A stays fp32 and is
   * split on its way into LDS; a w ~= a_hi w_hi + a_hi w_lo + a_lo w_hi on the bf16 matrix cores with fp32 accumulation
   * (relative error ~2^-16 per product instead of exact fp32 products).  K and ldw multiples of 32.
   * Bit flags: 1 = W pre-split (above); | 2 = A is stored pre-split in the same layout by its producer (lda counted in
   * floats, multiple of 32): the three-stage LDS-DMA form, no conversion work inside the GEMM (STORE_F32, QKV_VT,
   * GATE_RESID, BIAS_GELU; N, ldc multiples of 4); | 4 = with 2 and MH_EPI_BIAS_GELU: C is written pre-split as well
   * (it is the next GEMM's A operand; ldc multiple of 32). */
  int w_split3;
  /* ABI 7 -- dtype = MH_MX8 (BASELINE configs[4] "fp8 MFMA"): A [M][lda] and W [N][ldw] hold e4m3 BYTES (lda, ldw counted in
   * elements = bytes, multiples of 16; K a multiple of 128), a_scale / w_scale their E8M0 scales in the row layout of
   * mh_quantize_mx8 (mh_mx8_scale_row_bytes(K) bytes per row).  The product runs on v_mfma_scale_f32_16x16x128_f8f6f4 with
   * fp32 accumulation; outputs are what the epilogue says with T = bf16 (STORE, STORE_F32, RESID, GEGLU, BIAS_GELU,
   * GATE_RESID, KV_SCATTER, QKV_VT; N, ldc multiples of 4). */
  const uint8_t* a_scale; const uint8_t* w_scale;
  /* ABI 9 -- dtype = MH_MX8 with MH_EPI_GEGLU / MH_EPI_BIAS_GELU: when mx_out != NULL the epilogue writes the MX-fp8 image of its
   * bf16 result (the next GEMM's A operand) instead of the bf16 matrix: e4m3 bytes [M][ldc] (ldc = bytes per row) at mx_out and
   * E8M0 scales at mx_out_scales (mh_mx8_scale_row_bytes(width) bytes per row; width = N, or N / 2 for GEGLU, a multiple of 128).
   * Bit for bit what mh_quantize_mx8 makes of the bf16 result; C is not written and may be NULL. */
  uint8_t* mx_out; uint8_t* mx_out_scales;
} MhGemm;
int mh_gemm(const MhGemm* g, void* stream);

/* MX-fp8 quantisation of a row-major matrix (the A / W operands of mh_gemm with dtype = MH_MX8).
 * x [rows][ldx] of in_dtype (MH_F32 / MH_BF16), K % 128 == 0 columns used.  Per (row, block of 32 consecutive k):
 *   amax = max |x|; e = floor(log2(amax)) - 8, raised by one when amax * 2^-e > 448 (the largest finite e4m3 value: nothing is
 *   ever clipped); scale byte = clamp(e + 127, 0, 254) (all-zero block: 0 with zero elements); element = RNE_e4m3(x * 2^-e).
 * q [rows][ldq] bytes (ldq >= K, multiple of 16).  scales [rows][mh_mx8_scale_row_bytes(K)] bytes, LANE-MAJOR in groups of
 * four 128-k steps: byte (kt / 4) * 16 + lg * 4 + (kt % 4) holds the scale of k block kt * 4 + lg (kt = k / 128, lg =
 * (k % 128) / 32) -- the dword a lane of the MFMA fetches for four K steps; unused bytes of the last group are 0. */
int64_t mh_mx8_scale_row_bytes(int K);
int mh_quantize_mx8(const void* x, int ldx, int rows, int K, int in_dtype, uint8_t* q, int ldq, uint8_t* scales, void* stream);
/* RMSNorm (mh_rmsnorm) with the MX-fp8 operand written directly: y = w * x * rsqrt(mean(x^2) + eps) in fp32, rounded to
 * `round_dtype` (MH_BF16: the value a bf16 activation buffer would have held; MH_F32: not rounded), then quantised as above. */
int mh_rmsnorm_mx8(const float* x, int ldx, const float* w, int rows, int d, float eps, int round_dtype, uint8_t* q, int ldq,
                   uint8_t* scales, void* stream);

/* T5 RMSNorm (HF T5LayerNorm; restated at custom_transformers/t5.py:50-62):
 * y[T] = w * x * rsqrt(mean(x^2) + eps), x fp32 [rows, d] (ldx), y [rows, ldy]. */
int mh_rmsnorm(const float* x, int ldx, const float* w, void* y, int ldy, int rows, int d, float eps,
               int out_dtype, void* stream);
/* (ABI 10) nn.LayerNorm with affine parameters -- the pre-norm of stock HF Whisper's blocks ('openai/whisper-*', the V28 / V29
 * backbones; transformers models/whisper/modeling_whisper.py WhisperEncoderLayer / WhisperDecoderLayer): y = (x - mean) *
 * rsqrt(var + eps) * w + b, biased variance of the centred values, fp32 arithmetic; x fp32 [rows, ldx], y [rows, ldy] of
 * `out_dtype`. */
int mh_layernorm(const float* x, int ldx, const float* w, const float* b, void* y, int ldy, int rows, int d, float eps,
                 int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3  T5 encoder self-attention (HF T5Attention.forward, restated custom_transformers/t5.py:170-250):
 *     softmax(scale * Q K^T + bias[h][k - q + L - 1]) V; T5: scale = 1 (no 1/sqrt(d)), bidirectional.
 * qk   [B*L, ld_qk] element type `dtype`: q of head h at columns [h*64, h*64+64), k at
 *      [k_col0 + h*64, ...);   vt [B][H][64][Lpad] = V transposed per head (MH_EPI_QKV_VT output,
 *      pad columns zero);  bias fp32 [H][2L-1] (host-built bucket lookup by k - q) or NULL;
 * out  [B*L, ld_out] element type `dtype`, head h at columns [h*64, h*64+64).
 * K7  also the DiT attention (osu_diffusion/utils/models.py:145-151): scale = 1/8, bias NULL and
 *     band > 0: query q attends key k iff -(band-1) <= k - q <= band, which is exactly the banded
 *     bool mask built at diffusion_pipeline.py:146-148 (band = seq_len = 128).  band <= 0: no mask. */
int mh_attention(const void* qk, int ld_qk, int k_col0, const void* vt, int Lpad, const float* bias,
                 void* out, int ld_out, int B, int L, int H, float scale, int band, int dtype,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2  Whisper-style audio front-end (HF WhisperEncoder.forward prologue; reference forks
 *     osuT5/osuT5/model/custom_transformers/modeling_varwhisper.py:779-780,813-816):
 *     y = gelu(conv1d(x, k=3, pad=1)); z = gelu(conv1d(y, k=3, stride=2, pad=1)); out = z (+ pos).
 * x   [B, Lin, C]   time-major activations (element type `dtype`) -- the transpose of HF's input_features
 * w1  [d, Kpad1], w2 [d, Kpad2]: conv weights repacked tap-major, W'[o][k*Cin + c] = W[o][c][k], K padded
 *     with zeros to a multiple of 32;  b1, b2 fp32 [d];  pos fp32 [Lout, d] or NULL (fixed sinusoid table)
 * out [B, Lout, d], Lout = (Lin - 1) / 2 + 1. */
int64_t mh_whisper_frontend_workspace_bytes(int B, int Lin, int C, int d, int dtype);
int mh_whisper_frontend(const void* x, int B, int Lin, int C, const void* w1, const float* b1, const void* w2,
                        const float* b2, const float* pos, int d, void* out, void* workspace,
                        int64_t workspace_bytes, int dtype, void* stream);

/* (ABI 10) The wrapper's conditioning vectors as INPUT CHANNELS of the front-end: with project_encoder_input = false
 * (configs/model/whisper_small_v2.yaml: the V30 / V31 'Tiger14n/ropewhisper-small' releases, cond_size 384) the per-row
 * difficulty / mapper / song-position vectors are repeated over the frames and concatenated to the mel channels
 * (osuT5/osuT5/model/modeling_mapperatorinator.py:183-202), so conv1 sees n_mels + cond_size channels -- a k = 3 convolution with
 * zero padding, where the constant channels do NOT reduce to a row bias (the first and last frame miss one tap).
 * frames [B * L, ld] (element type `dtype`, the K-padded buffer mh_mel wrote); cond fp32 [B, n_cond] (already rounded to the
 * storage type by the caller); writes frames[(b, t)][col0 + j] = cond[b][j]. */
int mh_cond_channels(void* frames, int B, int L, int ld, int col0, const float* cond, int n_cond, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * T5 model object: weights are caller-owned device buffers (packed by the host, see
 * mapperatorinator_amd/t5_engine.py) described by MhT5Weights.  All matrices are [N][Kpad] row major,
 * element type cfg.dtype, K padded with zeros to a multiple of 32.
 */
typedef struct MhT5Config {
  int d_model, d_kv, d_ff, n_heads, n_enc_layers, n_dec_layers;
  int vocab_in, vocab_out;
  int n_mels, n_mels_pad;   /* encoder_embedder K and its padded leading dimension */
  int src_len;              /* L: mel frames per chunk (encoder positions)             */
  int tgt_len;              /* max_target_positions = StaticCache length                */
  int dtype;                /* MhDtype                                                  */
  float eps;                /* RMSNorm epsilon (T5: 1e-6; Whisper family: nn.RMSNorm(eps=None) = finfo(dtype).eps) */
  /* --- ABI 5: the Whisper-family backbone of the released V30-V32 checkpoints ('OliBomby/varwhisper-*',
   * osuT5/osuT5/model/custom_transformers/modeling_varwhisper.py) behind the same entry points ------------------------ */
  int arch;                 /* 0 = T5-v1.1 (relative bias, unscaled scores, gated-GELU FFN); 1 = VarWhisper: conv front-end,
                               pre-norm RMSNorm blocks, fused Wqkv / Wo and fc1 -> gelu(erf) -> fc2 with optional biases,
                               rotate-half RoPE on q / k of the self-attentions, scores * attn_scale, untied head        */
  float attn_scale;         /* arch 1: 1 / sqrt(64)                                                                      */
  int in_frames;            /* arch 1: log-mel frames per chunk entering the front-end (`mel` rows per chunk); src_len is
                               then the number of ENCODER positions = (in_frames - 1) / 2 + 1 (conv2 stride 2), which is
                               what the cross-attention streams.  arch 0: unused (= src_len)                             */
  int local_every;          /* arch 1: global_attn_every_n_layers -- layer l is LOCAL iff l % local_every != 0; <= 1: none */
  int local_window;         /* arch 1: local_attention // 2 keys either side for local layers (the reference applies the
                               window on its flash-attention path only, modeling_varwhisper.py:330)                      */
  /* --- ABI 7: BASELINE configs[4] "fp8 MFMA" ----------------------------------------------------------------------- */
  int enc_operand_dtype;    /* 0 (= the storage type) or MH_MX8: the encoder blocks' four projections and the cross-K/V
                               projection take MX-fp8 operands (weights pre-quantised by the host: MhT5Weights.*_mx;
                               activations quantised where they are produced: RMSNorm output directly, attention output and
                               gated-GELU hidden by a pass over the bf16 buffer).  Needs dtype = MH_BF16, arch 0, d_model /
                               d_ff multiples of 128.  A reduced-precision mode of its own (the reference has none): gates
                               are error bounds against the fp32 reference goldens, never bit-exactness.                 */
  const MhOptionSet* options;   /* ABI 8: this engine's option overrides (mh_options_create), NULL = the process-wide values */
  /* --- ABI 10: arch 2 = stock HF Whisper ('openai/whisper-*': the V28 / V29 releases, configs/model/whisper_{base,small}.yaml;
   * transformers models/whisper/modeling_whisper.py): the wrapper's encoder_embedder (n_mels [+ cond] -> d, as arch 0) in front
   * of the conv front-end (C_in = d), + encoder.embed_positions; pre-norm blocks with AFFINE nn.LayerNorm (eps 1e-5; MhT5Weights
   * *_ln*_b), q / v / out projections with bias and k without (packed as arch 1: fused Wqkv / Wq + Wkv with a zero k-bias),
   * q scaled by 1 / 8, NO rotary embedding, fc1 -> gelu(erf) -> fc2, decoder input = decoder_embedder[id] +
   * decoder.embed_positions[position], final LayerNorm, proj_out.  in_frames / src_len as arch 1.                           */
  int dec_pos_from_mask;    /* arch 2: 0 = the position of column t is t (transformers 5.x: cache positions); 1 = t minus the
                               number of masked (padding) prompt columns of its row, clamped at 0 (transformers 4.57's
                               Whisper `prepare_inputs_for_generation`: decoder_position_ids = cumsum(mask) - 1; needs a
                               left-padded prompt mask)                                                                  */
} MhT5Config;

typedef struct MhT5Weights {
  const void* enc_embed_w;            /* [d, n_mels_pad]           modeling_mapperatorinator.py:123-124 */
  const float* enc_embed_b;           /* [d]                                                           */
  const void* dec_embed;              /* [vocab_in, d]             modeling_mapperatorinator.py:126-128 */
  const float* enc_rel_bias;          /* fp32 [H][2L-1]   bucketed lookup, encoder (bidirectional)      */
  const float* dec_rel_bias;          /* fp32 [H][tgt_len] lookup by distance q-k >= 0 (causal)         */
  /* encoder blocks */
  const float* enc_ln1[MH_MAX_LAYERS];   /* [d]  layer.0.layer_norm                                   */
  const void* enc_qkv[MH_MAX_LAYERS];    /* [3*inner, d]  q|k|v stacked                               */
  const void* enc_o[MH_MAX_LAYERS];      /* [d, inner]                                                */
  const float* enc_ln2[MH_MAX_LAYERS];   /* [d]  layer.1.layer_norm                                   */
  const void* enc_wi[MH_MAX_LAYERS];     /* [2*d_ff, d]  wi_0 / wi_1 interleaved in 16-row blocks     */
  const void* enc_wo[MH_MAX_LAYERS];     /* [d, d_ff]                                                 */
  const float* enc_final_ln;             /* [d]                                                       */
  /* decoder blocks */
  const float* dec_ln1[MH_MAX_LAYERS];
  const void* dec_qkv[MH_MAX_LAYERS];    /* [3*inner, d] self-attention                               */
  const void* dec_o[MH_MAX_LAYERS];
  const float* dec_ln2[MH_MAX_LAYERS];
  const void* dec_cq[MH_MAX_LAYERS];     /* [inner, d]  cross-attention query                         */
  const void* dec_ckv_all;               /* [n_dec*2*inner, d] rows ordered (layer, k|v, head, 64)    */
  const void* dec_co[MH_MAX_LAYERS];     /* [d, inner]                                                */
  const float* dec_ln3[MH_MAX_LAYERS];
  const void* dec_wi[MH_MAX_LAYERS];
  const void* dec_wo[MH_MAX_LAYERS];
  const float* dec_final_ln;
  const void* lm_head;                   /* [vocab_out, d]                                            */
  /* --- ABI 5, arch 1 only (all NULL for T5).  Matrix slots above are reused: *_qkv = fused Wqkv [3 d, d], *_o = Wo,
   * dec_cq = cross Wq, dec_ckv_all = the layers' cross Wkv stacked [n_dec * 2 d, d] (rows k | v per layer, head-major:
   * `kv.view(bs, -1, 2, H, 64)`, :544), dec_co = cross Wo, *_wi = fc1 [d_ff, d] (NOT interleaved), *_wo = fc2 [d, d_ff],
   * dec_embed = the wrapper's decoder_embedder, lm_head = proj_out, enc_ln1 / enc_ln2 = self_attn_layer_norm /
   * final_layer_norm, dec_ln1 / dec_ln2 / dec_ln3 = self_attn_ / cross_attn_ / final_layer_norm. ------------------- */
  const void* conv1_w; const float* conv1_b;      /* front-end, mh_whisper_frontend layout ([d, Kpad] tap-major)      */
  const void* conv2_w; const float* conv2_b;
  const float* enc_qkv_b[MH_MAX_LAYERS]; const float* enc_o_b[MH_MAX_LAYERS];     /* fp32, NULL = attention_bias false */
  const float* enc_fc1_b[MH_MAX_LAYERS]; const float* enc_fc2_b[MH_MAX_LAYERS];
  const float* dec_qkv_b[MH_MAX_LAYERS]; const float* dec_o_b[MH_MAX_LAYERS];
  const float* dec_cq_b[MH_MAX_LAYERS]; const float* dec_ckv_b_all;               /* [n_dec * 2 d]                     */
  const float* dec_co_b[MH_MAX_LAYERS];
  const float* dec_fc1_b[MH_MAX_LAYERS]; const float* dec_fc2_b[MH_MAX_LAYERS];
  /* rotary tables, fp32 [positions][64] = cos(32) | sin(32), built on the host by VarWhisperRotaryEmbedding's formulas
   * (:212-226) and rounded to the storage type there: encoder positions 0 .. src_len-1, decoder 0 .. tgt_len-1;
   * *_local = the same with local_rope_theta for local layers (may alias the global ones)                           */
  const float* enc_rope; const float* enc_rope_local; const float* dec_rope; const float* dec_rope_local;
  /* --- ABI 7, enc_operand_dtype = MH_MX8 only: the MX-fp8 copies (mh_quantize_mx8 layout: e4m3 [N][K] + scales
   * [N][mh_mx8_scale_row_bytes(K)]) of enc_qkv / enc_o / enc_wi (interleaved like enc_wi) / enc_wo / dec_ckv_all --------- */
  const uint8_t* enc_qkv_mx[MH_MAX_LAYERS]; const uint8_t* enc_qkv_mxs[MH_MAX_LAYERS];
  const uint8_t* enc_o_mx[MH_MAX_LAYERS]; const uint8_t* enc_o_mxs[MH_MAX_LAYERS];
  const uint8_t* enc_wi_mx[MH_MAX_LAYERS]; const uint8_t* enc_wi_mxs[MH_MAX_LAYERS];
  const uint8_t* enc_wo_mx[MH_MAX_LAYERS]; const uint8_t* enc_wo_mxs[MH_MAX_LAYERS];
  const uint8_t* dec_ckv_all_mx; const uint8_t* dec_ckv_all_mxs;
  /* --- ABI 10, arch 2 only: LayerNorm biases (fp32 [d]; the *_ln* slots above hold the LayerNorm weights) and the absolute
   * position tables -- encoder.embed_positions fp32 [src_len][d] (the fixed sinusoids, read from the state dict),
   * decoder.embed_positions fp32 [tgt_len][d] (learned) ------------------------------------------------------------------ */
  const float* enc_ln1_b[MH_MAX_LAYERS]; const float* enc_ln2_b[MH_MAX_LAYERS]; const float* enc_final_ln_b;
  const float* dec_ln1_b[MH_MAX_LAYERS]; const float* dec_ln2_b[MH_MAX_LAYERS]; const float* dec_ln3_b[MH_MAX_LAYERS];
  const float* dec_final_ln_b;
  const float* enc_pos; const float* dec_pos;
} MhT5Weights;

/* bytes of scratch needed by mh_t5_encode for a batch of B chunks */
int64_t mh_t5_encode_workspace_bytes(const MhT5Config* cfg, int B);

/* K2/K3/K4  mel -> encoder_embedder -> T5 encoder stack (final RMSNorm included).
 * Replaces OsuTEncoder.forward (modeling_mapperatorinator.py:392-443) + HF T5Stack (encoder).
 * mel    [B*L, n_mels_pad] element type cfg.dtype (output of mh_mel with ld_out = n_mels_pad);
 * enc_out[B*L, d] element type cfg.dtype (final-norm output, the cross-attention K/V GEMM operand);
 * enc_out_f32 optional fp32 copy [B*L, d] (NULL to skip) for parity checks. */
int mh_t5_encode(const MhT5Config* cfg, const MhT5Weights* w, const void* mel, int B, void* enc_out,
                 float* enc_out_f32, void* workspace, int64_t workspace_bytes, void* stream);
/* arch 1 (VarWhisperEncoder.forward, modeling_varwhisper.py:779-852): `mel` is [B * in_frames, n_mels_pad] log-mel
 * frames (mh_mel with the torchaudio parameterisation, log1p) -- the time-major transpose of HF's input_features --
 * and the same call runs conv1 + GELU, conv2 (stride 2) + GELU, the encoder layers (RMSNorm -> Wqkv -> RoPE ->
 * softmax(q k^T / 8) v -> Wo -> + ; RMSNorm -> fc1 -> GELU -> fc2 -> +) and the final RMSNorm; enc_out [B * src_len, d]. */

/* The same with the wrapper's conditioning embedders (difficulty / mapper style / song position / style:
 * modeling_mapperatorinator.py:395-414 -- per-row vectors repeated over the frames and concatenated to the mel frames in
 * front of encoder_embedder).  A vector that is constant along the frames only adds a constant to every frame of its
 * chunk: row_bias [B, d_model] fp32 = cond @ W[:, n_mels:]^T + b (computed by the caller; it REPLACES the bias of
 * encoder_embedder), NULL = mh_t5_encode. */
int mh_t5_encode_cond(const MhT5Config* cfg, const MhT5Weights* w, const void* mel, int B, const float* row_bias,
                      void* enc_out, float* enc_out_f32, void* workspace, int64_t workspace_bytes, void* stream);

/* Cross-attention K/V projection of all decoder layers at once (HF T5Attention with
 * key_value_states, computed once per chunk and kept in the encoder-side StaticCache:
 * osuT5/osuT5/inference/cache_utils.py:32-35).
 * cross_kv out: [n_dec][2][B][H][L][64] element type cfg.dtype. */
/* (ABI 7) the same with scratch for the MX-fp8 copy of enc_out when cfg->enc_operand_dtype = MH_MX8 (mh_t5_cross_kv refuses
 * that mode: it has nowhere to put the copy); workspace >= mh_t5_cross_kv_workspace_bytes(cfg, B) (0 for the plain modes). */
int64_t mh_t5_cross_kv_workspace_bytes(const MhT5Config* cfg, int B);
int mh_t5_cross_kv_ws(const MhT5Config* cfg, const MhT5Weights* w, const void* enc_out, int B, void* cross_kv, void* workspace,
                      int64_t workspace_bytes, void* stream);
int mh_t5_cross_kv(const MhT5Config* cfg, const MhT5Weights* w, const void* enc_out, int B,
                   void* cross_kv, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K5/K6  autoregressive decode.  Replaces HF GenerationMixin.generate/_sample as driven by
 * model_generate (osuT5/osuT5/inference/server.py:83-156) with MapperatorinatorCache
 * (cache_utils.py:7-35) and the reference logits processors
 * (osuT5/osuT5/inference/logit_processors.py:36-44,136-183; server.py:106-134).
 */
typedef struct MhSampling {
  int do_sample;            /* 0: greedy argmax (first maximal index)                          */
  int top_k;                /* >0: keep k best before sampling                                 */
  float top_p;              /* <1: nucleus                                                     */
  float temperature;        /* TemperatureLogitsWarper (server.py:131-132)                     */
  float timeshift_bias;     /* TimeshiftBias (logit_processors.py:36-44), 0 = off              */
  int ts_start, ts_end;     /* TIME_SHIFT id range [start, end)                                */
  int n_sos; int sos_ids[16];   /* sos + context_sos ids (MonotonicTimeShiftLogitsProcessor)   */
  int lookback_mask_end;    /* LookbackBiasLogitsWarper, types_first=False branch: ids in
                               [ts_start, lookback_mask_end) get -inf; <= ts_start disables     */
  int pad_id;
  int max_length;           /* prompt + new tokens cap (MaxLengthCriteria)                      */
  uint64_t seed;            /* Philox seed when do_sample                                       */
  /* --- ABI 2: the remaining processors of server.py:106-134 ----------------------------------- */
  float cfg_scale;          /* > 1: ClassifierFreeGuidanceLogitsProcessor (server.py:107-108).  The batch
                               holds 2G rows: rows [0, G) are fed the NEGATIVE prompt, rows [G, 2G) the
                               prompt (prepare_inputs_for_generation, modeling_mapperatorinator.py:243-254);
                               scores = l[G+g] + (l[g] - l[G+g]) * cfg_scale, exactly what HF's processor
                               computes on that row order; both rows of a pair receive the sampled id;
                               cross_kv then holds G rows (the encoder output is shared by the pair)      */
  int n_cond;               /* ConditionalTemperatureLogitsWarper (logit_processors.py:47-82): up to 3   */
  float cond_temp[3];       /*   (temperature, token set, offset) rules, first match wins; the token set */
  int cond_offset[3];       /*   of rule j is bit (2 << j) of tok_flags; the lookback is ROW 0's history */
  int lookback_types_first; /* LookbackBiasLogitsWarper types_first=True branch (:116-133) over
                               [ts_start, lookback_mask_end): renormalise instead of masking             */
  const uint8_t* tok_flags; /* device uint8 [vocab_out]: bit0 timed-event ids (:104-108), bits 1-3 the
                               conditional-temperature sets, bit4 eos + context-eos ids (:99).  May be
                               NULL when n_cond == 0 and lookback_types_first == 0                       */
  /* --- ABI 3 ------------------------------------------------------------------------------------ */
  int cond_per_row;         /* 0: the reference's batch behaviour -- ConditionalTemperatureLogitsWarper reads
                               `input_ids[0, -max_offset:]`, ROW 0's history, for every row of the batch
                               (logit_processors.py:75-80).  1: every row looks at its OWN history, i.e. what
                               the reference computes when each row is a batch-1 call (generate_sequential,
                               processor.py:308-368): used when rows of different songs / shards share a batch */
  unsigned rng_row0;        /* do_sample: global index of this call's first returned row in the RNG key
                               (seed, row, column), so shards of one job do not replay each other's draws  */
  /* --- ABI 4 ------------------------------------------------------------------------------------ */
  const void* cross_kv_fp8; /* NULL: the token steps stream `cross_kv` (cfg.dtype).  Non-NULL (bf16 storage only): the
                               packed OCP e4m3 copy written by mh_t5_quantize_cross_kv -- the token steps stream THAT
                               (half the HBM bytes of the dominant kernel; BASELINE configs[4]); the prompt prefill
                               still reads `cross_kv`.  Not a parity mode: K / V carry 3 mantissa bits.          */
} MhSampling;

int64_t mh_t5_decode_workspace_bytes(const MhT5Config* cfg, int B);

/* fp8 form of the cross-attention K/V (no reference counterpart; BASELINE configs[4] "fp8 ... KV-cached decode").
 * cross_kv  [n_dec][2][B][H][L][64] bf16 from mh_t5_cross_kv;
 * out       mh_t5_cross_kv_fp8_bytes(cfg, B) bytes: the same layout in OCP e4m3, one byte per element, followed (256-
 *           byte aligned) by fp32 scales [n_dec][2][B][H]: x ~ e4m3 * scale, scale = absmax of the (layer, k|v, row,
 *           head) slab / 448. */
int64_t mh_t5_cross_kv_fp8_bytes(const MhT5Config* cfg, int B);
int mh_t5_quantize_cross_kv(const MhT5Config* cfg, const void* cross_kv, int B, void* out, void* stream);

/* Runs prefill over the (left-padded) prompt and the AR loop until every row has emitted an id of
 * the EOS set or max_length is reached.
 *   prompt      int32 [B, P]    decoder_input_ids (pad_id = left padding)
 *   prompt_mask uint8 [B, P]    decoder_attention_mask (1 = attend), NULL = all ones
 *   eos_table   uint8 [vocab_out] 1 where the id is in get_eos_token_id(...) (server.py:72-80)
 *   cross_kv    from mh_t5_cross_kv
 *   tokens      int32 [B, max_length] out: prompt followed by generated ids, pad_id after EOS
 *   n_steps_out int32 [1] device: number of valid columns of `tokens` (= HF output length)
 *   logits_dump optional fp32 [max_length][B][vocab_out]: processed scores of every step (parity)
 *   forced      optional int32 [B, max_length]: teacher forcing -- when non-NULL the id appended at
 *               column c is forced[b][c] (argmax is still computed and written to `tokens`)
 * The loop is captured into a hipGraph per step shape and replayed; the host polls a device
 * "all finished" flag every `poll_every` steps (one 4-byte D2H on `stream`), otherwise no sync. */
int mh_t5_generate(const MhT5Config* cfg, const MhT5Weights* w, const void* cross_kv, int B,
                   const int32_t* prompt, const uint8_t* prompt_mask, int P, const uint8_t* eos_table,
                   const MhSampling* sp, int32_t* tokens, int32_t* n_steps_out, float* logits_dump,
                   const int32_t* forced, void* workspace, int64_t workspace_bytes, int poll_every,
                   void* stream);

/* (ABI 10) One beam-search step between two decoder positions as ONE kernel (csrc/beam.hip).  Replaces the per-step body of HF
 * `GenerationMixin._beam_search` (third-party, transformers >= 4.50's vectorised form) as the reference reaches it with
 * `num_beams > 1` (osuT5/osuT5/inference/processor.py:147,159; server.py:137; super_timing_generator.py:28 decodes with two beams):
 * log_softmax of the step's logits (mh_t5_step) -> [HF ClassifierFreeGuidanceLogitsProcessor on the reference's row order, rows
 * [negative | prompt]: modeling_mapperatorinator.py:243-254] -> the reference's processor list on log-probabilities (server.py:
 * 106-134; logit_processors.py:36-44,47-82,111-114,136-183) -> + running beam scores -> the K best continuations of every chunk
 * (sorted, ties by flat index) -> EOS / max_length split -> the next running beams -> merge of the finished hypotheses by score /
 * length ** penalty -> the early_stopping = False heuristic.  Greedy beams only (sp.do_sample = 0).
 * State: G chunks x num_beams rows, int32 sequences [G][nb][max_length] (columns beyond the current length hold the fill value),
 * fp32 scores [G][nb], int32 beam-index trails [G][nb][max_length - P], finished flags [G][nb]; the kernel reads the *_in arrays
 * and writes the *_out arrays (the caller swaps them every step), `heuristic_open` [G] in place.  Outputs for the caller (G nb entries,
 * 2 G nb under guidance: the second half repeats the first, as `beam_idx.repeat(2)` does): `src` [G
 * nb] = the row each new running beam continues (the argument of mh_t5_reorder_cache: MapperatorinatorCache.reorder_cache,
 * inference/cache_utils.py:16-20), `last` [G nb] = the token each running beam is fed next, `flags` [G][3] = (heuristic still
 * open, every candidate hit EOS / max_length, every finished slot filled) from which the host forms HF's loop condition.
 * num_beams in 2 .. 8, K <= 4096, 4 num_beams V + 8 K' bytes (K' = K rounded up to a power of two) within 120 KB of LDS. */
typedef struct MhBeamStep {
  const float* logits;          /* [RE][V] fp32: RE = G nb rows, or 2 G nb under guidance ([negative rows | prompt rows]) */
  const uint8_t* eos_table;     /* [V] 1 = an EOS id (get_eos_token_id, server.py:72-80)                                    */
  int G, num_beams, V, P, max_length, K, cur_len;
  int cfg; float cfg_scale;     /* classifier-free guidance (sp.cfg_scale is not read)                                      */
  float length_penalty; int early_stopping;   /* 0 = False, 1 = True, 2 = "never"                                           */
  MhSampling sp;                /* ts_start / ts_end / sos_ids / timeshift_bias / temperature / cond_* / lookback_mask_end / tok_flags */
  const int32_t* run_in; const float* rs_in; const int32_t* rb_in;      /* running beams: sequences, scores, beam-index trail */
  const int32_t* seq_in; const float* bs_in; const int32_t* bb_in; const uint8_t* fin_in;   /* finished set                 */
  int32_t* run_out; float* rs_out; int32_t* rb_out;
  int32_t* seq_out; float* bs_out; int32_t* bb_out; uint8_t* fin_out;
  uint8_t* heuristic_open;      /* [G] in place                                                                              */
  int32_t* src; int32_t* last; int32_t* flags;
} MhBeamStep;
int64_t mh_beam_step_lds_bytes(int num_beams, int V);
int mh_beam_step(const MhBeamStep* bs, void* stream);

/* Step-wise decode for host-driven search.  Replaces the per-position `self(**model_inputs)` of HF
 * `GenerationMixin._beam_search` (num_beams > 1: osuT5/osuT5/inference/processor.py:147,159; server.py:137) and
 * `MapperatorinatorCache.reorder_cache` (osuT5/osuT5/inference/cache_utils.py:16-20).
 *   mh_t5_step: feeds ids[b] (device int32 [B]) at position `pos` to the decoder (self K/V appended at `pos`) and writes the
 *     RAW logits fp32 [B, vocab_out] -- no processors, no selection.  Rows are (chunk, beam) pairs, kv_group beams per
 *     chunk: row b reads cross K/V row b / kv_group (cross_kv holds B / kv_group rows).  prompt_mask uint8 [B, P] as in
 *     mh_t5_generate (NULL = ones).  `workspace` = mh_t5_decode_workspace_bytes(cfg, B) bytes, owned by the caller and kept
 *     between calls: it holds the self-attention caches.
 *   mh_t5_reorder_cache: cache rows b <- src[b] (device int32 [B]) for positions 0 .. n_pos-1 of every layer;
 *     scratch = mh_t5_reorder_cache_scratch_bytes(cfg, B, n_pos) bytes. */
int mh_t5_step(const MhT5Config* cfg, const MhT5Weights* w, const void* cross_kv, int B, int kv_group, const int32_t* ids,
               int pos, const uint8_t* prompt_mask, int P, float* logits, void* workspace, int64_t workspace_bytes,
               void* stream);
int64_t mh_t5_reorder_cache_scratch_bytes(const MhT5Config* cfg, int B, int n_pos);
int mh_t5_reorder_cache(const MhT5Config* cfg, int B, const int32_t* src, int n_pos, void* workspace, int64_t workspace_bytes,
                        void* scratch, int64_t scratch_bytes, void* stream);

/* Teacher-forced decoder forward over whole sequences: `Mapperatorinator.forward` with `encoder_outputs` given
 * (seam B2, osuT5/osuT5/model/modeling_mapperatorinator.py:174-228; used by server.py:160-181 `model_forward`).
 *   ids  int32 [B, T] decoder_input_ids, mask uint8 [B, T] decoder_attention_mask (NULL = ones), T <= tgt_len
 *   logits fp32 [B, T, vocab_out]: lm_head(final RMSNorm(decoder(ids)))  -- no processors, no sampling */
int64_t mh_t5_forward_workspace_bytes(const MhT5Config* cfg, int B, int T);
int mh_t5_decoder_forward(const MhT5Config* cfg, const MhT5Weights* w, const void* cross_kv, int B,
                          const int32_t* ids, const uint8_t* mask, int T, float* logits, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* Measurement hook (bench.py `roofline`): launches the dominant decode kernel -- cross-attention over
 * the encoder keys, algorithmic bytes per launch = B*H*src_len*64*2*sizeof(elem) -- `reps` times
 * back to back between two HIP events on `stream`, cycling through the decoder layers as a decode
 * step does.  With `w` given it is the kernel the decode step really launches (the cross-attention that also
 * projects its own query from the residual row); w == NULL times the stand-alone attention kernel.
 * ms_out (HOST float[1]) = average milliseconds per launch.  Synchronises `stream`. */
int mh_t5_cross_attn_probe(const MhT5Config* cfg, const MhT5Weights* w, const void* cross_kv, int B, int reps,
                           float* ms_out, void* workspace, int64_t workspace_bytes, void* stream);

/* Measurement hook (bench.py `roofline`, in situ): with `buf` set, every cross-attention launch of the following
 * mh_t5_generate calls records (earliest workgroup start, latest workgroup end) in wall-clock ticks
 * (hipDeviceAttributeWallClockRate, kHz) into buf[chain][pos][layer][2] for decode positions pos < ring (device uint64,
 * pre-filled by the caller with (UINT64_MAX, 0)): the duration of the kernel exactly as the decode step launches it -- rows of one chain per
 * launch, the other chain's kernels running beside it.  Two atomics per workgroup: use an extra decode pass, not a
 * timed one.  buf == NULL (ring 0) switches it off.  mh_t5_decode_chains(B) = row chains mh_t5_generate uses for B. */
int mh_t5_decode_timing(void* buf, int ring);
int mh_t5_decode_chains(int B);
int mh_t5_decode_chains_cfg(const MhT5Config* cfg, int B);   /* ABI 8: ... under cfg->options */
/* ABI 8: instantiated step graphs are kept across mh_t5_generate calls (option "decode_graph_cache", default 1: LRU of 16, a
 * call replays an earlier call's graph iff every address, size, sampling field other than seed / rng_row0, option value and the
 * weights struct agree byte for byte).  Counters of graphs replayed from / captured into the cache since the last reset. */
int mh_t5_step_graph_cache_stats(long* hits, long* misses, int reset);
int mh_wall_clock_khz(void);

/* ------------------------------------------------------------------------------------------------
 * K7/K8/K9  osu_diffusion DiT + DDPM.  Replaces DiT.forward_with_cfg
 * (osu_diffusion/utils/models.py:281-317) and GaussianDiffusion.p_sample
 * (osu_diffusion/utils/diffusion/gaussian_diffusion.py:273-369, 420-467).  fp32 like the reference (it never
 * casts the DiT: inference.py:637-642) by default; operand_dtype = MH_BF16 is the reduced-precision mode of BASELINE
 * configs[4]: the four projections of every block take bf16 operands (weights stored bf16; LayerNorm-modulate, q / k / v,
 * the attention output and the GELU hidden are rounded to bf16 where they become a GEMM operand), one bf16 MFMA pass
 * with fp32 accumulation; the residual stream, LayerNorm, softmax, the adaLN conditioning, the first and the final
 * layer and the DDPM update stay fp32.  Parity of that mode is an error bound against the fp32 reference golden.
 */
typedef struct MhDiTConfig {
  int hidden, depth, n_heads, context_size, class_size, in_channels; /* in_channels = 2 */
  int freq_dim;      /* FirstLayer frequency_embedding_size = 128                        */
  int t_freq_dim;    /* TimestepEmbedder frequency_embedding_size = 256                  */
  int first_k_pad;   /* padded K of the first layer GEMM (in_channels*freq_dim+context)  */
  int class_pad;     /* padded K of y_embedder[0]                                        */
  int operand_dtype; /* MH_F32 (reference semantics), MH_BF16 (block GEMM operands, see above) or -- ABI 7, BASELINE configs[4]
                        "fp8 MFMA" -- MH_MX8: the four block projections on MX-fp8 operands (LayerNorm-modulate writes the operand
                        directly; attention output and GELU hidden are bf16 as in the MH_BF16 mode and quantised by a pass of their
                        own; attention itself on bf16 operands); hidden %% 128 == 0.  Error-bound gates, as MH_BF16. */
  const MhOptionSet* options;   /* ABI 8: this engine's option overrides, NULL = the process-wide values */
} MhDiTConfig;

typedef struct MhDiTWeights {       /* all fp32, matrices [N][Kpad]                                  */
  const float* pos_freqs;   /* [freq_dim/2]   exp(-ln(1e4) i / (freq_dim/2)), host-built exactly as     */
  const float* t_freqs;     /* [t_freq_dim/2] timestep_embedding does (positional_embedding.py:38-43)   */
  const float* first_w; const float* first_b;        /* context_embedder.mlp[0]  models.py:194-201  */
  const float* t_w0; const float* t_b0; const float* t_w1; const float* t_b1;   /* t_embedder.mlp   */
  const float* y_w0; const float* y_b0; const float* y_w1; const float* y_b1;   /* y_embedder       */
  const float* ada_w[MH_MAX_LAYERS]; const float* ada_b[MH_MAX_LAYERS];   /* [6D, D]               */
  const float* qkv_w[MH_MAX_LAYERS]; const float* qkv_b[MH_MAX_LAYERS];   /* attn.in_proj [3D, D]  */
  const float* out_w[MH_MAX_LAYERS]; const float* out_b[MH_MAX_LAYERS];   /* attn.out_proj         */
  const float* fc1_w[MH_MAX_LAYERS]; const float* fc1_b[MH_MAX_LAYERS];   /* mlp.fc1 [4D, D]       */
  const float* fc2_w[MH_MAX_LAYERS]; const float* fc2_b[MH_MAX_LAYERS];   /* mlp.fc2 [D, 4D]       */
  const float* fin_ada_w; const float* fin_ada_b;    /* final_layer.adaLN_modulation [2D, D]       */
  const float* fin_w; const float* fin_b;            /* final_layer.linear [4, D]                  */
  /* optional pre-split copies (MhGemm.w_split3 layout) of the five big projections; NULL = exact fp32 only.  Used for
   * denoiser batches of at least `dit_split3_min_rows` rows (option, default 2048 = 8 chunks of 128 points). */
  const void* first_w3;
  const void* qkv_w3[MH_MAX_LAYERS]; const void* out_w3[MH_MAX_LAYERS];
  const void* fc1_w3[MH_MAX_LAYERS]; const void* fc2_w3[MH_MAX_LAYERS];
  /* bf16 copies [N][K] of the four block projections; required when operand_dtype = MH_BF16 */
  const void* qkv_wb[MH_MAX_LAYERS]; const void* out_wb[MH_MAX_LAYERS];
  const void* fc1_wb[MH_MAX_LAYERS]; const void* fc2_wb[MH_MAX_LAYERS];
  /* (ABI 7) MX-fp8 copies of the four block projections (mh_quantize_mx8 layout: e4m3 [N][K] | scales [N][mh_mx8_scale_row_bytes(K)]);
   * required when operand_dtype = MH_MX8 */
  const uint8_t* qkv_wm[MH_MAX_LAYERS]; const uint8_t* qkv_wms[MH_MAX_LAYERS];
  const uint8_t* out_wm[MH_MAX_LAYERS]; const uint8_t* out_wms[MH_MAX_LAYERS];
  const uint8_t* fc1_wm[MH_MAX_LAYERS]; const uint8_t* fc1_wms[MH_MAX_LAYERS];
  const uint8_t* fc2_wm[MH_MAX_LAYERS]; const uint8_t* fc2_wms[MH_MAX_LAYERS];
} MhDiTWeights;

int64_t mh_dit_workspace_bytes(const MhDiTConfig* cfg, int N, int T);

/* One denoiser evaluation with classifier-free guidance.
 *   x [N,2,T] fp32 (first half is duplicated internally as forward_with_cfg does, models.py:306-307)
 *   t [N] int32 original-scale timesteps (already mapped by _WrappedModel, respace.py:127-132)
 *   c [N,context,T] fp32;  y [N,class] fp32;  band: |i-j| < band attends (banded bool mask of
 *   diffusion_pipeline.py:146-148), band <= 0 = full attention
 *   open_from: with `pad_sequence` (diffusion_pipeline.py:186-193) the window is padded to max_seq_len and the band mask
 *   is padded with "allowed" -- positions >= open_from attend and are attended by everything (the reference builds a
 *   key_padding_mask for them but DiTBlock.forward never hands it to the attention, models.py:133-150); 0 = no padding
 *   out [N,4,T] fp32: CFG-combined eps (duplicated) ++ variance channels (models.py:312-317). */
int mh_dit_forward_cfg(const MhDiTConfig* cfg, const MhDiTWeights* w, const float* x, const int32_t* t,
                       const float* c, const float* y, float cfg_scale, int band, int open_from, int N, int T,
                       float* out, void* workspace, int64_t workspace_bytes, void* stream);

/* One p_sample update (learned-range variance, epsilon prediction, clip_denoised -> clamp(-2,2)):
 *   coef fp32[7] for this step, extracted on the host from the float64 schedule exactly as
 *   _extract_into_tensor does (gaussian_diffusion.py:951-963):
 *     {posterior_log_variance_clipped, log(beta), sqrt_recip_alphas_cumprod,
 *      sqrt_recipm1_alphas_cumprod, posterior_mean_coef1, posterior_mean_coef2, nonzero_mask}
 *   model_out [N,4,T]; x [N,2,T] in; noise [N,2,T]; inpaint_mask uint8 [N,2,T] (1 = generate) and
 *   inpaint_ref [N,2,T] implement denoised_fn's `torch.where(mask, x, z_part)`
 *   (diffusion_pipeline.py:203-206), both NULL to skip;  x_out [N,2,T]; pred_xstart optional.
 *   Arbitrary host `denoised_fn`s (the slider re-projection of diffusion_pipeline.py:208-220) are
 *   served by two calls: raw_pred=1 writes the unprocessed eps->x0 prediction to pred_xstart and
 *   returns; the caller transforms it and passes it back as x0_override (clamp, posterior mean and
 *   the noise step then proceed from that value). */
int mh_ddpm_step(const float* model_out, const float* x, const float* noise, const float* coef,
                 const uint8_t* inpaint_mask, const float* inpaint_ref, const float* x0_override,
                 int raw_pred, int N, int T, float* x_out, float* pred_xstart, void* stream);

/* Sliders whose END point `denoised_fn` re-projects onto the slider's own path every denoising step (reference
 * diffusion_pipeline.py:208-220; DiffusionSlider :30-35; SliderPath osuT5/osuT5/inference/slider_path.py:26-230 and
 * path_approximator.py).  All arrays live in DEVICE memory; indices are relative to the window (column of x0).
 *   chunk b owns rows b and b + pair_stride of x0 (the CFG pair layout [cond_0..cond_{B-1} | null_0..null_{B-1}] of
 *   mh_dit_forward_cfg: pair_stride = B = n_chunks, N = 2 B), or just row b when pair_stride = 0 (n_chunks = N).
 *   chunk_active[b] != 0: the song of chunk b has sliders at all -- its positions then take the reference's pixel round
 *   trip ((x + 1) / 2 * (512, 384)) / (512, 384) * 2 - 1 and row b is broadcast over its pair, exactly as
 *   `x[:, :, :] = ...` (:220) does, even when no slider of the song lies inside this window.
 *   Sliders chunk_off[b] .. chunk_off[b+1] belong to chunk b; slider s has control points cp_idx[cp_off[s] .. cp_off[s+1]),
 *   curve type[s] (0 Linear, 1 PerfectCurve, 2 Catmull, 3 Bezier), end point end_idx[s] and length[s] (playfield pixels).
 *   The caller leaves out sliders that are not entirely inside the window (:210-212) and guarantees that no end_idx is
 *   another slider's control point or end (true for event streams: a SLIDER_END is its own point).  One Bezier span
 *   (control points between repeated points) holds at most 32 points; a longer one yields NaN positions. */
typedef struct MhSliderSet {
  int32_t n_chunks, pair_stride, n_sliders;
  const uint8_t* chunk_active; /* [n_chunks] */
  const int32_t* chunk_off;    /* [n_chunks + 1] */
  const int32_t* type;         /* [n_sliders] */
  const int32_t* cp_off;       /* [n_sliders + 1] */
  const int32_t* cp_idx;       /* [cp_off[n_sliders]] */
  const int32_t* end_idx;      /* [n_sliders] */
  const double* length;        /* [n_sliders] */
} MhSliderSet;

/* `denoised_fn` of the pipeline on x0 [N,2,T] in place: x0 = where(mask, x0, ref) (mask / ref may be NULL), then the
 * slider end re-projection above.  The path arithmetic follows the reference's numpy dtype flow (float32 Linear /
 * Catmull / circular-arc paths, float64 Bezier subdivision). */
int mh_slider_project(float* x0, const uint8_t* inpaint_mask, const float* inpaint_ref, int N, int T,
                      const MhSliderSet* sliders, void* stream);

/* Whole p_sample_loop on device (gaussian_diffusion.py:469-561): `n_steps` iterations of
 * mh_dit_forward_cfg + mh_ddpm_step captured once into a hipGraph and replayed.
 *   t_map int32 [n_steps] : timestep_map[i] for loop index i (SpacedDiffusion, respace.py:72-86)
 *   coefs fp32 [n_steps][7]; noise fp32 [n_steps][N,2,T], both indexed by loop index i
 *   (the loop runs i = n_steps-1 ... 0).  x_io [N,2,T] is updated in place.
 *   Everything that depends only on (t, y) -- timestep/label embedders and every block's adaLN modulation --
 *   is computed for all steps before the loop; workspace size from mh_ddpm_loop_workspace_bytes.
 *   sliders (NULL = none): the step becomes eps -> x0, mh_slider_project (in-paint + slider ends), posterior + noise --
 *   still inside the one captured graph. */
int64_t mh_ddpm_loop_workspace_bytes(const MhDiTConfig* cfg, int N, int T, int n_steps);
int mh_ddpm_sample_loop(const MhDiTConfig* cfg, const MhDiTWeights* w, float* x_io, const float* c,
                        const float* y, float cfg_scale, int band, int open_from, int N, int T, int n_steps,
                        const int32_t* t_map, const float* coefs, const float* noise,
                        const uint8_t* inpaint_mask, const float* inpaint_ref, const MhSliderSet* sliders,
                        void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAPPERHIP_H_ */
