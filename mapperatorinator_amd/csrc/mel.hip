// K1  framed STFT + mel filterbank (fp32), replacing nnAudio.features.MelSpectrogram as used at
// osuT5/osuT5/model/spectrogram.py:50-61,63-83 (center=True zero padding, periodic hann, power 2,
// Slaney/area-normalised filterbank, optional log1p, (B, frames, n_mels) output).
//
// One workgroup (256 threads) transforms TWO consecutive frames at once: frame 2f is the real part and
// frame 2f+1 the imaginary part of one 1024-point complex radix-4 Stockham FFT held in LDS (5 passes,
// ping-pong buffers, host-computed float64->fp32 twiddles); the two half spectra are separated with the
// conjugate-symmetry identity, squared, and the sparse (CSR) triangular filterbank is applied from LDS.
// HBM traffic = audio read once (8x overlap served by L2) + mel written once: the kernel is
// launch/HBM-bound (2.58 MB per 10 s chunk).
#include "internal.hpp"

namespace mh {
namespace {

constexpr int NFFT = 1024;

struct MelP {
  const float* audio; int B, n_samples, hop, n_mels, n_frames;
  const float* window; const float2* tw;
  const int32_t* fb_start; const int32_t* fb_len; const int32_t* fb_off; const float* fb_w;
  int log_scale, reflect; void* out; int ld_out;
};

__device__ inline float2 cmul_conj_tw(float2 v, float2 w) {  // v * (w.x - i w.y)  == v * exp(-i theta)
  return make_float2(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y);
}

template <typename TO>
__global__ __launch_bounds__(256) void mel_kernel(MelP p) {
  __shared__ float2 buf[2][NFFT];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * 2, f1 = f0 + 1;
  const bool has1 = f1 < p.n_frames;
  const float* au = p.audio + (long)b * p.n_samples;

  // ---- pass 0 (Ns = 1): inputs straight from global memory, windowed ----
  float2 v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = tid + 256 * r;
    const float w = p.window[n];
    const int i0 = f0 * p.hop + n - NFFT / 2;
    const int i1 = i0 + p.hop;
    float a, c;
    if (p.reflect) {   // torch.stft center=True, pad_mode='reflect': sample -i for i < 0, 2 (Ns - 1) - i beyond the end
      const int last = p.n_samples - 1;
      int j0 = i0 < 0 ? -i0 : (i0 > last ? 2 * last - i0 : i0);
      int j1 = i1 < 0 ? -i1 : (i1 > last ? 2 * last - i1 : i1);
      j0 = j0 < 0 ? 0 : (j0 > last ? last : j0);
      j1 = j1 < 0 ? 0 : (j1 > last ? last : j1);
      a = au[j0];
      c = has1 ? au[j1] : 0.f;
    } else {
      a = (i0 >= 0 && i0 < p.n_samples) ? au[i0] : 0.f;
      c = (has1 && i1 >= 0 && i1 < p.n_samples) ? au[i1] : 0.f;
    }
    v[r] = make_float2(a * w, c * w);
  }
  int cur = 0;
#pragma unroll
  for (int pass = 0; pass < 5; ++pass) {
    const int Ns = 1 << (2 * pass);
    const int jm = tid & (Ns - 1);
    if (pass > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = buf[cur ^ 1][tid + 256 * r];
      const int tstep = 256 / Ns;  // twiddle index unit: exp(-2 pi i * r*jm / (4 Ns)) = tw[r*jm*256/Ns]
#pragma unroll
      for (int r = 1; r < 4; ++r) v[r] = cmul_conj_tw(v[r], p.tw[r * jm * tstep]);
    }
    // radix-4 forward butterfly
    const float2 t0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y);
    const float2 t1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
    const float2 t2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y);
    const float2 d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
    const float2 t3 = make_float2(d.y, -d.x);  // -i * (v1 - v3)
    const int j0 = ((tid - jm) << 2) + jm;     // (tid / Ns) * Ns * 4 + tid % Ns
    buf[cur][j0] = make_float2(t0.x + t2.x, t0.y + t2.y);
    buf[cur][j0 + Ns] = make_float2(t1.x + t3.x, t1.y + t3.y);
    buf[cur][j0 + 2 * Ns] = make_float2(t0.x - t2.x, t0.y - t2.y);
    buf[cur][j0 + 3 * Ns] = make_float2(t1.x - t3.x, t1.y - t3.y);
    __syncthreads();
    cur ^= 1;
  }
  // result is in buf[cur ^ 1]; reuse buf[cur] for the two power spectra
  const float2* Z = buf[cur ^ 1];
  float* PA = reinterpret_cast<float*>(buf[cur]);
  float* PB = PA + (NFFT / 2 + 1);
  for (int k = tid; k <= NFFT / 2; k += 256) {
    const float2 zk = Z[k];
    const float2 zn = Z[(NFFT - k) & (NFFT - 1)];
    const float ar = 0.5f * (zk.x + zn.x), ai = 0.5f * (zk.y - zn.y);
    const float br = 0.5f * (zk.y + zn.y), bi = -0.5f * (zk.x - zn.x);
    const float ma = sqrtf(ar * ar + ai * ai);  // nnAudio: sqrt(re^2+im^2) ** 2
    const float mb = sqrtf(br * br + bi * bi);
    PA[k] = ma * ma;
    PB[k] = mb * mb;
  }
  __syncthreads();
  TO* o0 = reinterpret_cast<TO*>(p.out) + ((long)b * p.n_frames + f0) * p.ld_out;
  TO* o1 = o0 + p.ld_out;
  for (int mI = tid; mI < p.ld_out; mI += 256) {
    float sa = 0.f, sb = 0.f;
    if (mI < p.n_mels) {
      const int st = p.fb_start[mI], ln = p.fb_len[mI], of = p.fb_off[mI];
      for (int i = 0; i < ln; ++i) {
        const float w = p.fb_w[of + i];
        sa += w * PA[st + i];
        sb += w * PB[st + i];
      }
      if (p.log_scale) { sa = log1pf(sa); sb = log1pf(sb); }
    }
    o0[mI] = Elem<TO>::from_f32(sa);
    if (has1) o1[mI] = Elem<TO>::from_f32(sb);
  }
}

}  // namespace
}  // namespace mh

extern "C" int mh_mel(const float* audio, int B, int n_samples, int n_fft, int hop, int n_mels, const float* window,
                      const float* twiddle, const int32_t* fb_start, const int32_t* fb_len, const int32_t* fb_off,
                      const float* fb_w, int log_scale, void* out, int ld_out, int out_dtype, void* stream) {
  using namespace mh;
  MH_REQUIRE(audio && window && twiddle && fb_start && fb_len && fb_off && fb_w && out, "mh_mel: null pointer");
  MH_REQUIRE(n_fft == NFFT, "mh_mel: only n_fft=1024 is built (got %d)", n_fft);
  MH_REQUIRE(B > 0 && n_samples > 0 && hop > 0 && n_mels > 0 && ld_out >= n_mels, "mh_mel: bad shape");
  MelP p;
  p.audio = audio; p.B = B; p.n_samples = n_samples; p.hop = hop; p.n_mels = n_mels;
  p.n_frames = n_samples / hop + 1;
  p.window = window; p.tw = reinterpret_cast<const float2*>(twiddle);
  p.fb_start = fb_start; p.fb_len = fb_len; p.fb_off = fb_off; p.fb_w = fb_w;
  p.log_scale = log_scale & 1; p.reflect = (log_scale >> 1) & 1; p.out = out; p.ld_out = ld_out;
  dim3 grid((p.n_frames + 1) / 2, B), block(256);
  if (out_dtype == MH_BF16)
    hipLaunchKernelGGL(mel_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(mel_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
  return check_launch("mel_kernel");
}
