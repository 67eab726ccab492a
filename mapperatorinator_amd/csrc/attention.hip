// Fused (flash-style) multi-head attention for head_dim = 64 on the gfx950 matrix cores.
//   T5 encoder self-attention : softmax(Q K^T + rel_bias[h][k - q]) V, scale 1 (T5 folds 1/sqrt(d) into init)
//                               HF T5Attention.forward; restated custom_transformers/t5.py:170-250
//   DiT self-attention        : softmax(Q K^T / 8 + band mask) V (nn.MultiheadAttention with the banded
//                               bool mask of diffusion_pipeline.py:146-148; models.py:111-116,145-151)
// Layout: block = 4 waves, 64 query rows (16 per wave) of one (batch, head); K tile [64 keys][64] and the
// pre-transposed V tile [64 d][64 keys] staged in LDS with 16-byte accesses; S and O live in MFMA
// accumulators, softmax reductions are 16-lane shuffles, P goes through a wave-private LDS patch to
// become the A operand of the PV product.  V^T is produced by the QKV GEMM epilogue (MH_EPI_QKV_VT).
#include "internal.hpp"

namespace mh {
namespace {

struct AttnP {
  const char* qk; long ld_qk_b; int k_col0;
  const char* vt; int Lpad;
  const float* bias;
  char* out; long ld_out_b;
  int B, L, H;
  float scale; int band;
};


template <typename T>
__global__ __launch_bounds__(256) void flash_attn_kernel(AttnP p) {
  constexpr int ES = (int)sizeof(T);
  constexpr int KM = Atom<T>::KM, KCH = Atom<T>::KCH;
  constexpr int KS = 64 / KM;               // MFMA k-steps over a 64-wide contraction
  constexpr int RS = 64 * ES + 16;          // padded LDS row stride (bytes)
  constexpr int CPR = 64 * ES / 16;         // 16-byte chunks per tile row
  constexpr int CPT = 64 * CPR / 256;       // chunks per thread per tile

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vts = smem + 64 * RS;
  char* Ps = smem + 128 * RS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int L = p.L;
  const int qb0 = qt * 64;
  const int q0w = qb0 + wid * 16;
  const int l15 = lane & 15, lg = lane >> 4;

  // Q fragments straight from global memory (A operand: row = l15, k chunk = lg)
  typename Atom<T>::frag_t qf[KS];
  {
    int qr = q0w + l15;
    qr = qr < L ? qr : L - 1;
    const char* qp = p.qk + (long)(b * L + qr) * p.ld_qk_b + (long)(h * 64 + lg * KCH) * ES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = Atom<T>::load(reinterpret_cast<const T*>(qp + ks * KM * ES));
  }

  f32x4_t o[4];
  float m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -1e30f; l[r] = 0.f; }

  int t_lo = 0, t_hi = (L - 1) / 64;
  if (p.band > 0) {
    int klo = qb0 - (p.band - 1);
    klo = klo < 0 ? 0 : klo;
    int khi = qb0 + 63 + p.band;
    khi = khi > L - 1 ? L - 1 : khi;
    t_lo = klo / 64;
    t_hi = khi / 64;
  }

  const char* kbase = p.qk + (long)(p.k_col0 + h * 64) * ES;
  const char* vbase = p.vt + ((long)(b * p.H + h) * 64) * (long)p.Lpad * ES;
  char* Pw = Ps + wid * 16 * RS;

  for (int kt = t_lo; kt <= t_hi; ++kt) {
    const int kv0 = kt * 64;
    // ---- stage K tile [key][d] and V^T tile [d][key] ----
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPR, ch = idx % CPR;
      int kr = kv0 + row;
      kr = kr < L ? kr : L - 1;
      uint4 kv = *reinterpret_cast<const uint4*>(kbase + (long)(b * L + kr) * p.ld_qk_b + ch * 16);
      uint4 vv = *reinterpret_cast<const uint4*>(vbase + ((long)row * p.Lpad + kv0) * ES + ch * 16);
      *reinterpret_cast<uint4*>(Ks + row * RS + ch * 16) = kv;
      *reinterpret_cast<uint4*>(Vts + row * RS + ch * 16) = vv;
    }
    __syncthreads();

    // ---- S = Q K^T (16 queries x 64 keys per wave) ----
    f32x4_t s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typename Atom<T>::frag_t kf =
            Atom<T>::load(reinterpret_cast<const T*>(Ks + (j * 16 + l15) * RS + (ks * KM + lg * KCH) * ES));
        s[j] = Atom<T>::mma(qf[ks], kf, s[j]);
      }
    }

    // ---- scale, bias, mask, online softmax ----
    float mx[4], rs[4], alpha[4];
    // relative-position bias: ONE wave-uniform branch, 16 unconditional loads from clamped indices (a
    // predicated load per element would serialise into 16 L2 round trips per tile)
    float bv[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[r][j] = 0.f;
    if (p.bias) {
      const float* bh = p.bias + (long)h * (2 * L - 1) + (L - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qg = q0w + lg * 4 + r;
        const int qgc = qg < L ? qg : L - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kg = kv0 + j * 16 + l15;
          bv[r][j] = bh[(kg < L ? kg : L - 1) - qgc];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int qg = q0w + lg * 4 + r;
      const int qgc = qg < L ? qg : L - 1;
      float mxr = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kg = kv0 + j * 16 + l15;
        float v = s[j][r] * p.scale + bv[r][j];
        const int rel = kg - qgc;
        bool ok = kg < L;
        if (p.band > 0) ok = ok && (rel >= -(p.band - 1)) && (rel <= p.band);
        v = ok ? v : -INFINITY;
        s[j][r] = v;
        mxr = fmaxf(mxr, v);
      }
#pragma unroll
      for (int o2 = 8; o2 > 0; o2 >>= 1) mxr = fmaxf(mxr, __shfl_xor(mxr, o2, 64));
      mx[r] = mxr;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mn = fmaxf(m[r], mx[r]);
      alpha[r] = fexp<T>(m[r] - mn);
      m[r] = mn;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = fexp<T>(s[j][r] - mn);
        s[j][r] = pv;
        sum += pv;
      }
#pragma unroll
      for (int o2 = 8; o2 > 0; o2 >>= 1) sum += __shfl_xor(sum, o2, 64);
      rs[r] = sum;
      l[r] = l[r] * alpha[r] + rs[r];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j][r] *= alpha[r];
    }
    // ---- P (C layout) -> wave-private LDS patch (A layout source) ----
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<T*>(Pw + (lg * 4 + r) * RS + (j * 16 + l15) * ES) = Elem<T>::from_f32(s[j][r]);
    __syncthreads();

    // ---- O += P V ----
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      typename Atom<T>::frag_t pf =
          Atom<T>::load(reinterpret_cast<const T*>(Pw + l15 * RS + (ks * KM + lg * KCH) * ES));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typename Atom<T>::frag_t vf =
            Atom<T>::load(reinterpret_cast<const T*>(Vts + (j * 16 + l15) * RS + (ks * KM + lg * KCH) * ES));
        o[j] = Atom<T>::mma(pf, vf, o[j]);
      }
    }
    __syncthreads();
  }

  // ---- normalise and store ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qg = q0w + lg * 4 + r;
    if (qg >= L) continue;
    const float inv = 1.0f / l[r];
    T* op = reinterpret_cast<T*>(p.out + (long)(b * L + qg) * p.ld_out_b) + h * 64 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) op[j * 16] = Elem<T>::from_f32(o[j][r] * inv);
  }
}

}  // namespace

int attention(const void* qk, int ld_qk, int k_col0, const void* vt, int Lpad, const float* bias, void* out,
              int ld_out, int B, int L, int H, float scale, int band, int dtype, hipStream_t s) {
  MH_REQUIRE(qk && vt && out, "mh_attention: null operand");
  MH_REQUIRE(B > 0 && L > 0 && H > 0, "mh_attention: bad shape");
  MH_REQUIRE(Lpad % 64 == 0 && Lpad >= L, "mh_attention: Lpad=%d must be a multiple of 64 and >= L=%d", Lpad, L);
  const int es = dtype == MH_BF16 ? 2 : 4;
  MH_REQUIRE((ld_qk * es) % 16 == 0 && (k_col0 * es) % 16 == 0, "mh_attention: rows must be 16-byte aligned");
  AttnP p;
  p.qk = (const char*)qk; p.ld_qk_b = (long)ld_qk * es; p.k_col0 = k_col0;
  p.vt = (const char*)vt; p.Lpad = Lpad; p.bias = bias;
  p.out = (char*)out; p.ld_out_b = (long)ld_out * es;
  p.B = B; p.L = L; p.H = H; p.scale = scale; p.band = band;
  dim3 grid(ceil_div(L, 64), H, B), block(256);
  const size_t smem = (size_t)(128 + 64) * (64 * es + 16);
  if (dtype == MH_BF16)
    hipLaunchKernelGGL(flash_attn_kernel<bf16_t>, grid, block, smem, s, p);
  else
    hipLaunchKernelGGL(flash_attn_kernel<float>, grid, block, smem, s, p);
  return check_launch("flash_attn_kernel");
}

}  // namespace mh

extern "C" int mh_attention(const void* qk, int ld_qk, int k_col0, const void* vt, int Lpad, const float* bias,
                            void* out, int ld_out, int B, int L, int H, float scale, int band, int dtype,
                            void* stream) {
  return mh::attention(qk, ld_qk, k_col0, vt, Lpad, bias, out, ld_out, B, L, H, scale, band, dtype,
                       (hipStream_t)stream);
}
