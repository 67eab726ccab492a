// Micro-benchmark: what does each part of the decode GEMV (csrc/decode_kernels.hpp: gemv_kernel) cost?
// Replays a graph of 72 dependent launches of the REAL kernel at the three decode shapes of osuT5-base (o / co
// projection, wi + gated GELU, wo) on 16 rows, alone and as two concurrent chains, and prints microseconds per kernel.
// Built several times with -DMH_GEMV_PROBE=<mask> (parts of the kernel switched off: 1 activation loads, 2 weight
// loads, 4 old residual values, 8 cross-wave reduction, 16 stores) to price the parts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <thread>

#include "../../mapperatorinator_amd/csrc/decode_kernels.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

namespace mh {   // symbols the header expects from the library
void set_error(const char*, ...) {}
}

using namespace mh;
using namespace mh::dec;

struct Bufs { bf16_t* attn; float* h; bf16_t* ff; bf16_t* W; float* lnw; };
static const int D = 768, DFF = 2048, B = 16;
static const long kLayerEl = 8L << 20;   // elements per layer of "weights" (16 MB)

template <int WHICH>
static void enqueue(hipStream_t s, const Bufs& b, int i, int nv) {
  const bf16_t* W = b.W + (long)(i % 12) * kLayerEl;
  SkinnyP p{};
  p.B = B; p.ln_w = b.lnw; p.eps = 1e-6f;
  if (WHICH == 0) {          // o / co projection + residual: N = 768, K = 768
    p.A = b.attn; p.lda = D; p.W = W; p.ldw = D; p.N = D; p.K = D; p.h = b.h; p.ldh = D; p.nv = nv;
    hipLaunchKernelGGL((gemv_kernel<bf16_t, 1, 4, PRO_PLAIN, SK_RESID>), dim3((D + nv - 1) / nv), dim3(256), 0, s, MH_GEMV_LEAD_ARGS(p), p, DepP{});
  } else if (WHICH == 1) {   // RMSNorm + wi + gated GELU: N = 2 * 2048, K = 768
    p.A = b.h; p.lda = D; p.W = W; p.ldw = D; p.N = 2 * DFF; p.K = D; p.out = b.ff; p.ldo = DFF; p.nv = 16;
    hipLaunchKernelGGL((gemv_kernel<bf16_t, 1, 4, PRO_RMSNORM, SK_GEGLU>), dim3(DFF / 8), dim3(256), 0, s, MH_GEMV_LEAD_ARGS(p), p, DepP{});
  } else {                   // wo + residual: N = 768, K = 2048 (8 waves)
    p.A = b.ff; p.lda = DFF; p.W = W; p.ldw = DFF; p.N = D; p.K = DFF; p.h = b.h; p.ldh = D; p.nv = nv;
    hipLaunchKernelGGL((gemv_kernel<bf16_t, 1, 8, PRO_PLAIN, SK_RESID>), dim3((D + nv - 1) / nv), dim3(512), 0, s, MH_GEMV_LEAD_ARGS(p), p, DepP{});
  }
}

typedef void (*Fn)(hipStream_t, const Bufs&, int, int);

static double run_chain(Fn fn, hipStream_t s, const Bufs& b, int nv, int reps) {
  const int chain = 72;
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < chain; ++i) fn(s, b, i, nv);
  CHECK(hipStreamEndCapture(s, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(ge, s));
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(ge, s));
  CHECK(hipEventRecord(e1, s));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  return ms * 1e3 / ((double)reps * chain);
}

int main() {
  Bufs b[2];
  bf16_t* W; CHECK(hipMalloc(&W, (size_t)13 * kLayerEl * 2)); CHECK(hipMemset(W, 0, (size_t)13 * kLayerEl * 2));
  hipStream_t s[2];
  for (int c = 0; c < 2; ++c) {
    b[c].W = W;
    CHECK(hipMalloc(&b[c].attn, B * D * 2)); CHECK(hipMemset(b[c].attn, 0, B * D * 2));
    CHECK(hipMalloc(&b[c].h, B * D * 4)); CHECK(hipMemset(b[c].h, 0, B * D * 4));
    CHECK(hipMalloc(&b[c].ff, B * DFF * 2)); CHECK(hipMemset(b[c].ff, 0, B * DFF * 2));
    CHECK(hipMalloc(&b[c].lnw, 1024 * 4)); CHECK(hipMemset(b[c].lnw, 0, 1024 * 4));
    CHECK(hipStreamCreateWithFlags(&s[c], hipStreamNonBlocking));
  }
  struct Case { const char* name; Fn fn; int nv; };
  const Case cases[] = {{"o-proj nv=4 (192 WG)", enqueue<0>, 4}, {"o-proj nv=8 (96 WG)", enqueue<0>, 8},
                        {"o-proj nv=16 (48 WG)", enqueue<0>, 16}, {"wi GEGLU (256 WG)", enqueue<1>, 16},
                        {"wo nv=4 (192 WG, 8 waves)", enqueue<2>, 4}, {"wo nv=16 (48 WG, 8 waves)", enqueue<2>, 16}};
  printf("MH_GEMV_PROBE = %d\n", MH_GEMV_PROBE);
  for (const Case& c : cases) {
    const double alone = run_chain(c.fn, s[0], b[0], c.nv, 40);
    double r[2];
    std::thread t([&] { r[1] = run_chain(c.fn, s[1], b[1], c.nv, 40); });
    r[0] = run_chain(c.fn, s[0], b[0], c.nv, 40);
    t.join();
    printf("  %-28s alone %5.2f us   two chains %5.2f / %5.2f us\n", c.name, alone, r[0], r[1]);
  }
  return 0;
}
