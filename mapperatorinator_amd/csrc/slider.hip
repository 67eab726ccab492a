// Slider end re-projection inside the DDPM loop (reference diffusion_pipeline.py:201-222 `denoised_fn`): after the in-paint
// `where`, every slider whose control points and end lie inside the window gets its END point moved onto its own path at
// the slider's length -- SliderPath(curve_type, control points).position_at(length / get_distance())
// (osuT5/osuT5/inference/slider_path.py:82-230, path_approximator.py).  The reference does this on the host in numpy every
// denoising step; here it is one kernel between the two halves of the DDPM step, so the loop stays one replayed hipGraph.
//
// One workgroup per song chunk (the CFG pair rows b and b + pair share one chunk).  The chunk's positions are staged in LDS
// in playfield pixels; wave w walks sliders w, w + 4, ... of the chunk with lane 0 (the path algorithms are short serial
// recurrences over <= a few hundred points; the other lanes only help with the staging).  The path is never stored: it is
// generated twice, once to measure its length and once to find the segment that holds the target distance.
//
// Arithmetic follows the reference's numpy dtype flow on float32 control points (numpy >= 2 promotion rules: python
// scalars are weak): Linear / Catmull / circular-arc paths and their lengths in float32, Bezier paths in float64 with the
// reference's buffer reuse restated (the first Bezier buffer is the float32 copy of the control points and is recycled
// through `free_buffers`, so some sub-curves are rounded to float32 -- the `f32` flag below).
#include "internal.hpp"

#pragma clang fp contract(off)

namespace mh {
namespace {

constexpr int SL_MAXCP = 32;    // control points of ONE Bezier span (spans split at repeated points)
constexpr int SL_DEPTH = 16;    // subdivision stack; second differences shrink 4x per level, 16 levels are never reached
constexpr int SL_WAVES = 4;

struct SlWork {                 // per wave, LDS
  double stack[SL_DEPTH][SL_MAXCP][2];
  double mid[SL_MAXCP][2];      // subdivision_buffer1 (also the right half, path_approximator.py:229-231)
  double left[2 * SL_MAXCP][2]; // subdivision_buffer2
  unsigned char sflag[SL_DEPTH];
  unsigned char freef[2 * SL_DEPTH];
};

// A numpy scalar as the reference's arithmetic sees it: the value and whether its dtype is float32.  Two float32 operands
// give a float32 result computed in float32, anything else promotes to float64 (numpy >= 2: python scalars are weak).
struct TN {
  double v; bool s;
};
__device__ inline TN tn_add(TN a, TN b) { return (a.s && b.s) ? TN{(double)((float)a.v + (float)b.v), true} : TN{a.v + b.v, false}; }
__device__ inline TN tn_sub(TN a, TN b) { return (a.s && b.s) ? TN{(double)((float)a.v - (float)b.v), true} : TN{a.v - b.v, false}; }
__device__ inline TN tn_mul(TN a, TN b) { return (a.s && b.s) ? TN{(double)((float)a.v * (float)b.v), true} : TN{a.v * b.v, false}; }
__device__ inline TN tn_div(TN a, TN b) { return (a.s && b.s) ? TN{(double)((float)a.v / (float)b.v), true} : TN{a.v / b.v, false}; }
__device__ inline TN tn_sqrt(TN a) { return a.s ? TN{(double)sqrtf((float)a.v), true} : TN{sqrt(a.v), false}; }

// consumer of path points: pass 0 measures the length, pass 1 finds position_at(target).  `s` of a point = its span's
// array dtype is float32 (Linear / Catmull / arc points, 1- and 2-point Bezier spans); float64 otherwise.
struct Walk {
  int pass;
  bool have, done, ps;
  double px, py, rx, ry;
  TN cum, target;
  __device__ void begin(int p, TN t) { pass = p; have = false; done = false; cum = TN{0.0, true}; target = t; rx = ry = px = py = 0; ps = true; }
  // slider_path.py:121-143 (drop repeated points), :145-180 (cumulative length), :182-228 (index_of_distance,
  // interpolate_vertices): i = first vertex with cumulative length >= target
  __device__ void emit(double x, double y, bool s) {
    if (have && px == x && py == y) return;
    if (!have) {
      have = true;
      if (pass == 1 && !(cum.v < target.v)) { rx = x; ry = y; done = true; }
      px = x; py = y; ps = s;
      return;
    }
    const bool ds = s && ps;                                   // dtype of path[i + 1] - path[i]
    const TN dx = tn_sub(TN{x, ds}, TN{px, ds}), dy = tn_sub(TN{y, ds}, TN{py, ds});
    const TN d = tn_sqrt(tn_add(tn_mul(dx, dx), tn_mul(dy, dy)));
    const TN nc = tn_add(cum, d);
    if (pass == 1 && !done && nc.v >= target.v) {
      if (fabs(cum.v - nc.v) <= 1e-8 + 1e-5 * fabs(nc.v)) { rx = px; ry = py; }   // np.isclose(d0, d1) -> p0
      else {
        const TN w = tn_div(tn_sub(target, cum), tn_sub(nc, cum));
        rx = tn_add(TN{px, ps}, tn_mul(dx, w)).v;
        ry = tn_add(TN{py, ps}, tn_mul(dy, w)).v;
      }
      done = true;
    }
    cum = nc; px = x; py = y; ps = s;
  }
  __device__ void finish() { if (pass == 1 && !done) { rx = px; ry = py; done = true; } }
};

struct Pts {                    // control points of one slider: pixel positions in LDS, addressed through the index list
  const float* x; const float* y; const int* idx;
  __device__ float X(int i) const { return x[idx[i]]; }
  __device__ float Y(int i) const { return y[idx[i]]; }
};

// ---- path_approximator.py:198-247 ----------------------------------------------------------------------------------
__device__ bool bezier_flat(const double (*cp)[2], int count, bool f32) {
  for (int i = 1; i < count - 1; ++i) {
    if (f32) {
      const float ax = (float)cp[i - 1][0] - 2.0f * (float)cp[i][0] + (float)cp[i + 1][0];
      const float ay = (float)cp[i - 1][1] - 2.0f * (float)cp[i][1] + (float)cp[i + 1][1];
      if (ax * ax + ay * ay > 0.25f) return false;
    } else {
      const double ax = cp[i - 1][0] - 2.0 * cp[i][0] + cp[i + 1][0];
      const double ay = cp[i - 1][1] - 2.0 * cp[i][1] + cp[i + 1][1];
      if (ax * ax + ay * ay > 0.25) return false;
    }
  }
  return true;
}

// left[0..count) and mid[0..count) (= right) from cp
__device__ void bezier_subdivide(const double (*cp)[2], SlWork& w, int count) {
  for (int i = 0; i < count; ++i) { w.mid[i][0] = cp[i][0]; w.mid[i][1] = cp[i][1]; }
  for (int i = 0; i < count; ++i) {
    w.left[i][0] = w.mid[0][0]; w.left[i][1] = w.mid[0][1];
    for (int j = 0; j < count - i - 1; ++j) {
      w.mid[j][0] = (w.mid[j][0] + w.mid[j + 1][0]) / 2;
      w.mid[j][1] = (w.mid[j][1] + w.mid[j + 1][1]) / 2;
    }
  }
}

// approximate_b_spline with p = 0 (path_approximator.py:16-88) over control points [i0, i0 + count)
__device__ void bezier_span(SlWork& w, const Pts& P, int i0, int count, Walk& walk) {
  if (count <= 0) return;
  if (count > SL_MAXCP) {          // refused by the host wrapper; a raw C caller gets NaN positions, never a quiet overrun
    walk.emit(__builtin_nan(""), __builtin_nan(""), false);
    walk.emit(0.0, 0.0, false);
    return;
  }
  const bool s32 = count <= 2;     // np.vstack: float64 as soon as the span has an interior (float64) point
  for (int i = 0; i < count; ++i) { w.stack[0][i][0] = P.X(i0 + i); w.stack[0][i][1] = P.Y(i0 + i); }
  w.sflag[0] = 1;
  int sp = 1, nfree = 0;
  while (sp > 0) {
    --sp;
    double (*parent)[2] = w.stack[sp];
    const bool pf = w.sflag[sp];
    if (sp + 2 > SL_DEPTH || bezier_flat(parent, count, pf)) {
      bezier_subdivide(parent, w, count);
      for (int i = 0; i < count - 1; ++i) { w.left[count + i][0] = w.mid[i + 1][0]; w.left[count + i][1] = w.mid[i + 1][1]; }
      walk.emit(parent[0][0], parent[0][1], s32);
      for (int i = 1; i < count - 1; ++i) {
        const int k = 2 * i;
        walk.emit(0.25 * (w.left[k - 1][0] + 2 * w.left[k][0] + w.left[k + 1][0]),
                  0.25 * (w.left[k - 1][1] + 2 * w.left[k][1] + w.left[k + 1][1]), false);
      }
      if (nfree < 2 * SL_DEPTH) w.freef[nfree++] = pf;
      continue;
    }
    const bool rf = nfree > 0 ? w.freef[--nfree] : false;
    bezier_subdivide(parent, w, count);
    // to_flatten.append(right_child); to_flatten.append(parent = left half)
    double (*right)[2] = w.stack[sp];
    double (*lchild)[2] = w.stack[sp + 1];
    for (int i = 0; i < count; ++i) {
      const double rx = w.mid[i][0], ry = w.mid[i][1], lx = w.left[i][0], ly = w.left[i][1];
      right[i][0] = rf ? (double)(float)rx : rx; right[i][1] = rf ? (double)(float)ry : ry;
      lchild[i][0] = pf ? (double)(float)lx : lx; lchild[i][1] = pf ? (double)(float)ly : ly;
    }
    w.sflag[sp] = rf; w.sflag[sp + 1] = pf;
    sp += 2;
  }
  walk.emit((double)P.X(i0 + count - 1), (double)P.Y(i0 + count - 1), s32);
}

// ---- path_approximator.py:91-105, 250-281: float32 throughout, t / t^2 / t^3 rounded from the python floats ----------
__device__ void catmull_point(float x1, float x2, float x3, float x4, float y1, float y2, float y3, float y4, double t,
                              float& ox, float& oy) {
  const float tf = (float)t, t2 = (float)(t * t), t3 = (float)(t * (t * t));
  ox = 0.5f * (2.0f * x2 + (-x1 + x3) * tf + (2.0f * x1 - 5.0f * x2 + 4.0f * x3 - x4) * t2 + (-x1 + 3.0f * x2 - 3.0f * x3 + x4) * t3);
  oy = 0.5f * (2.0f * y2 + (-y1 + y3) * tf + (2.0f * y1 - 5.0f * y2 + 4.0f * y3 - y4) * t2 + (-y1 + 3.0f * y2 - 3.0f * y3 + y4) * t3);
}

__device__ void catmull_span(const Pts& P, int i0, int n, Walk& walk) {
  for (int i = 0; i < n - 1; ++i) {
    const float x2 = P.X(i0 + i), y2 = P.Y(i0 + i);
    const float x1 = i > 0 ? P.X(i0 + i - 1) : x2, y1 = i > 0 ? P.Y(i0 + i - 1) : y2;
    const float x3 = P.X(i0 + i + 1), y3 = P.Y(i0 + i + 1);
    const float x4 = i < n - 2 ? P.X(i0 + i + 2) : x3 + x3 - x2, y4 = i < n - 2 ? P.Y(i0 + i + 2) : y3 + y3 - y2;
    for (int c = 0; c < 50; ++c) {
      float ax, ay, bx, by;
      catmull_point(x1, x2, x3, x4, y1, y2, y3, y4, (double)c / 50.0, ax, ay);
      catmull_point(x1, x2, x3, x4, y1, y2, y3, y4, (double)(c + 1) / 50.0, bx, by);
      walk.emit(ax, ay, true);
      walk.emit(bx, by, true);
    }
  }
}

// ---- path_approximator.py:108-176 --------------------------------------------------------------------------------------
struct Arc { float cx, cy, r, theta0, range; int dir, n; };

__device__ bool arc_setup(const Pts& P, Arc& a) {
  const float ax = P.X(0), ay = P.Y(0), bx = P.X(1), by = P.Y(1), cx = P.X(2), cy = P.Y(2);
  const float aSq = (bx - cx) * (bx - cx) + (by - cy) * (by - cy);
  const float bSq = (ax - cx) * (ax - cx) + (ay - cy) * (ay - cy);
  const float cSq = (ax - bx) * (ax - bx) + (ay - by) * (ay - by);
  if (fabsf(aSq) <= 1e-8f || fabsf(bSq) <= 1e-8f || fabsf(cSq) <= 1e-8f) return false;
  const float s = aSq * (bSq + cSq - aSq), t = bSq * (aSq + cSq - bSq), u = cSq * (aSq + bSq - cSq);
  const float sum = s + t + u;
  if (fabsf(sum) <= 1e-8f) return false;
  a.cx = (s * ax + t * bx + u * cx) / sum;
  a.cy = (s * ay + t * by + u * cy) / sum;
  const float dAx = ax - a.cx, dAy = ay - a.cy, dCx = cx - a.cx, dCy = cy - a.cy;
  a.r = sqrtf(dAx * dAx + dAy * dAy);
  const float two_pi = (float)(2.0 * 3.141592653589793);
  a.theta0 = atan2f(dAy, dAx);
  float theta_end = atan2f(dCy, dCx);
  while (theta_end < a.theta0) theta_end += two_pi;
  a.dir = 1;
  a.range = theta_end - a.theta0;
  const float ox = cy - ay, oy = -(cx - ax);
  if (ox * (bx - ax) + oy * (by - ay) < 0.0f) { a.dir = -1; a.range = two_pi - a.range; }
  if (2.0f * a.r <= 0.1f) a.n = 2;
  else {
    const float q = ceilf(a.range / (2.0f * acosf(1.0f - 0.1f / a.r)));
    a.n = q > 2.0f ? (q < 1.0e6f ? (int)q : 1000000) : 2;    // NaN (acos of a rounding overshoot) -> 2, like max(2, nan)
  }
  return true;
}

__device__ void arc_path(const Arc& a, Walk& walk) {
  for (int i = 0; i < a.n; ++i) {
    const float fr = (float)((double)a.dir * ((double)i / (double)(a.n - 1)));
    const float theta = a.theta0 + fr * a.range;
    walk.emit(a.cx + cosf(theta) * a.r, a.cy + sinf(theta) * a.r, true);
  }
}

// SliderPath.calculate_path (slider_path.py:121-143): spans end where a control point repeats
template <typename SpanFn> __device__ void for_spans(const Pts& P, int ncp, Walk& walk, SpanFn span) {
  int start = 0;
  for (int i = 0; i < ncp; ++i) {
    if (i == ncp - 1 || (P.X(i) == P.X(i + 1) && P.Y(i) == P.Y(i + 1))) {
      span(start, i + 1 - start, walk);
      start = i + 1;
    }
  }
}

enum { SL_LINEAR = 0, SL_PERFECT = 1, SL_CATMULL = 2, SL_BEZIER = 3 };

// returns false when the path has no length (the reference skips the slider)
__device__ bool slider_end(SlWork& w, const Pts& P, int ncp, int type, double length, float& ex, float& ey) {
  Arc arc;
  bool use_arc = false;
  if (type == SL_PERFECT && ncp == 3) {
    const bool rep = (P.X(0) == P.X(1) && P.Y(0) == P.Y(1)) || (P.X(1) == P.X(2) && P.Y(1) == P.Y(2));
    use_arc = !rep && arc_setup(P, arc);
  }
  Walk walk;
  TN target{0.0, true};
  for (int pass = 0; pass < 2; ++pass) {
    walk.begin(pass, target);
    if (use_arc) arc_path(arc, walk);
    else if (type == SL_LINEAR) { for (int i = 0; i < ncp; ++i) walk.emit(P.X(i), P.Y(i), true); }
    else if (type == SL_CATMULL) for_spans(P, ncp, walk, [&](int i0, int n, Walk& wk) { catmull_span(P, i0, n, wk); });
    else for_spans(P, ncp, walk, [&](int i0, int n, Walk& wk) { bezier_span(w, P, i0, n, wk); });
    walk.finish();
    if (pass == 0) {
      const TN total = walk.cum;
      if (!walk.have || total.v == 0.0) return false;
      // np.clip(length / total, 0, 1) * total: the python float `length` takes the dtype of `total`
      TN prog = tn_div(TN{total.s ? (double)(float)length : length, total.s}, total);
      prog.v = prog.v < 0.0 ? 0.0 : (prog.v > 1.0 ? 1.0 : prog.v);
      target = tn_mul(prog, total);
    }
  }
  ex = (float)walk.rx; ey = (float)walk.ry;
  return true;
}

__global__ __launch_bounds__(SL_WAVES * 64) void slider_project_kernel(float* __restrict__ x0, const uint8_t* __restrict__ imask,
                                                                       const float* __restrict__ iref, int T, int pair,
                                                                       MhSliderSet ss) {
  extern __shared__ float pos[];            // [2][T] playfield pixels of row b
  __shared__ SlWork work[SL_WAVES];
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool active = ss.chunk_active[b] != 0;
  const int nrows = pair > 0 ? 2 : 1;
  for (int r = 0; r < nrows; ++r) {
    const long base = ((long)b + (long)r * pair) * 2 * T;
    for (int i = tid; i < 2 * T; i += SL_WAVES * 64) {
      float v = x0[base + i];
      if (imask) v = imask[base + i] ? v : iref[base + i];
      if (!active) x0[base + i] = v;
      else if (r == 0) pos[i] = ((v + 1.0f) / 2.0f) * (i < T ? 512.0f : 384.0f);   // to_positions (:171-177)
    }
  }
  if (!active) return;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) {
    for (int s = ss.chunk_off[b] + wave; s < ss.chunk_off[b + 1]; s += SL_WAVES) {
      const int o = ss.cp_off[s], ncp = ss.cp_off[s + 1] - o;
      Pts P{pos, pos + T, ss.cp_idx + o};
      float ex, ey;
      if (slider_end(work[wave], P, ncp, ss.type[s], ss.length[s], ex, ey)) {
        // distinct sliders touch distinct points (the host checks), so no other wave reads or writes this one
        pos[ss.end_idx[s]] = ex;
        pos[T + ss.end_idx[s]] = ey;
      }
    }
  }
  __syncthreads();
  // x[:, :, :] = positions / (512, 384) * 2 - 1, broadcast over the CFG pair (:220)
  for (int i = tid; i < 2 * T; i += SL_WAVES * 64) {
    const float v = pos[i] / (i < T ? 512.0f : 384.0f) * 2.0f - 1.0f;
    x0[(long)b * 2 * T + i] = v;
    if (pair > 0) x0[((long)b + pair) * 2 * T + i] = v;
  }
}

}  // namespace

int slider_project(float* x0, const uint8_t* imask, const float* iref, int N, int T, const MhSliderSet& ss, hipStream_t s) {
  const int pair = ss.pair_stride;
  const int chunks = pair > 0 ? pair : N;
  hipLaunchKernelGGL(slider_project_kernel, dim3(chunks), dim3(SL_WAVES * 64), (size_t)2 * T * sizeof(float), s, x0, imask,
                     iref, T, pair, ss);
  return check_launch("slider_project_kernel");
}

int check_slider_set(const MhSliderSet* ss, int N) {
  MH_REQUIRE(ss->n_chunks > 0 && ss->n_sliders >= 0 && ss->chunk_active && ss->chunk_off, "slider set: empty");
  MH_REQUIRE(ss->pair_stride == 0 ? ss->n_chunks == N : (ss->pair_stride == ss->n_chunks && N == 2 * ss->n_chunks),
             "slider set: n_chunks / pair_stride do not describe N = %d rows", N);
  MH_REQUIRE(ss->n_sliders == 0 || (ss->type && ss->cp_off && ss->cp_idx && ss->end_idx && ss->length),
             "slider set: null slider array");
  return MH_OK;
}

}  // namespace mh

using namespace mh;

extern "C" int mh_slider_project(float* x0, const uint8_t* inpaint_mask, const float* inpaint_ref, int N, int T,
                                 const MhSliderSet* sliders, void* stream) {
  MH_REQUIRE(x0 && sliders && N > 0 && T > 0, "mh_slider_project: bad argument");
  MH_REQUIRE((inpaint_mask == nullptr) == (inpaint_ref == nullptr), "mh_slider_project: inpaint mask/ref must come together");
  if (int rc = check_slider_set(sliders, N)) return rc;
  return slider_project(x0, inpaint_mask, inpaint_ref, N, T, *sliders, (hipStream_t)stream);
}
