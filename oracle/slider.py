"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the slider end re-projection that the reference's diffusion
`denoised_fn` runs every denoising step.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module; it is the checker,
never the product.  It follows

  denoised_fn                  diffusion_pipeline.py:201-222 (in-paint `where`, pixel round trip, per-slider end update,
                               broadcast of row 0 over the CFG pair)
  SliderPath                   osuT5/osuT5/inference/slider_path.py:26-230 (spans split at repeated control points,
                               repeated path points dropped, cumulative length, position_at = binary search + lerp)
  approximate_bezier           osuT5/osuT5/inference/path_approximator.py:12-88 (b-spline degree 0 = plain Bezier:
                               adaptive de Casteljau subdivision, flatness 0.25 px), :198-247
  approximate_catmull          :91-105, :250-281 (50 samples per segment)
  approximate_circular_arc     :108-176 (tolerance 0.1 px)
  approximate_linear           :179-185

numpy dtype flow (numpy >= 2, python scalars are weak): the control points arrive float32 (`to_positions(x)...numpy()`),
so Linear / Catmull / arc paths, their lengths and the interpolation stay float32, while a Bezier span of three or more
points is float64 (np.vstack with its float64 interior points); 1- and 2-point Bezier spans stay float32, and a path that
mixes both promotes per operation (float32 until the first float64 operand arrives).  The
reference recycles its Bezier buffers: the first one is the float32 copy of the control points, it is re-filled with the
left halves and later handed out again through `free_buffers`, so part of the subdivision tree is rounded to float32.
That is restated here with a per-buffer dtype instead of buffer aliasing.

PINNING: tests/test_oracle_pinned.py runs this against the imported reference modules on random sliders of every curve
type (bit-equal end points), and tests/golden/sliders.npz holds reference outputs for the GPU box.
"""
from __future__ import annotations

import numpy as np
import torch

PLAYFIELD = (512, 384)
LINEAR, PERFECT, CATMULL, BEZIER = 0, 1, 2, 3
CURVE_CODE = {"Linear": LINEAR, "PerfectCurve": PERFECT, "Catmull": CATMULL, "Bezier": BEZIER}


# ---- Bezier ------------------------------------------------------------------------------------
def _flat(buf):
    """bezier_is_flat_enough: every second difference shorter than 2 * 0.25 px (in the buffer's own dtype)"""
    for i in range(1, len(buf) - 1):
        p = buf[i - 1] - 2 * buf[i] + buf[i + 1]
        if np.inner(p, p) > 0.25 * 0.25 * 4:
            return False
    return True


def _halves(buf):
    """de Casteljau split at 1/2 in float64 (the reference's scratch buffers are np.empty = float64)"""
    n = len(buf)
    mid = np.array(buf, dtype=np.float64)
    left = np.empty((n, 2))
    for i in range(n):
        left[i] = mid[0]
        for j in range(n - i - 1):
            mid[j] = (mid[j] + mid[j + 1]) / 2
    return left, mid


def bezier_points(cps):
    """cps (n, 2) float32 -> path points of one span; `np.vstack` gives them ONE dtype: float64 as soon as an interior
    point exists (n >= 3), float32 for the 1- and 2-point spans"""
    n = len(cps)
    if n == 0:
        return []
    out = []
    stack = [cps.copy()]            # dtype of an entry = dtype of the reference buffer that holds it
    free = []                       # dtypes of the recycled buffers
    while stack:
        parent = stack.pop()
        if _flat(parent):
            left, right = _halves(parent)
            poly = np.concatenate([left, right[1:]])
            out.append(parent[0].copy())
            for i in range(1, n - 1):
                k = 2 * i
                out.append(0.25 * (poly[k - 1] + 2 * poly[k] + poly[k + 1]))
            free.append(parent.dtype)
            continue
        rdtype = free.pop() if free else np.dtype(np.float64)
        left, right = _halves(parent)
        stack.append(right.astype(rdtype))
        stack.append(left.astype(parent.dtype))
    out.append(cps[n - 1].copy())
    return list(np.vstack(out))


# ---- Catmull -------------------------------------------------------------------------------------
def _catmull(v1, v2, v3, v4, t):
    t2 = t * t
    t3 = t * t2
    return np.array([0.5 * (2 * v2[k] + (-v1[k] + v3[k]) * t + (2 * v1[k] - 5 * v2[k] + 4 * v3[k] - v4[k]) * t2
                            + (-v1[k] + 3 * v2[k] - 3 * v3[k] + v4[k]) * t3) for k in range(2)])


def catmull_points(cps):
    out = []
    n = len(cps)
    for i in range(n - 1):
        v2 = cps[i]
        v1 = cps[i - 1] if i > 0 else v2
        v3 = cps[i + 1]
        v4 = cps[i + 2] if i < n - 2 else v3 + v3 - v2
        for c in range(50):
            out.append(_catmull(v1, v2, v3, v4, c / 50))
            out.append(_catmull(v1, v2, v3, v4, (c + 1) / 50))
    return out


# ---- circular arc --------------------------------------------------------------------------------
def arc_points(cps):
    a, b, c = cps[0], cps[1], cps[2]
    a_sq, b_sq, c_sq = np.inner(b - c, b - c), np.inner(a - c, a - c), np.inner(a - b, a - b)
    if np.isclose(a_sq, 0) or np.isclose(b_sq, 0) or np.isclose(c_sq, 0):
        return []
    s, t, u = a_sq * (b_sq + c_sq - a_sq), b_sq * (a_sq + c_sq - b_sq), c_sq * (a_sq + b_sq - c_sq)
    tot = s + t + u
    if np.isclose(tot, 0):
        return []
    centre = (s * a + t * b + u * c) / tot
    d_a, d_c = a - centre, c - centre
    r = np.linalg.norm(d_a)
    th0, th1 = np.arctan2(d_a[1], d_a[0]), np.arctan2(d_c[1], d_c[0])
    while th1 < th0:
        th1 += 2 * np.pi
    direction, rng = 1, th1 - th0
    chord = c - a
    if np.dot(np.array([chord[1], -chord[0]]), b - a) < 0:
        direction, rng = -1, 2 * np.pi - rng
    count = 2 if 2 * r <= 0.1 else int(max(2, np.ceil(rng / (2 * np.arccos(1 - 0.1 / r)))))
    out = []
    for i in range(count):
        th = th0 + direction * (i / (count - 1)) * rng
        out.append(centre + np.array([np.cos(th), np.sin(th)]) * r)
    return out


# ---- SliderPath -----------------------------------------------------------------------------------
def path_points(curve: int, cps):
    """calculate_path: per span (ending where a control point repeats) the curve's approximation, repeats dropped"""
    n = len(cps)
    path = []
    start = 0
    for i in range(n):
        if i == n - 1 or (cps[i] == cps[i + 1]).all():
            span = cps[start:i + 1]
            if curve == LINEAR:
                pts = [p.copy() for p in span]
            elif curve == PERFECT:
                pts = arc_points(span) if (n == 3 and len(span) == 3) else []
                if len(pts) == 0:
                    pts = bezier_points(span)
            elif curve == CATMULL:
                pts = catmull_points(span)
            else:
                pts = bezier_points(span)
            for p in pts:
                if not path or (path[-1] != p).any():
                    path.append(p)
            start = i + 1
    return path


def slider_end_position(curve: int, cps, length: float):
    """SliderPath(curve, cps).position_at(length / get_distance()); None when the path has no length"""
    path = path_points(curve, cps)
    cum = [0]
    for i in range(len(path) - 1):
        cum.append(cum[-1] + np.linalg.norm(path[i + 1] - path[i]))
    total = cum[-1]
    if total == 0:
        return None
    d = np.clip(length / total, 0, 1) * total
    i = next((k for k, v in enumerate(cum) if v >= d), len(cum))      # what binary_search + ~ resolve to (cum ascending)
    if i <= 0:
        return path[0]
    if i >= len(path):
        return path[-1]
    if np.isclose(cum[i - 1], cum[i]):
        return path[i - 1]
    w = (d - cum[i - 1]) / (cum[i] - cum[i - 1])
    return path[i - 1] + (path[i] - path[i - 1]) * w


def denoised_fn(x: torch.Tensor, mask: torch.Tensor, ref: torch.Tensor, sliders, start: int, end: int) -> torch.Tensor:
    """x (2, 2, T) CFG pair of one song.  `sliders`: objects with seq_indices / end_index / curve_type / length
    (DiffusionSlider, diffusion_pipeline.py:30-35), indices relative to the whole song."""
    x = torch.where(mask, x, ref)
    if len(sliders) == 0:
        return x
    half = x.clone().chunk(2, dim=0)[0]
    half += 1
    half /= 2
    half *= torch.tensor(PLAYFIELD).repeat(1, 1).unsqueeze(2)
    x2 = half.squeeze(0).T.numpy()
    for s in sliders:
        idx = np.asarray(s.seq_indices)
        if np.any((idx < start) | (idx >= end)) or s.end_index < start or s.end_index >= end:
            continue
        p = slider_end_position(CURVE_CODE.get(s.curve_type, BEZIER), x2[idx - start], s.length)
        if p is not None:
            x2[s.end_index - start] = p
    x[:, :, :] = torch.from_numpy(x2.T) / torch.tensor(PLAYFIELD).unsqueeze(1) * 2 - 1
    return x
