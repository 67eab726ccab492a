"""Host side of the osu_diffusion refinement stage: `DiTHIP` (boundary B3) and `SpacedDiffusionHIP`
(boundary B4), SURVEY.md 8b.

  DiTHIP.forward_with_cfg(x, t, c, y, cfg_scale, attn_mask=None, key_padding_mask=None)
      == DiT.forward_with_cfg            osu_diffusion/utils/models.py:301-317
  create_diffusion(...).p_sample_loop(model.forward_with_cfg, shape, noise, denoised_fn=..., ...)
      == SpacedDiffusion.p_sample_loop   utils/diffusion/gaussian_diffusion.py:469-561, respace.py:63-132

The schedule tables are derived in float64 on the host from the same published IDDPM formulas
(cosine alpha-bar schedule, respacing by re-deriving betas from the kept alpha-bars, learned-range
variance interpolation) and handed to the kernels as fp32 exactly the way `_extract_into_tensor`
casts them (gaussian_diffusion.py:951-963).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib
# osu_diffusion/utils/models.py:384-405 (depth, hidden, heads); "DiT-XS" is a test-only size
DIT_PRESETS = {
    "DiT-XS": (2, 128, 2),
    "DiT-S": (12, 384, 6),
    "DiT-B": (12, 768, 12),
    "DiT-L": (24, 1024, 16),
}


def _round_up(a, b):
    return (a + b - 1) // b * b


class DiTHIP:
    """fp32 DiT denoiser on libmapperhip.  `state_dict` uses the reference's parameter names."""

    def __init__(self, state_dict: dict, depth: int, hidden: int, num_heads: int, context_size: int = 272,
                 class_size: int = 300, device="cuda", operand_dtype: torch.dtype = torch.float32,
                 options: Optional[dict] = None):
        self.require_gpu()
        if hidden != num_heads * 64:
            raise NotImplementedError("HIP attention kernels are built for head_dim = 64")
        if operand_dtype not in (torch.float32, torch.bfloat16, "mx8"):
            raise ValueError("operand_dtype: torch.float32 (the reference's semantics), torch.bfloat16 (block GEMM operands "
                             "rounded to bf16, fp32 accumulation / residual stream / LayerNorm / softmax / DDPM update) or 'mx8' "
                             "(BASELINE configs[4]: the four block projections on MX-fp8 operands, the rest as the bf16 mode)")
        if operand_dtype == "mx8" and hidden % 128:
            raise ValueError("MX-fp8 operands need hidden % 128 == 0")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.operand_dtype = operand_dtype
        self.in_channels, self.learn_sigma = 2, True
        self.depth, self.hidden, self.num_heads = depth, hidden, num_heads
        self.context_size, self.class_size = context_size, class_size
        self._keep = []
        dev = self.device

        def t(x, kpad=None):
            x = x.detach().to(torch.float32)
            if kpad is not None and x.shape[-1] != kpad:
                x = torch.nn.functional.pad(x, (0, kpad - x.shape[-1]))
            x = x.contiguous().to(dev)
            self._keep.append(x)
            return x.data_ptr()

        def t3(x, kpad=None):
            """pre-split copy of a weight matrix for the bf16 x 3 GEMM path (MhGemm.w_split3): every 32-float block of
            a row becomes [32 x bf16 hi | 32 x bf16 lo], hi = bf16(w), lo = bf16(w - hi)"""
            x = x.detach().to(torch.float32)
            if kpad is not None and x.shape[-1] != kpad:
                x = torch.nn.functional.pad(x, (0, kpad - x.shape[-1]))
            n, k = x.shape
            if k % 32:
                return None
            hi = x.to(torch.bfloat16)
            lo = (x - hi.to(torch.float32)).to(torch.bfloat16)
            packed = torch.stack([hi.reshape(n, k // 32, 32), lo.reshape(n, k // 32, 32)], dim=2).reshape(n, 2 * k)
            packed = packed.contiguous().to(dev)
            self._keep.append(packed)
            return packed.data_ptr()

        k1 = 2 * 128 + context_size
        def tb(x):
            x = x.detach().to(torch.bfloat16).contiguous().to(dev)
            self._keep.append(x)
            return x.data_ptr()

        def tm(x):
            from .mx8 import quantize_mx8
            q, sc = quantize_mx8(x.detach().to(torch.float32).to(torch.bfloat16).to(torch.float32).to(dev))   # the bf16 mode's weights, quantised
            self._keep += [q, sc]
            return q.data_ptr(), sc.data_ptr()

        lowp = operand_dtype == torch.bfloat16
        mx = isinstance(operand_dtype, str) and operand_dtype == "mx8"
        cfg = _lib.MhDiTConfig(hidden, depth, num_heads, context_size, class_size, 2, 128, 256, _round_up(k1, 32),
                               class_size, _lib.MH_MX8 if mx else (_lib.MH_BF16 if lowp else _lib.MH_F32))
        w = _lib.MhDiTWeights()
        # frequency tables with the same fp32 tensor ops as timestep_embedding (positional_embedding.py:38-43)
        w.pos_freqs = t(torch.exp(-math.log(10000) * torch.arange(0, 64, dtype=torch.float32) / 64))
        w.t_freqs = t(torch.exp(-math.log(10000) * torch.arange(0, 128, dtype=torch.float32) / 128))
        sd = state_dict
        # a state dict of another size would make the kernels read past its tensors: refuse before anything is packed
        for key, want in (("y_embedder.class_embedding.0.weight", (hidden, class_size)),
                          ("context_embedder.mlp.0.weight", (hidden, 2 * 128 + context_size)),
                          ("blocks.0.attn.in_proj_weight", (3 * hidden, hidden))):
            if tuple(sd[key].shape) != want:
                raise ValueError(f"{key} is {tuple(sd[key].shape)}, the configuration (hidden {hidden}, context_size {context_size}, "
                                 f"class_size {class_size}) needs {want}")
        w.first_w, w.first_b = t(sd["context_embedder.mlp.0.weight"], cfg.first_k_pad), t(sd["context_embedder.mlp.0.bias"])
        w.first_w3 = t3(sd["context_embedder.mlp.0.weight"], cfg.first_k_pad)
        w.t_w0, w.t_b0 = t(sd["t_embedder.mlp.0.weight"]), t(sd["t_embedder.mlp.0.bias"])
        w.t_w1, w.t_b1 = t(sd["t_embedder.mlp.2.weight"]), t(sd["t_embedder.mlp.2.bias"])
        w.y_w0, w.y_b0 = t(sd["y_embedder.class_embedding.0.weight"]), t(sd["y_embedder.class_embedding.0.bias"])
        w.y_w1, w.y_b1 = t(sd["y_embedder.class_embedding.2.weight"]), t(sd["y_embedder.class_embedding.2.bias"])
        for l in range(depth):
            b = f"blocks.{l}."
            w.ada_w[l], w.ada_b[l] = t(sd[b + "adaLN_modulation.1.weight"]), t(sd[b + "adaLN_modulation.1.bias"])
            w.qkv_w[l], w.qkv_b[l] = t(sd[b + "attn.in_proj_weight"]), t(sd[b + "attn.in_proj_bias"])
            w.out_w[l], w.out_b[l] = t(sd[b + "attn.out_proj.weight"]), t(sd[b + "attn.out_proj.bias"])
            w.fc1_w[l], w.fc1_b[l] = t(sd[b + "mlp.fc1.weight"]), t(sd[b + "mlp.fc1.bias"])
            w.fc2_w[l], w.fc2_b[l] = t(sd[b + "mlp.fc2.weight"]), t(sd[b + "mlp.fc2.bias"])
            w.qkv_w3[l], w.out_w3[l] = t3(sd[b + "attn.in_proj_weight"]), t3(sd[b + "attn.out_proj.weight"])
            w.fc1_w3[l], w.fc2_w3[l] = t3(sd[b + "mlp.fc1.weight"]), t3(sd[b + "mlp.fc2.weight"])
            if lowp:
                w.qkv_wb[l], w.out_wb[l] = tb(sd[b + "attn.in_proj_weight"]), tb(sd[b + "attn.out_proj.weight"])
                w.fc1_wb[l], w.fc2_wb[l] = tb(sd[b + "mlp.fc1.weight"]), tb(sd[b + "mlp.fc2.weight"])
            if mx:
                w.qkv_wm[l], w.qkv_wms[l] = tm(sd[b + "attn.in_proj_weight"])
                w.out_wm[l], w.out_wms[l] = tm(sd[b + "attn.out_proj.weight"])
                w.fc1_wm[l], w.fc1_wms[l] = tm(sd[b + "mlp.fc1.weight"])
                w.fc2_wm[l], w.fc2_wms[l] = tm(sd[b + "mlp.fc2.weight"])
        w.fin_ada_w, w.fin_ada_b = t(sd["final_layer.adaLN_modulation.1.weight"]), t(sd["final_layer.adaLN_modulation.1.bias"])
        w.fin_w, w.fin_b = t(sd["final_layer.linear.weight"]), t(sd["final_layer.linear.bias"])
        self.cfg, self.w = cfg, w
        self.options = options                          # this denoiser's own overrides of the library's tuning options (property below)
        self.stream = self.new_stream()
        self._ws = None

    # replacing the set rewrites cfg.options: the config never points at a destroyed set
    @property
    def options(self) -> "_lib.OptionSet":
        return self._options

    @options.setter
    def options(self, value):
        new = value if isinstance(value, _lib.OptionSet) else _lib.OptionSet(value)
        self.cfg.options = new.handle
        self._options = new

    @classmethod
    def from_preset(cls, name: str, state_dict: dict, **kw):
        depth, hidden, heads = DIT_PRESETS[name]
        return cls(state_dict, depth, hidden, heads, **kw)

    @classmethod
    def from_reference(cls, model, device="cuda", **kw):
        """`model`: a reference osu_diffusion `DiT` module."""
        return cls(model.state_dict(), len(model.blocks), model.final_layer.linear.in_features, model.num_heads,
                   context_size=model.context_size, class_size=model.y_embedder.class_embedding[0].in_features,
                   device=device, **kw)

    # nn.Module-ish conveniences used by the reference pipeline
    def eval(self):
        return self

    def parameters(self):
        return iter(self._keep)

    # ---- helpers -----------------------------------------------------------------------------------
    def workspace(self, N: int, T: int, n_steps: int = 0) -> torch.Tensor:
        need = (self.lib.mh_ddpm_loop_workspace_bytes(C.byref(self.cfg), N, T, n_steps) if n_steps > 0
                else self.lib.mh_dit_workspace_bytes(C.byref(self.cfg), N, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- device and streams: the only places this class touches torch.cuda, so that a CPU test can put a stand-in for the
    # library underneath the unchanged host logic (tests/test_oracle_pinned.py); the product has no CPU path ------------------
    def require_gpu(self):
        if not torch.cuda.is_available():
            raise RuntimeError("DiTHIP needs a ROCm GPU; there is no CPU fallback")

    def new_stream(self):
        return torch.cuda.Stream(self.device)

    def caller_stream(self) -> int:
        """the raw handle of the stream the caller is on"""
        return torch.cuda.current_stream(self.device).cuda_stream

    @contextlib.contextmanager
    def on_own_stream(self):
        """the denoiser's own stream, ordered behind the caller's stream on entry and in front of it on exit; yields its handle"""
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            yield self.stream.cuda_stream
        cur.wait_stream(self.stream)

    def band_from_mask(self, attn_mask: Optional[torch.Tensor], T: int):
        """The reference passes a (T, T) bool mask, True = masked, built as a band
        (diffusion_pipeline.py:146-148): query q may attend key k iff -(band-1) <= k - q <= band; with `pad_sequence` it is
        padded with "allowed" rows and columns (:190).  Recover (band, open_from) and verify the mask really has that
        structure (anything else is refused)."""
        if attn_mask is None:
            return 0, 0
        if isinstance(attn_mask, BandMask):      # our own pipeline: the band is known, nothing to analyse
            if attn_mask.T != T:
                raise ValueError(f"band mask built for T={attn_mask.T}, sequence has T={T}")
            return attn_mask.band, attn_mask.open_from
        m = attn_mask.to("cpu")
        if m.dtype != torch.bool or m.shape != (T, T):
            raise NotImplementedError("attn_mask must be a (T, T) bool band mask")
        if not bool(m.any()):
            return 0, 0
        # trailing rows AND columns that mask nothing = padding; the band is among the positions before them
        real = T
        while real > 0 and not bool(m[real - 1].any()) and not bool(m[:, real - 1].any()):
            real -= 1
        allowed0 = int((~m[0, :real]).sum().item())      # keys 0..band visible from query 0
        band = allowed0 - 1
        expect = BandMask(T, band, open_from=real if real < T else 0).to_tensor()
        if band <= 0 or not torch.equal(expect, m):
            raise NotImplementedError("attn_mask is not the banded mask of diffusion_pipeline.py:146-148 (optionally padded, :190)")
        return band, (real if real < T else 0)

    # ---- B3 ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_with_cfg(self, x, t, c, y, cfg_scale, attn_mask=None, key_padding_mask=None):
        """key_padding_mask is accepted and ignored, exactly like the reference block
        (models.py:132,145-151 never forwards it)."""
        dev = self.device
        N, _, T = x.shape
        band, open_from = self.band_from_mask(attn_mask, T)
        x = x.to(dev, torch.float32).contiguous()
        c = c.to(dev, torch.float32).contiguous()
        y = y.to(dev, torch.float32).contiguous()
        t32 = t.to(dev, torch.int32).contiguous()
        out = torch.empty((N, 4, T), dtype=torch.float32, device=dev)
        ws = self.workspace(N, T)
        s = self.caller_stream()
        rc = self.lib.mh_dit_forward_cfg(C.byref(self.cfg), C.byref(self.w), x.data_ptr(), t32.data_ptr(),
                                         c.data_ptr(), y.data_ptr(), float(cfg_scale), band, open_from, N, T, out.data_ptr(),
                                         ws.data_ptr(), ws.numel(), s)
        _lib.check(rc, "mh_dit_forward_cfg")
        return out

    __call__ = forward_with_cfg


class BandMask:
    """The banded attention mask of diffusion_pipeline.py:145-148 as a description instead of a (T, T) tensor: query q
    may attend key k iff -(band-1) <= k - q <= band (band = the pipeline's `seq_len`).  What DiffusionPipelineHIP
    passes as `attn_mask`; a foreign caller's bool tensor is analysed (and verified to be a band) on every call.

    `open_from` (pad_sequence, :186-193): the window was padded to T positions, the mask padded with "allowed" -- positions
    >= open_from attend and are attended by everything; the band lives among the first open_from positions."""

    def __init__(self, T: int, seq_len: int, open_from: int = 0):
        real = open_from if open_from else T
        self.T, self.band = int(T), (int(seq_len) if seq_len < real else 0)   # band >= the real length masks nothing
        self.open_from = int(open_from) if (open_from and open_from < T and self.band) else 0

    def to_tensor(self, device="cpu") -> torch.Tensor:
        q = torch.arange(self.T, device=device)[:, None]
        k = torch.arange(self.T, device=device)[None, :]
        if self.band == 0:
            return torch.zeros(self.T, self.T, dtype=torch.bool, device=device)
        m = ~((q >= k - self.band) & (q < k + self.band))
        if self.open_from:
            m = m & (q < self.open_from) & (k < self.open_from)
        return m


class InpaintSpec:
    """`denoised_fn` of the reference pipeline when no sliders need re-projection:
    `x = torch.where(mask, x, z_part)` (diffusion_pipeline.py:203-206).  Callable so that it also works
    with the reference's own python loop; recognised by SpacedDiffusionHIP and fused into K9."""

    def __init__(self, mask: torch.Tensor, ref: torch.Tensor):
        self.mask, self.ref = mask, ref

    def __call__(self, x):
        return torch.where(self.mask.to(x.device), x, self.ref.to(x.device))


CURVE_CODE = {"Linear": 0, "PerfectCurve": 1, "Catmull": 2, "Bezier": 3}
MAX_BEZIER_SPAN = 32     # SL_MAXCP of csrc/slider.hip


def pack_slider_set(sliders_per_chunk, start: int, end: int):
    """Host side of MhSliderSet for the window [start, end): per chunk the sliders that lie entirely inside it
    (diffusion_pipeline.py:210-212), indices made window-relative.  Returns the seven flat lists of the struct
    (chunk_active, chunk_off, type, cp_off, cp_idx, end_idx, length).  Refuses what the kernel cannot reproduce: a curve
    span of more than MAX_BEZIER_SPAN points, and sliders that share sequence points (the reference re-projects in order,
    the kernel in parallel)."""
    active, chunk_off, types, cp_off, cp_idx, end_idx, length = [], [0], [], [0], [], [], []
    for sl in sliders_per_chunk:
        active.append(1 if len(sl) > 0 else 0)
        used = set()
        for s in sl:
            idx = np.asarray(s.seq_indices, dtype=np.int64)
            if np.any((idx < start) | (idx >= end)) or s.end_index < start or s.end_index >= end:
                continue                                       # (:210-212)
            run = longest = 1
            for a, b in zip(idx[:-1], idx[1:]):                # spans only ever split further (equal positions)
                run = 1 if a == b else run + 1
                longest = max(longest, run)
            if longest > MAX_BEZIER_SPAN:
                raise NotImplementedError(f"slider with a {longest}-point curve span (device limit {MAX_BEZIER_SPAN})")
            pts = set(int(i) for i in idx) | {int(s.end_index)}
            if int(s.end_index) in set(int(i) for i in idx) or (used & pts):
                raise NotImplementedError("sliders sharing sequence points are re-projected in order by the reference; "
                                          "the device kernel needs them disjoint")
            used |= pts
            types.append(CURVE_CODE.get(s.curve_type, 3))      # calculate_subpath: anything else is a Bezier
            cp_idx.extend(int(i) - start for i in idx)
            cp_off.append(len(cp_idx))
            end_idx.append(int(s.end_index) - start)
            length.append(float(s.length))
        chunk_off.append(len(types))
    return active, chunk_off, types, cp_off, cp_idx, end_idx, length


class SliderInpaintSpec(InpaintSpec):
    """`denoised_fn` of the reference pipeline WITH sliders (diffusion_pipeline.py:201-222): the in-paint `where`, then
    every slider that lies entirely inside the window [start, end) gets its end point moved to
    `SliderPath(curve_type, control points).position_at(length / get_distance())`.  Runs on the device
    (mh_slider_project); SpacedDiffusionHIP recognises it and keeps the whole loop one replayed hipGraph.

    sliders_per_chunk: one list per song chunk (row b and row B + b of the CFG batch) of objects with the fields of
    the reference's DiffusionSlider (:30-35): seq_indices, end_index, curve_type, length -- indices count points of
    the whole song, `start` is the window's first point."""

    def __init__(self, mask: torch.Tensor, ref: torch.Tensor, sliders_per_chunk, start: int, end: int):
        super().__init__(mask, ref)
        dev = ref.device
        if dev.type != "cuda":
            raise RuntimeError("SliderInpaintSpec lives on the GPU; there is no CPU path")
        N, _, T = ref.shape
        B = len(sliders_per_chunk)
        if N != 2 * B or T < end - start:        # (T > end - start: the window was padded, pad_sequence)
            raise ValueError(f"slider lists for {B} chunks / window {start}:{end} do not match x0 {tuple(ref.shape)}")
        active, chunk_off, types, cp_off, cp_idx, end_idx, length = pack_slider_set(sliders_per_chunk, start, end)
        self.n_sliders = len(types)

        def dv(a, dt):
            return torch.tensor(a if len(a) else [0], dtype=dt).to(dev)

        self._t = [dv(active, torch.uint8), dv(chunk_off, torch.int32), dv(types, torch.int32), dv(cp_off, torch.int32),
                   dv(cp_idx, torch.int32), dv(end_idx, torch.int32), dv(length, torch.float64)]
        self.cset = _lib.MhSliderSet(B, B, self.n_sliders, *[t.data_ptr() for t in self._t])
        self._mask8 = self.mask.to(dev).to(torch.uint8).contiguous()
        self._ref = self.ref.to(dev, torch.float32).contiguous()

    def project_(self, x0: torch.Tensor, stream: int) -> torch.Tensor:
        """in place on a contiguous fp32 device tensor [2B, 2, T]"""
        N, _, T = x0.shape
        _lib.check(_lib.load().mh_slider_project(x0.data_ptr(), self._mask8.data_ptr(), self._ref.data_ptr(), N, T,
                                                 C.byref(self.cset), stream), "mh_slider_project")
        return x0

    def __call__(self, x):
        x0 = x.to(self._ref.device, torch.float32).contiguous().clone()
        return self.project_(x0, torch.cuda.current_stream(x0.device).cuda_stream)


# ---- schedule (host, float64) --------------------------------------------------------------------
def named_beta_schedule(name: str, n: int) -> np.ndarray:
    if name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if name == "squaredcos_cap_v2":
        f = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)], dtype=np.float64)
    raise NotImplementedError(f"unknown beta schedule: {name}")


def space_timesteps(num_timesteps: int, section_counts) -> set:
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(v) for v in section_counts.split(",")]
    per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class SpacedDiffusionHIP:
    """epsilon-prediction / learned-range-variance sampler (what `create_diffusion` returns with its
    defaults, utils/diffusion/__init__.py:10-47), sampling only."""

    def __init__(self, use_timesteps, betas: np.ndarray):
        base_ac = np.cumprod(1.0 - np.asarray(betas, dtype=np.float64))
        last, new_betas, self.timestep_map = 1.0, [], []
        for i, ac in enumerate(base_ac):
            if i in use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        b = np.array(new_betas, dtype=np.float64)
        self.betas = b
        self.num_timesteps = len(b)
        ac = np.cumprod(1.0 - b)
        ac_prev = np.append(1.0, ac[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        pv = b * (1.0 - ac_prev) / (1.0 - ac)
        self.posterior_log_variance_clipped = np.log(np.append(pv[1], pv[1:])) if len(pv) > 1 else np.array([])
        self.posterior_mean_coef1 = b * np.sqrt(ac_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - ac_prev) * np.sqrt(1.0 - b) / (1.0 - ac)

    def coef_table(self) -> torch.Tensor:
        """fp32 [n_steps, 7] rows = (min_log, max_log, sqrt_recip, sqrt_recipm1, coef1, coef2, nonzero)."""
        n = self.num_timesteps
        tab = np.stack([self.posterior_log_variance_clipped, np.log(self.betas), self.sqrt_recip_alphas_cumprod,
                        self.sqrt_recipm1_alphas_cumprod, self.posterior_mean_coef1, self.posterior_mean_coef2,
                        (np.arange(n) != 0).astype(np.float64)], axis=1)
        return torch.from_numpy(tab).float()  # float64 -> .float(), as _extract_into_tensor does

    @torch.no_grad()
    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn: Optional[Callable] = None,
                      cond_fn=None, model_kwargs=None, device=None, progress=False, step_noise=None):
        """Signature of GaussianDiffusion.p_sample_loop.  `model` must be `DiTHIP.forward_with_cfg` (or a
        DiTHIP).  `step_noise` (optional, [n_steps, *shape], index = call order) injects the per-step
        gaussian noise for parity tests; otherwise `torch.randn_like` is called once per step on the
        device, consuming the global generator exactly as the reference loop does."""
        dit = getattr(model, "__self__", model)
        if not isinstance(dit, DiTHIP):
            raise TypeError("p_sample_loop: model must be DiTHIP.forward_with_cfg")
        if cond_fn is not None or not clip_denoised:
            raise NotImplementedError("cond_fn / clip_denoised=False are not used by the pipeline and not built")
        mk = dict(model_kwargs or {})
        dev = dit.device
        x = (noise if noise is not None else torch.randn(*shape, device=dev)).to(dev, torch.float32).contiguous().clone()
        N, _, T = x.shape
        c = mk["c"].to(dev, torch.float32).contiguous()
        y = mk["y"].to(dev, torch.float32).contiguous()
        cfg_scale = float(mk.get("cfg_scale", 1.0))
        band, open_from = dit.band_from_mask(mk.get("attn_mask"), T)
        n = self.num_timesteps
        if step_noise is None:
            step_noise = torch.stack([torch.randn_like(x) for _ in range(n)])
        step_noise = step_noise.to(dev, torch.float32)
        noise_by_i = torch.flip(step_noise, dims=[0]).contiguous()  # call k handles loop index i = n-1-k
        coefs = self.coef_table().to(dev).contiguous()
        t_map = torch.tensor(self.timestep_map, dtype=torch.int32, device=dev)
        lib, s = dit.lib, None
        ws = dit.workspace(N, T, n)

        if denoised_fn is None or isinstance(denoised_fn, InpaintSpec):
            imask = iref = sset = None
            if denoised_fn is not None:
                imask = denoised_fn.mask.to(dev).to(torch.uint8).contiguous()
                iref = denoised_fn.ref.to(dev, torch.float32).contiguous()
            if isinstance(denoised_fn, SliderInpaintSpec):
                sset = C.byref(denoised_fn.cset)
            with dit.on_own_stream() as own:
                rc = lib.mh_ddpm_sample_loop(C.byref(dit.cfg), C.byref(dit.w), x.data_ptr(), c.data_ptr(),
                                             y.data_ptr(), cfg_scale, band, open_from, N, T, n, t_map.data_ptr(),
                                             coefs.data_ptr(), noise_by_i.data_ptr(), _lib.ptr(imask), _lib.ptr(iref),
                                             sset, ws.data_ptr(), ws.numel(), own)
            _lib.check(rc, "mh_ddpm_sample_loop")
            return x

        # arbitrary host denoised_fn (slider re-projection): per-step launches, x0 round trip through python
        s = dit.caller_stream()
        mout = torch.empty((N, 4, T), dtype=torch.float32, device=dev)
        x0 = torch.empty_like(x)
        for i in reversed(range(n)):
            t32 = torch.full((N,), self.timestep_map[i], dtype=torch.int32, device=dev)
            _lib.check(lib.mh_dit_forward_cfg(C.byref(dit.cfg), C.byref(dit.w), x.data_ptr(), t32.data_ptr(),
                                              c.data_ptr(), y.data_ptr(), cfg_scale, band, open_from, N, T, mout.data_ptr(),
                                              ws.data_ptr(), ws.numel(), s), "mh_dit_forward_cfg")
            _lib.check(lib.mh_ddpm_step(mout.data_ptr(), x.data_ptr(), noise_by_i[i].data_ptr(), coefs[i].data_ptr(),
                                        None, None, None, 1, N, T, x.data_ptr(), x0.data_ptr(), s), "mh_ddpm_step")
            x0n = denoised_fn(x0.clone()).to(dev, torch.float32).contiguous()
            _lib.check(lib.mh_ddpm_step(mout.data_ptr(), x.data_ptr(), noise_by_i[i].data_ptr(), coefs[i].data_ptr(),
                                        None, None, x0n.data_ptr(), 0, N, T, x.data_ptr(), None, s), "mh_ddpm_step")
        return x


    @torch.no_grad()
    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn: Optional[Callable] = None, cond_fn=None,
                 model_kwargs=None, noise=None):
        """One reverse step, signature of GaussianDiffusion.p_sample (gaussian_diffusion.py:420-467); `t` holds the
        SPACED loop index of every batch row (all equal), mapped through `timestep_map` like _WrappedModel does
        (respace.py:127-132).  `noise` injects the gaussian draw (default: torch.randn_like on the device -- drawn
        even when t == 0, where it is multiplied by 0, as in the reference).  Returns sample / pred_xstart."""
        dit = getattr(model, "__self__", model)
        if not isinstance(dit, DiTHIP):
            raise TypeError("p_sample: model must be DiTHIP.forward_with_cfg")
        if cond_fn is not None or not clip_denoised:
            raise NotImplementedError("cond_fn / clip_denoised=False are not used by the pipeline and not built")
        mk = dict(model_kwargs or {})
        dev = dit.device
        x = x.to(dev, torch.float32).contiguous()
        N, _, T = x.shape
        ti = t.to("cpu").tolist()
        if len(set(ti)) != 1:
            raise NotImplementedError("p_sample: one loop index per call")
        i = int(ti[0])
        c = mk["c"].to(dev, torch.float32).contiguous()
        y = mk["y"].to(dev, torch.float32).contiguous()
        band, open_from = dit.band_from_mask(mk.get("attn_mask"), T)
        noise = (torch.randn_like(x) if noise is None else noise.to(dev, torch.float32)).contiguous()
        coef = self.coef_table()[i].to(dev).contiguous()
        t32 = torch.full((N,), self.timestep_map[i], dtype=torch.int32, device=dev)
        mout = torch.empty((N, 4, T), dtype=torch.float32, device=dev)
        out, x0 = torch.empty_like(x), torch.empty_like(x)
        ws = dit.workspace(N, T)
        s = dit.caller_stream()
        lib = dit.lib
        _lib.check(lib.mh_dit_forward_cfg(C.byref(dit.cfg), C.byref(dit.w), x.data_ptr(), t32.data_ptr(), c.data_ptr(),
                                          y.data_ptr(), float(mk.get("cfg_scale", 1.0)), band, open_from, N, T, mout.data_ptr(),
                                          ws.data_ptr(), ws.numel(), s), "mh_dit_forward_cfg")
        if isinstance(denoised_fn, SliderInpaintSpec):     # eps -> x0 | in-paint + slider ends | posterior, all on the device
            _lib.check(lib.mh_ddpm_step(mout.data_ptr(), x.data_ptr(), noise.data_ptr(), coef.data_ptr(), None, None,
                                        None, 1, N, T, out.data_ptr(), x0.data_ptr(), s), "mh_ddpm_step")
            denoised_fn.project_(x0, s)
            _lib.check(lib.mh_ddpm_step(mout.data_ptr(), x.data_ptr(), noise.data_ptr(), coef.data_ptr(), None, None,
                                        x0.data_ptr(), 0, N, T, out.data_ptr(), x0.data_ptr(), s), "mh_ddpm_step")
        elif denoised_fn is None or isinstance(denoised_fn, InpaintSpec):
            imask = iref = None
            if denoised_fn is not None:
                imask = denoised_fn.mask.to(dev).to(torch.uint8).contiguous()
                iref = denoised_fn.ref.to(dev, torch.float32).contiguous()
            _lib.check(lib.mh_ddpm_step(mout.data_ptr(), x.data_ptr(), noise.data_ptr(), coef.data_ptr(), _lib.ptr(imask),
                                        _lib.ptr(iref), None, 0, N, T, out.data_ptr(), x0.data_ptr(), s), "mh_ddpm_step")
        else:
            _lib.check(lib.mh_ddpm_step(mout.data_ptr(), x.data_ptr(), noise.data_ptr(), coef.data_ptr(), None, None,
                                        None, 1, N, T, out.data_ptr(), x0.data_ptr(), s), "mh_ddpm_step")
            x0n = denoised_fn(x0.clone()).to(dev, torch.float32).contiguous()
            _lib.check(lib.mh_ddpm_step(mout.data_ptr(), x.data_ptr(), noise.data_ptr(), coef.data_ptr(), None, None,
                                        x0n.data_ptr(), 0, N, T, out.data_ptr(), x0.data_ptr(), s), "mh_ddpm_step")
        return {"sample": out, "pred_xstart": x0}


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, sigma_small=False,
                     predict_xstart=False, learn_sigma=True, rescale_learned_sigmas=False, diffusion_steps=1000,
                     use_l1=False) -> SpacedDiffusionHIP:
    """Same signature as the reference factory (utils/diffusion/__init__.py:10-47); only the sampling
    configuration the pipeline uses is built (epsilon prediction, learned-range sigma)."""
    if predict_xstart or not learn_sigma or sigma_small:
        raise NotImplementedError("the pipeline samples with epsilon prediction + learned-range variance only")
    betas = named_beta_schedule(noise_schedule, diffusion_steps)
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    return SpacedDiffusionHIP(space_timesteps(diffusion_steps, timestep_respacing), betas)
