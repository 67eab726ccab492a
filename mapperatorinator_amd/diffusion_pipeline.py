"""Host glue of the osu_diffusion stage: what `DiffisionPipeline.generate` does between
`events_to_sequence` and `events_with_pos` (reference diffusion_pipeline.py:111-287), on top of the HIP DiT + DDPM
loop (K7-K9).  Integer / indexing logic on the host, every float of the hot loop on the device:

  points_to_sequence   the tensor assembly at the end of `events_to_sequence` (:361-387): normalised positions,
                       times, and the 272-row conditioning `timestep_embedding(0.1 t, 128) | timestep_embedding(d, 128)
                       | 16 one-hot types`
  DiffusionPipelineHIP.generate_positions
                       banded mask (:145-148), CFG doubling (:158-166), `random_init` (:168-169), the overlapping
                       window loop (:276-284), per-window in-paint mask incl. start_time / end_time (:223-234),
                       `p_sample_loop` (:243-252), the refine iterations (:254-267) and `to_positions` (:171-176)

Grouping Events into hit-object points (`get_groups`, `update_event_times`) and writing positions back into Events
(`events_with_pos`) is the reference's own integer host code either side of this seam and stays there; the
`DiffusionSlider` list it builds (:389-437) is handed in as `sliders` and the slider end re-projection of `denoised_fn`
(:208-220) runs on the device inside the DDPM graph (csrc/slider.hip).  `pad_sequence=True` (:186-193) is reproduced with
the reference's own quirk: the pad positions stay attendable (the attention kernels' `open_from`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .dit import BandMask, DiTHIP, InpaintSpec, SliderInpaintSpec, create_diffusion


@dataclass
class DiffusionSlider:
    """(diffusion_pipeline.py:30-35) sequence points of the head and anchors, the point of the slider end, the curve
    type of the first anchor ('Bezier' | 'PerfectCurve' | 'Catmull') and the length in playfield pixels"""
    seq_indices: np.ndarray
    end_index: int
    curve_type: Optional[str]
    length: Optional[float]


# one-hot row of each hit-object type inside the 16 type rows (diffusion_pipeline.py:304-315); +1 for a new combo on
# CIRCLE / SLIDER_HEAD (:339-340), + repeat_type(repeats) on SLIDER_END (:343-347)
EVENT_INDEX = {"CIRCLE": 0, "SPINNER": 2, "SPINNER_END": 3, "SLIDER_HEAD": 4, "BEZIER_ANCHOR": 6, "PERFECT_ANCHOR": 7,
               "CATMULL_ANCHOR": 8, "RED_ANCHOR": 9, "LAST_ANCHOR": 10, "SLIDER_END": 11}
PLAYFIELD = (512.0, 384.0)


def repeat_type(repeat: int) -> int:
    """(osu_diffusion/utils/data_loading.py:43-49)"""
    if repeat < 4:
        return repeat - 1
    return 3 if repeat % 2 == 0 else 4


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """(osu_diffusion/utils/positional_embedding.py:28-49) -- same torch ops in the same order, on t's device."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def points_to_sequence(x, y, time, distance, type_index):
    """Hit-object points -> (seq_x (2,T), seq_o (T,), seq_c (272,T)), the tensors `events_to_sequence` returns
    (diffusion_pipeline.py:361-387).  `type_index` in [0, 16) is the final one-hot row (EVENT_INDEX + offsets)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    T = x.shape[0]
    seq = torch.zeros(20, T)
    seq[0], seq[1] = x, torch.as_tensor(y, dtype=torch.float32)
    seq[2], seq[3] = torch.as_tensor(time, dtype=torch.float32), torch.as_tensor(distance, dtype=torch.float32)
    idx = torch.as_tensor(type_index, dtype=torch.long)
    if T and (int(idx.min()) < 0 or int(idx.max()) >= 16):
        raise ValueError("type_index out of [0, 16)")
    seq[idx + 4, torch.arange(T)] = 1
    seq_x = seq[:2, :] / torch.tensor(PLAYFIELD).unsqueeze(1) * 2 - 1
    seq_o, seq_d = seq[2, :], seq[3, :]
    seq_c = torch.concatenate([timestep_embedding(seq_o * 0.1, 128).T, timestep_embedding(seq_d, 128).T, seq[4:, :]], 0)
    return seq_x, seq_o, seq_c


def band_mask(T: int, seq_len: int, device="cpu") -> torch.Tensor:
    """(diffusion_pipeline.py:145-148) True = masked; column i is visible from rows [i - seq_len, i + seq_len)."""
    q = torch.arange(T, device=device)[:, None]
    k = torch.arange(T, device=device)[None, :]
    return ~((q >= k - seq_len) & (q < k + seq_len))


class DiffusionPipelineHIP:
    """Same knobs as the reference object (diffusion_pipeline.py:40-63 / config.py:98-106)."""

    def __init__(self, model: DiTHIP, *, timesteps, diffusion_steps: int = 1000, noise_schedule: str = "squaredcos_cap_v2",
                 seq_len: int = 128, max_seq_len: int = 1024, overlap_buffer: int = 128, cfg_scale: float = 1.0,
                 refine_model: Optional[DiTHIP] = None, refine_iters: int = 10, random_init: bool = False,
                 pad_sequence: bool = False, start_time: Optional[float] = None, end_time: Optional[float] = None):
        # pad_sequence (reference diffusion_pipeline.py:186-193) pads every window to max_seq_len with zero positions and zero
        # context, pads the band mask with "allowed" and builds a key_padding_mask -- which DiTBlock.forward never hands to
        # its attention (models.py:133-150): every real query then attends all the pad tokens, so padding CHANGES the real
        # positions (tests/test_oracle_pinned.py::test_reference_dit_padding_changes_real_positions).  Reproduced as it is:
        # BandMask(open_from=) / the attention kernels' `open_from`.
        if not 0 <= 2 * overlap_buffer < max_seq_len:
            raise ValueError("overlap_buffer must be less than half of max_seq_len")
        self.model, self.refine_model = model, refine_model
        self.device = model.device
        self.timesteps, self.diffusion_steps, self.noise_schedule = timesteps, diffusion_steps, noise_schedule
        self.seq_len, self.max_seq_len, self.overlap_buffer = seq_len, max_seq_len, overlap_buffer
        self.cfg_scale, self.refine_iters, self.random_init = cfg_scale, refine_iters, random_init
        self.start_time, self.end_time, self.pad_sequence = start_time, end_time, bool(pad_sequence)

    def to_positions(self, samples: torch.Tensor) -> torch.Tensor:
        """(:171-176) drop the null-class half, [-1, 1] -> playfield pixels, to the CPU.  (2B, 2, T) -> (B, 2, T)"""
        samples, _ = samples.clone().chunk(2, dim=0)
        samples += 1
        samples /= 2
        samples *= torch.tensor(PLAYFIELD, device=samples.device).repeat(1, 1).unsqueeze(2)
        return samples.cpu()

    @torch.no_grad()
    def generate_positions(self, seq_x: torch.Tensor, seq_o: torch.Tensor, seq_c: torch.Tensor,
                           class_vector: torch.Tensor, unk_class_vector: torch.Tensor,
                           noise_source: Optional[Callable] = None,
                           denoised_fn_factory: Optional[Callable] = None,
                           sliders: Optional[Sequence] = None) -> torch.Tensor:
        """seq_* as returned by `events_to_sequence`; class vectors (C,) multi-hot.  Returns positions (1, 2, T) on the
        CPU, what the reference hands to `events_with_pos`.  (= generate_positions_batch with one chunk.)

        sliders: the DiffusionSlider list `events_to_sequence` returns (its 6th value); their end points are
            re-projected onto the slider paths every denoising step, on the device.
        noise_source(n, shape) -> fp32 [n, *shape]: the gaussian noise of n consecutive p_sample calls (parity tests
            inject the reference's draws); default: torch.randn on the device, one draw per call like `th.randn_like`.
        denoised_fn_factory(mask, z_part, start, end) -> callable: replaces the in-paint-only `denoised_fn` (e.g. with
            the slider re-projection); forces the per-step host round trip."""
        if seq_x.shape[1] == 0:
            return torch.zeros(1, 2, 0)
        return self.generate_positions_batch(seq_x[None], seq_o[None], seq_c[None], class_vector[None],
                                             unk_class_vector[None], noise_source, denoised_fn_factory,
                                             None if sliders is None else [list(sliders)])

    @torch.no_grad()
    def generate_positions_batch(self, seq_x: torch.Tensor, seq_o: torch.Tensor, seq_c: torch.Tensor,
                                 class_vectors: torch.Tensor, unk_class_vectors: torch.Tensor,
                                 noise_source: Optional[Callable] = None,
                                 denoised_fn_factory: Optional[Callable] = None,
                                 sliders: Optional[Sequence[Sequence]] = None) -> torch.Tensor:
        """B song-chunks with the SAME number of points T through the diffusion stage as ONE denoiser batch:
        seq_x (B, 2, T), seq_o (B, T), seq_c (B, 272, T), class vectors (B, C).  Returns positions (B, 2, T) on the CPU.
        `sliders`: one DiffusionSlider list per chunk (indices count the chunk's own points).

        The reference runs its pipeline once per beatmap with `n = 1` (diffusion_pipeline.py:157-166): a CFG batch of
        2 rows, i.e. M = 2 T rows per GEMM -- 67 latency-bound launches per step at T = 128.  Chunks are independent,
        so the B conditional rows and the B null-class rows are stacked as [cond_0..cond_{B-1} | null_0..null_{B-1}]
        (the layout `forward_with_cfg` splits in halves, models.py:306-317): M = 2 B T rows reach the 128x128 MFMA
        tiles and every launch is shared by all chunks.  Row b of the result equals the single-chunk run of chunk b
        (bit for bit when both use the same GEMM tile family: option gemm_splitk_tiles = 0).
        `noise_source(n, shape)` is asked for shape (2B, 2, T_window): rows b and B + b are chunk b's pair."""
        dev = self.device
        B, _, seq_len = seq_x.shape
        if seq_len == 0:
            return torch.zeros(B, 2, 0)
        if sliders is not None and len(sliders) != B:
            raise ValueError(f"{len(sliders)} slider lists for {B} chunks")
        diffusion = create_diffusion(timestep_respacing=self.timesteps, diffusion_steps=self.diffusion_steps,
                                     noise_schedule=self.noise_schedule)
        z = seq_x.to(dev, torch.float32)
        c = seq_c.to(dev, torch.float32)
        y = class_vectors.to(dev, torch.float32)
        y_null = unk_class_vectors.to(dev, torch.float32)
        z = torch.cat([z, z], 0)
        c = torch.cat([c, c], 0)
        y = torch.cat([y, y_null], 0)
        if self.random_init:
            z = torch.randn(*z.shape, device=dev)
        seq_o = seq_o.to(torch.float32).cpu()
        if noise_source is None:
            def noise_source(k, shape):
                return torch.stack([torch.randn(*shape, device=dev) for _ in range(k)])

        def sample_part(zfull, start, end, start_mask_size=0):
            z_part = zfull[:, :, start:end].contiguous()
            c_part = c[:, :, start:end].contiguous()
            real = end - start
            pad = self.max_seq_len - real if self.pad_sequence else 0          # (:186-193)
            if pad > 0:
                z_part = torch.nn.functional.pad(z_part, (0, pad)).contiguous()
                c_part = torch.nn.functional.pad(c_part, (0, pad)).contiguous()
            T = real + max(pad, 0)
            # True means it will be generated (:223-234); per chunk, the pair of a chunk shares its mask
            mask = torch.full(z_part.shape, False, dtype=torch.bool, device=dev)
            mask[:, :, start_mask_size:] = True
            for b in range(B):
                o_part = seq_o[b, start:end].contiguous()
                if self.start_time is not None:
                    k0 = int(torch.searchsorted(o_part, self.start_time, right=False))
                    mask[b, :, :k0] = False
                    mask[B + b, :, :k0] = False
                if self.end_time is not None:
                    k1 = int(torch.searchsorted(o_part, self.end_time, right=True))
                    mask[b, :, k1:] = False
                    mask[B + b, :, k1:] = False
            if not bool(mask.any()):
                return z_part[:, :, :real]
            if denoised_fn_factory is not None:
                denoised_fn = denoised_fn_factory(mask, z_part, start, end)
            elif sliders is not None and any(len(sl) > 0 for sl in sliders):
                denoised_fn = SliderInpaintSpec(mask, z_part, sliders, start, end)
            else:
                denoised_fn = InpaintSpec(mask, z_part)
            z_part = denoised_fn(z_part)
            model_kwargs = dict(c=c_part, y=y, cfg_scale=self.cfg_scale,
                                attn_mask=BandMask(T, self.seq_len, open_from=real if pad > 0 else 0), key_padding_mask=None)
            samples = diffusion.p_sample_loop(self.model.forward_with_cfg, z_part.shape, z_part, denoised_fn=denoised_fn,
                                              clip_denoised=True, model_kwargs=model_kwargs, device=dev,
                                              step_noise=noise_source(diffusion.num_timesteps, tuple(z_part.shape)))
            if self.refine_model is not None:
                # the reference refines with `self.model.forward_with_cfg` (:261), not the refine model: kept as is
                for _ in range(self.refine_iters):
                    t = torch.tensor([0] * samples.shape[0], device=dev)
                    out = diffusion.p_sample(self.model.forward_with_cfg, samples, t, denoised_fn=denoised_fn,
                                             clip_denoised=True, model_kwargs=model_kwargs,
                                             noise=noise_source(1, tuple(samples.shape))[0])
                    samples = out["sample"]
            return samples[:, :, :real] if pad > 0 else samples

        full = z.clone()
        ob = self.overlap_buffer
        for i in range(0, seq_len - ob * 2, self.max_seq_len - ob * 2):
            end = min(i + self.max_seq_len, seq_len)
            if i > 0:
                # the first buffer is done; the second was generated but is regenerated from the initial values (:279-282)
                full[:, :, i + ob:i + ob * 2] = z[:, :, i + ob:i + ob * 2]
            full[:, :, i:end] = sample_part(full, i, end, start_mask_size=ob if i > 0 else 0)
        return self.to_positions(full)
