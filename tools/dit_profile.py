#!/usr/bin/env python
"""DiT-S 100-step DDPM loop only (for rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mapperatorinator_amd.dit import DiTHIP, create_diffusion
from mh_testing import DIT_PRESETS, random_dit_state_dict, synthetic_dit_inputs
name = sys.argv[1] if len(sys.argv) > 1 else "DiT-S"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
depth, hidden, heads = DIT_PRESETS[name]
dit = DiTHIP(random_dit_state_dict(depth, hidden, seed=0), depth, hidden, heads, device="cuda")
z, c, y = [v.cuda() for v in synthetic_dit_inputs(T, seed=0)]
diff = create_diffusion([100, 0, 0, 0, 0, 0, 0, 0, 0, 0], noise_schedule="squaredcos_cap_v2", diffusion_steps=1000)
kw = dict(c=c, y=y, cfg_scale=1.0, attn_mask=None)
noise = torch.randn(100, *z.shape, device="cuda")
diff.p_sample_loop(dit.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=noise)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    diff.p_sample_loop(dit.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=noise)
torch.cuda.synchronize()
print(name, "T", T, "ms per 100 steps", (time.perf_counter() - t0) / 3 * 1e3)
