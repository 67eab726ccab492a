"""One song chunk (Tq = 128 points, CFG batch 2) through the 100-step DDPM loop: the reference's own call shape of the diffusion stage
(diffusion_pipeline.py:243-252).    python tools/dit_one_chunk.py [DiT-B|DiT-S|DiT-XS]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapperatorinator_amd.dit import BandMask, DiTHIP, create_diffusion
from mh_testing import DIT_PRESETS, random_dit_state_dict, synthetic_dit_inputs
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "DiT-B"
d, h, n = DIT_PRESETS[name]
dit = DiTHIP(random_dit_state_dict(d, h, seed=0), d, h, n, device=dev)
z, c, y = [t.to(dev) for t in synthetic_dit_inputs(128, seed=0)]
diff = create_diffusion([100] + [0] * 9, noise_schedule="squaredcos_cap_v2", diffusion_steps=1000)
noise = torch.randn(100, *z.shape, device=dev)
def run():
    return diff.p_sample_loop(dit.forward_with_cfg, z.shape, z, model_kwargs=dict(c=c, y=y, cfg_scale=1.0, attn_mask=BandMask(128, 128)), step_noise=noise)
out = run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t = time.perf_counter(); o2 = run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print(name, "one chunk ms per 100 steps:", round(sorted(ts)[2] * 1e3, 2), "checksum", float(out.double().abs().sum()))
