"""Batched DiT denoiser: B chunks x 100 DDPM steps as one replayed hipGraph.
    python tools/dit_batch_bench.py B [preset=DiT-S] [fp32|bf16|mx8]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapperatorinator_amd.dit import BandMask, DiTHIP, create_diffusion
from mh_testing import DIT_PRESETS, random_dit_state_dict, synthetic_dit_inputs
dev = torch.device("cuda", 0)
B, Tq = int(sys.argv[1]), 128
preset = sys.argv[2] if len(sys.argv) > 2 else "DiT-S"
mode = sys.argv[3] if len(sys.argv) > 3 else "fp32"
depth, hidden, heads = DIT_PRESETS[preset]
dit = DiTHIP(random_dit_state_dict(depth, hidden, seed=0), depth, hidden, heads, device=dev,
             operand_dtype="mx8" if mode == "mx8" else (torch.bfloat16 if mode == "bf16" else torch.float32))
parts = [synthetic_dit_inputs(Tq, seed=b) for b in range(B)]
z = torch.cat([p[0][:1] for p in parts] + [p[0][1:] for p in parts]).to(dev)
c = torch.cat([p[1][:1] for p in parts] + [p[1][1:] for p in parts]).to(dev)
y = torch.cat([p[2][:1] for p in parts] + [p[2][1:] for p in parts]).to(dev)
diff = create_diffusion([100, 0, 0, 0, 0, 0, 0, 0, 0, 0], noise_schedule="squaredcos_cap_v2", diffusion_steps=1000)
kw = dict(c=c, y=y, cfg_scale=1.0, attn_mask=BandMask(Tq, 128))
noise = torch.randn(100, *z.shape, device=dev)
diff.p_sample_loop(dit.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=noise); torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(2): diff.p_sample_loop(dit.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=noise)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/2
fl = 2*B*(depth*(2.0*Tq*12*hidden*hidden+4.0*Tq*Tq*hidden)+2.0*Tq*528*hidden)*100
print(f"{preset} {mode} B={B}: {dt*1e3:.1f} ms per 100 steps, {fl/dt/1e12:.1f} TFLOP/s (fp32-equivalent flops)")
