import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapperatorinator_amd.dit import BandMask, DiTHIP, create_diffusion
from mapperatorinator_amd.testing import DIT_PRESETS, random_dit_state_dict, synthetic_dit_inputs
dev = torch.device("cuda", 0)
depth, hidden, heads = DIT_PRESETS["DiT-S"]
dit = DiTHIP(random_dit_state_dict(depth, hidden, seed=0), depth, hidden, heads, device=dev)
B, Tq = int(sys.argv[1]), 128
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
print(f"B={B}: {dt*1e3:.1f} ms per 100 steps, {fl/dt/1e12:.1f} TFLOP/s")
