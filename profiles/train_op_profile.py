#!/usr/bin/env python3
"""torch.profiler view of training steps (PointMaze, B=256): which ATen ops / memcpys the Python side of a step adds
around the engine's two C calls.  `python3 profiles/train_op_profile.py [--sgd] [--dim=48]`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
from dynamics_aware_diffusion_amd.utils import synth
dev = torch.device("cuda:0")
od, ad, dim, mults, T = synth.ARCHS["pointmaze"]
for a in sys.argv[1:]:
    if a.startswith("--dim="): dim = int(a[6:])
td = od + ad
unet = TemporalUnet(td, dim=dim, dim_mults=mults)
unet.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_unet_state(td, dim, mults, seed=0).items()})
diff = GaussianDiffusion(unet, 32, od, ad, n_timesteps=T).to(dev)
x0 = torch.from_numpy(synth.normal_like(3, "train.x0", (256, 32, td))).to(dev).clamp(-1, 1)
sgd = "--sgd" in sys.argv
opt = torch.optim.SGD(diff.parameters(), lr=1e-5)
def step():
    opt.zero_grad(set_to_none=True)
    diff.loss(x0).backward()
    if sgd: opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
print(f"{N} steps, dim={dim}, sgd={sgd}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
