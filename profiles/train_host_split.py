#!/usr/bin/env python3
"""Host side of one training step (PointMaze, B=256): how long Python + the library take to ISSUE a step (no
synchronisation inside the loop) against the per-step time with the GPU drained at the end, and the host time of
the pieces (time MLPs in torch, the two C calls).  `python3 profiles/train_host_split.py`"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
from dynamics_aware_diffusion_amd.utils import synth
from dynamics_aware_diffusion_amd import _engine
dev = torch.device("cuda:0")
od, ad, dim, mults, T = synth.ARCHS["pointmaze"]
for a in sys.argv[1:]:
    if a.startswith("--dim="): dim = int(a[6:])
td = od + ad
unet = TemporalUnet(td, dim=dim, dim_mults=mults)
unet.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_unet_state(td, dim, mults, seed=0).items()})
diff = GaussianDiffusion(unet, 32, od, ad, n_timesteps=T).to(dev)
x0 = torch.from_numpy(synth.normal_like(3, "train.x0", (256, 32, td))).to(dev).clamp(-1, 1)
acc = {}
def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, w)
for _ in range(3):
    for p in diff.parameters(): p.grad = None
    diff.loss(x0).backward()
eng = diff.model._engine
wrap(eng, "train_forward"); wrap(eng, "train_backward"); wrap(diff.model, "_time_projections"); wrap(diff, "q_sample")
wrap(diff.model, "_forward_autograd"); wrap(diff.model, "engine"); wrap(eng, "grad_layout")
lib = eng.lib
N = 20
torch.cuda.synchronize()
t0 = time.perf_counter(); tl = tb = 0.0
for _ in range(N):
    for p in diff.parameters(): p.grad = None
    a = time.perf_counter(); l = diff.loss(x0); b = time.perf_counter(); l.backward(); c = time.perf_counter()
    tl += b - a; tb += c - b
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"dim {dim}: host loop {host / N * 1e3:.2f} ms/step (loss() {tl / N * 1e3:.2f}, backward() {tb / N * 1e3:.2f}); with final sync {total / N * 1e3:.2f} ms/step")
for k, v in acc.items(): print(f"  {k}: {v / N * 1e3:.3f} ms/step (host)")
