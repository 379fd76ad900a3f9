#!/usr/bin/env python3
"""Time one training step's device work (GaussianDiffusion.loss forward + loss.backward()) on the HIP engine:

    python3 profiles/train_step_timing.py --arch pointmaze --batch 256

Reports ms for the training forward, the backward pass, and the algorithmic conv TFLOP/s (backward = 2x the
forward's conv FLOPs: data gradient + weight gradient).  The engine is rebuilt when parameters change (weights
are re-packed on the host), so the timed region re-uses one engine: it measures kernels, not the re-pack."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet  # noqa: E402
from dynamics_aware_diffusion_amd.utils import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="pointmaze", choices=sorted(synth.ARCHS))
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--repeats", type=int, default=5)
ap.add_argument("--dim", type=int, default=0, help="override the architecture's dim (48 / 96: zero-padded GroupNorm groups)")
ap.add_argument("--horizon", type=int, default=32, help="24 / 48 / 100: zero-padded rows")
args = ap.parse_args()
dev = torch.device("cuda:0")
od, ad, dim, mults, T = synth.ARCHS[args.arch]
dim = args.dim or dim
H = args.horizon
td = od + ad
unet = TemporalUnet(td, dim=dim, dim_mults=mults)
unet.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_unet_state(td, dim, mults, seed=0).items()})
diff = GaussianDiffusion(unet, H, od, ad, n_timesteps=T).to(dev)
x0 = torch.from_numpy(synth.normal_like(3, "train.x0", (args.batch, H, td))).to(dev).clamp(-1, 1)
f = synth.unet_flops_per_sample(td, dim, mults, H) * args.batch      # (the REAL net's FLOPs: padding is overhead)
fwd, bwd = [], []
for rep in range(args.repeats + 1):
    for p in diff.parameters():
        p.grad = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = diff.loss(x0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if rep:
        fwd.append(t1 - t0)
        bwd.append(t2 - t1)
# steady state: steps back to back as a training loop issues them (no synchronisation in between — the host side of
# step i + 1 runs under the kernels of step i)
N = 10
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    for p in diff.parameters():
        p.grad = None
    diff.loss(x0).backward()
torch.cuda.synchronize()
steady = (time.perf_counter() - t0) / N
# the same with an optimiser step in the loop: every parameter changes, so the next forward re-derives the engine's
# packed copies on the device (dad_model_refresh_weights) — the cost this design adds to a training step
opt = torch.optim.SGD(diff.parameters(), lr=1e-5)
for _ in range(2):
    opt.zero_grad(set_to_none=True)
    diff.loss(x0).backward()
    opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    opt.zero_grad(set_to_none=True)
    diff.loss(x0).backward()
    opt.step()
torch.cuda.synchronize()
steady_opt = (time.perf_counter() - t0) / N
fm, bm = min(fwd) * 1e3, min(bwd) * 1e3
print(f"{args.arch} dim={dim} H={H} B={args.batch}: training forward {fm:.2f} ms ({f / fm / 1e9:.1f} TFLOP/s), backward {bm:.2f} ms "
      f"({2 * f / bm / 1e9:.1f} TFLOP/s algorithmic), {N} steps back to back {steady * 1e3:.2f} ms per step "
      f"({3 * f / steady / 1e12:.1f} TFLOP/s, {args.batch / steady:.0f} samples/s), with SGD step + weight refresh "
      f"{steady_opt * 1e3:.2f} ms per step ({args.batch / steady_opt:.0f} samples/s), loss {float(loss):.5f}", flush=True)
