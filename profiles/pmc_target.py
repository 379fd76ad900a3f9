#!/usr/bin/env python3
"""Minimal target for the rocprofv3 counter (--pmc) passes of profiles/collect.sh.

Under --pmc every dispatch is serialised and wrapped in counter start/stop packets (the retained
round-1 logs show 106-169 us per conv launch against 23 us unprofiled), so the target is kept to
what per-launch averages need: build the workload, ONE eager sampling loop, synchronise, exit.
No HIP events, no second arithmetic, no CPU baseline, no host-side reductions, no hipGraph."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dynamics_aware_diffusion_amd import GaussianDiffusion, GuidedPolicy, TemporalUnet  # noqa: E402
from dynamics_aware_diffusion_amd.utils import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="pointmaze", choices=sorted(synth.ARCHS))
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--denoise-steps", type=int, default=100)
ap.add_argument("--precision", default="fp32", choices=["fp32", "f16x3"])
args = ap.parse_args()

dev = torch.device("cuda:0")
od, ad, dim, mults, T = synth.ARCHS[args.arch]
td = od + ad
unet = TemporalUnet(td, dim=dim, dim_mults=mults)
unet.precision = args.precision
unet.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_unet_state(td, dim, mults, seed=0).items()})
diff = GaussianDiffusion(unet, 32, od, ad, n_timesteps=T).to(dev)
diff.sampler_rng, diff.seed, diff.use_graph = "philox", 1, False
diff.n_timesteps = min(T, args.denoise_steps)
cond = torch.zeros(1, td)
cond[0, :od] = torch.from_numpy(synth.uniform(1, "bench.cond", (od,), 0.9))
print("[pmc_target] model on the device, starting the loop", flush=True)
plans = GuidedPolicy(diff, None).sample_loop(batch_size=args.batch, conditions={0: cond.to(dev)})
torch.cuda.synchronize()
print(f"[pmc_target] done: {tuple(plans.shape)}", flush=True)
