#!/usr/bin/env python3
"""A/B timing of one engine option on one workload (tuning aid; run on the GPU box):

    python3 profiles/ab_option.py --arch halfcheetah --batch 1 --option ccw_prefetch --values 0,4 [--steps 200]

Prints microseconds per denoise step for every value, hipGraph replay of a `--steps`-step loop, best of
`--repeats` after one warm-up replay; values are visited round-robin so that clock drift hits all alike."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dynamics_aware_diffusion_amd import GaussianDiffusion, GuidedPolicy, TemporalUnet  # noqa: E402
from dynamics_aware_diffusion_amd.utils import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="halfcheetah", choices=sorted(synth.ARCHS))
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--option", required=True)
ap.add_argument("--values", required=True)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--repeats", type=int, default=5)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--horizon", type=int, default=32)
args = ap.parse_args()

dev = torch.device("cuda:0")
od, ad, dim, mults, T = synth.ARCHS[args.arch]
td = od + ad
unet = TemporalUnet(td, dim=dim, dim_mults=mults)
unet.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_unet_state(td, dim, mults, seed=0).items()})
diff = GaussianDiffusion(unet, args.horizon, od, ad, n_timesteps=T).to(dev)
diff.sampler_rng, diff.seed, diff.use_graph = "philox", 1, bool(args.graph)
diff.n_timesteps = min(T, args.steps)
cond = torch.zeros(1, td)
cond[0, :od] = torch.from_numpy(synth.uniform(1, "bench.cond", (od,), 0.9))
pol = GuidedPolicy(diff, None)
c = {0: cond.to(dev)}
eng = diff._engine(dev)
values = [int(v) for v in args.values.split(",")]
best = {v: float("inf") for v in values}
for rep in range(args.repeats + 1):
    for v in values:
        eng.debug_set_option(args.option, v)
        pol.sample_loop(batch_size=args.batch, conditions=c)          # capture / warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pol.sample_loop(batch_size=args.batch, conditions=c)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) * 1e6 / diff.n_timesteps
        if rep > 0:
            best[v] = min(best[v], us)
for v in values:
    print(f"{args.arch} B={args.batch} {args.option}={v}: {best[v]:.1f} us per denoise step", flush=True)
