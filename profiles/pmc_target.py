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
ap.add_argument("--project", action="store_true",
                help="BASELINE config 3: the dynamics projection after every step (PointMaze: double integrator, D = 196)")
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
policy = GuidedPolicy(diff, None)
if args.project:
    import contextlib
    import io
    import numpy as np
    from dynamics_aware_diffusion_amd import DynamicsAwarePolicy
    from dynamics_aware_diffusion_amd.dynamics import ProjectionMatrixBuilder

    class Norm:
        obs_mean = synth.normal_like(41, "bench.norm.obs_mean", (od,))
        obs_std = 1.0 + synth.uniform(41, "bench.norm.obs_std", (od,), 0.5)
        action_mean = synth.normal_like(41, "bench.norm.act_mean", (ad,))
        action_std = 1.0 + synth.uniform(41, "bench.norm.act_std", (ad,), 0.5)

    dt = 0.1
    A = np.eye(4); A[0, 2] = A[1, 3] = dt
    Bm = np.zeros((4, 2)); Bm[0, 0] = Bm[1, 1] = 0.5 * dt * dt; Bm[2, 0] = Bm[3, 1] = dt
    with contextlib.redirect_stdout(io.StringIO()):
        Pm = ProjectionMatrixBuilder(A, Bm, 4, 2).get_projection_matrix(32)
        policy = DynamicsAwarePolicy(diff, projection_matrix=Pm, normalizer=Norm(), state_dim=4, observation_dim=od,
                                     action_dim=ad, horizon=32, projection_schedule="noise_schedule",
                                     projection_strength=1.0, project_during_sampling=True)
print("[pmc_target] model on the device, starting the loop", flush=True)
plans = policy.sample_loop(batch_size=args.batch, conditions={0: cond.to(dev)})
torch.cuda.synchronize()
print(f"[pmc_target] done: {tuple(plans.shape)}", flush=True)
