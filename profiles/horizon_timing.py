import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dynamics_aware_diffusion_amd import GaussianDiffusion, GuidedPolicy, TemporalUnet
from dynamics_aware_diffusion_amd.utils import synth
dev = torch.device("cuda:0")
for H in (32, 64, 128):
    for B in (1, 256 * 32 // H):
        td, dim, mults = 6, 128, (1, 2, 4)
        unet = TemporalUnet(td, dim=dim, dim_mults=mults)
        unet.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_unet_state(td, dim, mults, seed=0).items()})
        diff = GaussianDiffusion(unet, H, 4, 2, n_timesteps=100).to(dev)
        diff.sampler_rng, diff.seed, diff.use_graph = "philox", 1, B == 1
        pol = GuidedPolicy(diff, None)
        cond = {0: torch.zeros(1, td, device=dev)}
        pol.sample_loop(batch_size=B, conditions=cond); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); pol.sample_loop(batch_size=B, conditions=cond); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        f = synth.unet_flops_per_sample(td, dim, mults, H) * B
        print(f"PointMaze net H={H} B={B}: {best*1e4:.1f} us per denoise step, {f*100/best/1e12:.1f} TFLOP/s, plan(cc launches)={diff._engine(dev).small_batch_plan(B)}", flush=True)
