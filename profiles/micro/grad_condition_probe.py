#!/usr/bin/env python3
"""Probe of the open finding of gradient-fuzz seed 132 (tests/fuzz_parity.py header): net dim 24, mults (1, 1, 1, 4),
horizon 8, kernel_size 7 and its neighbours — HIP gradients against the fp32 oracle and against the oracle in float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import denoiser as orc
from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
from dynamics_aware_diffusion_amd.utils import synth
from tests.util import grad_scales
dev = torch.device("cuda:0")
it, td, Bg = 26, 2, 6
for dim, mults, H, ks in [(24, (1, 1, 1, 4), 8, 7), (24, (1, 1, 1, 4), 8, 5), (24, (1, 1, 1, 4), 8, 3), (24, (1, 1, 1, 4), 16, 7),
                          (24, (1, 1, 1, 4), 32, 7), (32, (1, 1, 1, 4), 8, 7), (32, (1, 1, 1, 4), 32, 7), (24, (1, 1, 4), 8, 7)]:
    state = synth.synth_unet_state(td, dim, mults, seed=100 + it, affine_jitter=0.3, kernel_size=ks)
    w = {k: torch.from_numpy(v) for k, v in state.items()}
    unet = TemporalUnet(td, dim=dim, dim_mults=mults, kernel_size=ks); unet.load_state_dict(w)
    diff = GaussianDiffusion(unet, H, td - 1, 1, n_timesteps=20).to(dev)
    x0 = torch.from_numpy(np.clip(synth.normal_like(400 + it, "fuzz.x0", (Bg, H, td)) * 0.5, -1, 1).astype(np.float32))
    tt = torch.from_numpy(np.array([(3 * i + 1) % 20 for i in range(Bg)], dtype=np.int64))
    nz = torch.from_numpy(synth.normal_like(400 + it, "fuzz.nz", (Bg, H, td)))
    with torch.enable_grad():
        x_t = diff.q_sample(x0.to(dev), tt.to(dev), nz.to(dev)).detach().requires_grad_(True)
        ((diff.model(x_t, tt.to(dev)) - nz.to(dev)) ** 2).mean().backward()
    torch.cuda.synchronize()
    sch = orc.schedule_buffers("cosine", 20)
    _, og, _ = orc.training_gradients(w, sch, x0, tt, nz)
    _, t64, _ = orc.training_gradients(orc.cast_weights(w, torch.float64), {k: v.double() for k, v in sch.items()}, x0.double(), tt, nz.double())
    sc = grad_scales(t64)
    rows = []
    for k, p in diff.model.named_parameters():
        g = p.grad.cpu().double()
        rows.append((float((g - og[k].double()).abs().max()) / sc[k], float((g - t64[k]).abs().max()) / sc[k], float((og[k].double() - t64[k]).abs().max()) / sc[k], k))
    a = max(rows)
    print(f"dim {dim} mults {mults} H {H} k {ks}: worst hip-vs-oracle {a[0]:.1e} ({a[3]}: hip-vs-fp64 {a[1]:.1e}, oracle-vs-fp64 {a[2]:.1e}); "
          f"max hip-vs-fp64 {max(r[1] for r in rows):.1e}, max oracle-vs-fp64 {max(r[2] for r in rows):.1e}", flush=True)
