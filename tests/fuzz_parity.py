"""Randomized parity sweep on the GPU (manual tool, not collected by pytest):

    python tests/fuzz_parity.py [seed] [cases] [small]

draws random supported architectures (dim, dim_mults incl. shrinking widths, horizon,
transition_dim), batch sizes and conv arithmetics, runs one denoiser evaluation through the HIP
path and compares it with the CPU oracle at the forward tolerance of the parity tests (5e-6).
and a short conditioned sampling loop with injected noise (2e-5).
Round 1: seeds 1, 7, 11, 12, 13 (forward) and 21, 22 (forward + loop) x 40 cases, 0 failures,
0 refusals (the first sweep found the identity-residual-over-concat decoder block, since supported).
Round 2: seeds 31 / 52 `small` (130 + 65 cases), 41 / 43 `wide` (56 + 99 cases, 142 of them through
conv_ccw), 51 default (58 cases): 0 failures, 0 refusals.
`wide`: the same on nets of 1024+ channels at batches of up to 128 rows (conv_ccw).
`small`: batches 1..16 in fp32 only, i.e. the consumer-combine kernels (conv_cc / conv_ccw); the line
shows how many launches of the case took them and how many the streamed-weight form."""
import sys, random
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import denoiser as orc
from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
from dynamics_aware_diffusion_amd.utils import synth
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
small = len(sys.argv) > 3 and sys.argv[3] in ("small", "wide")   # small batches, fp32: the consumer-combine kernels
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"              # ... on nets of 1024+ channels: conv_ccw
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    dim = rng.choice([32, 64, 128, 256])
    nlev = rng.choice([1, 2, 3, 4])
    mults = tuple([1] + [rng.choice([1, 2, 4, 8]) for _ in range(nlev - 1)])
    H = rng.choice([8, 16, 32, 64])
    if H >> (nlev - 1) < 4 or max(mults) * dim > 2048:
        continue
    if wide and (max(mults) * dim < 1024 or H > 32):
        continue
    td = rng.randint(2, 24)
    B = rng.choice([1, 2, 3, 4, 5, 8, 13, 16] if small else [1, 2, 3, 5, 8, 13, 31, 64, 100])
    if wide:
        B = rng.randint(1, max(1, 128 // H))
    prec = "fp32" if small else rng.choice(["fp32", "f16x3"])
    t = rng.randint(0, 19)
    try:
        state = synth.synth_unet_state(td, dim, mults, seed=100 + it, affine_jitter=0.3)
        w = {k: torch.from_numpy(v) for k, v in state.items()}
        unet = TemporalUnet(td, dim=dim, dim_mults=mults); unet.load_state_dict(w); unet.precision = prec
        diff = GaussianDiffusion(unet, H, td - 1, 1, n_timesteps=20).to(dev)
        x = torch.from_numpy(synth.normal_like(200 + it, "fuzz", (B, H, td)))
        with torch.no_grad():
            want = orc.unet_forward(w, x, torch.full((B,), t, dtype=torch.long))
        got = diff.model(x.to(dev), t); torch.cuda.synchronize()
        err = float((got.cpu() - want).abs().max())
        # and a short conditioned sampling loop with injected noise (<= 2e-5, the loop tolerance)
        T = rng.randint(3, 12)
        Bl = min(B, 8)
        noise = torch.from_numpy(synth.normal_like(300 + it, "fuzz.noise", (T + 1, Bl, H, td)))
        cond = torch.from_numpy(synth.uniform(300 + it, "fuzz.cond", (1, td), 0.9))
        want_loop = orc.sample_loop(w, orc.schedule_buffers("cosine", 20), noise, T, {0: cond})
        eng = diff._engine(dev)
        xl = noise[0].to(dev).clone()
        xl[:, 0] = cond.to(dev)
        eng.sample_loop(xl, T, noise_stack=noise[1:].to(dev).contiguous(), cond0=cond.to(dev))
        torch.cuda.synchronize()
        errl = float((xl.cpu() - want_loop).abs().max())
        flag = "" if err <= 5e-6 and errl <= 2e-5 else "   <<<<<< FAIL"
        if flag: bad += 1
        plan = eng.small_batch_plan(B)
        print(f"{it:3d} dim={dim} mults={mults} H={H} td={td} B={B} {prec} t={t} cc(launches, wide)={plan}: fwd {err:.2e} loop(T={T}) {errl:.2e}{flag}", flush=True)
        del unet, diff
    except Exception as e:
        msg = str(e)[:110]
        print(f"{it:3d} dim={dim} mults={mults} H={H} td={td} B={B} {prec}: refused/err: {msg}", flush=True)
print("failures:", bad)
