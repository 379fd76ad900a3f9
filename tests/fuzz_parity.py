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
shows how many launches of the case took them and how many the streamed-weight form.
`knobs` (round 3): kernel_size 3 / 5 / 7, level widths that need zero-padded GroupNorm groups (dim 8 / 24 / 40 /
48 / 56 / 96), horizon up to 128; every third case also runs loss.backward() through the engine against the
oracle's autograd (2e-5 x max|g| per tensor) where the net can be trained.
`grads`: fp32 only, gradients in every case (every dim and mults of `knobs`: padded widths and shrinking mults train).
Round 3: seeds 61 / 62 / 63 `knobs` (150 cases ran: 116 on padded widths, 78 with kernel_size 3 / 7; 4 refusals — an
identity residual over a concat with padded groups), seeds 71 / 72 `grads` (72 nets incl. widths up to 2048, horizons up to 128, kernel_size 3 / 7: worst gradient error
5.5e-6 x max|g|), seeds 81 / 82 `knobs` and 91 `grads` with horizons 12 / 24 / 40 / 48 / 96 / 100 (zero-padded rows,
gradients included): 0 failures.  Seeds 111 / 112 `grads` once padded WIDTHS train (130 cases, 93 of them on dims 8 / 24 / 40 /
48 / 56 / 96 and mults incl. 3): worst gradient error 1.1e-5 x max|g|, 0 failures.  Final library of the round: seeds 121 `small`
(32), 122 `wide`, 123 default (28), 124 `knobs` (42), 131 `grads` (53 nets, 11 of them with shrinking mults — identity residual
over a concat — trained): 0 failures; one dim-8 net (one-channel groups over 6 positions) passes by the fp64 criterion
(HIP 3.3e-6 from the float64 forward, the fp32 oracle 6.2e-6).  Seed 132 `grads` (57 nets): two flagged, both at horizon 8 with
kernel_size 7 on padded nets whose deepest level keeps ONE or TWO real positions — GroupNorm over 2 .. 12 values, where fp32
itself is short: dim 8 (1, 1, 3): the oracle's own fp32 gradients are 1e-3 x max|g| from its float64 gradients (the HIP
gradients pass by the fp64 criterion, the 4-step loop is 3.5e-5 off); dim 24 (1, 1, 1, 4): oracle fp32 3.4e-5 from float64
on ups.0.0 / mid_block1, HIP 1.7e-4 from the oracle on the same tensors (d x 1.5e-5) — about five times the oracle's
error, recorded as OPEN (ill-conditioned statistics amplify the convs' different summation order; no tensor is wrong
by more than 2e-4 and every better-conditioned net of the sweep is within 1.1e-5).  DAD_FUZZ_VERBOSE=1 prints the worst tensors."""
import sys, random
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import denoiser as orc
from tests.util import grad_scales
from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
from dynamics_aware_diffusion_amd.utils import synth
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
small = len(sys.argv) > 3 and sys.argv[3] in ("small", "wide")   # small batches, fp32: the consumer-combine kernels
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"              # ... on nets of 1024+ channels: conv_ccw
knobs = len(sys.argv) > 3 and sys.argv[3] in ("knobs", "grads")  # kernel sizes, padded widths, long horizons, gradients
grads = len(sys.argv) > 3 and sys.argv[3] == "grads"            # trainable nets only, loss.backward() in every case
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    dim = rng.choice([32, 64, 128, 256] + ([8, 24, 40, 48, 56, 96] * 2 if knobs else []))
    nlev = rng.choice([1, 2, 3, 4])
    mults = tuple([1] + [rng.choice([1, 2, 4, 8] + ([3] if knobs else [])) for _ in range(nlev - 1)])
    H = rng.choice([8, 16, 32, 64] + ([128, 12, 24, 40, 48, 96, 100] if knobs else []))
    ks = rng.choice([3, 5, 5, 7]) if knobs else 5
    if ((H >> (nlev - 1) < 4 or H & (H - 1)) and not knobs) or H % (1 << (nlev - 1)) or max(mults) * dim > 2048:
        continue
    if wide and (max(mults) * dim < 1024 or H > 32):
        continue
    td = rng.randint(2, 24)
    B = rng.choice([1, 2, 3, 4, 5, 8, 13, 16] if small else [1, 2, 3, 5, 8, 13, 31, 64, 100])
    if wide:
        B = rng.randint(1, max(1, 128 // H))
    prec = "fp32" if small or grads else rng.choice(["fp32", "f16x3"])
    t = rng.randint(0, 19)
    try:
        state = synth.synth_unet_state(td, dim, mults, seed=100 + it, affine_jitter=0.3, kernel_size=ks)
        w = {k: torch.from_numpy(v) for k, v in state.items()}
        unet = TemporalUnet(td, dim=dim, dim_mults=mults, kernel_size=ks); unet.load_state_dict(w); unet.precision = prec
        diff = GaussianDiffusion(unet, H, td - 1, 1, n_timesteps=20).to(dev)
        x = torch.from_numpy(synth.normal_like(200 + it, "fuzz", (B, H, td)))
        with torch.no_grad():
            want = orc.unet_forward(w, x, torch.full((B,), t, dtype=torch.long))
            got = diff.model(x.to(dev), t); torch.cuda.synchronize()
        err = float((got.cpu() - want).abs().max())
        note = ""
        if err > 5e-6:
            # ill-conditioned nets (GroupNorm groups of one channel over a few positions, dim 8): the fp64 criterion of
            # the parity tests — no farther from the float64 forward than twice the fp32 oracle is
            truth = orc.unet_forward(orc.cast_weights(w, torch.float64), x.double(), torch.full((B,), t, dtype=torch.long))
            e_hip = float((got.cpu().double() - truth).abs().max()); e_ref = float((want.double() - truth).abs().max())
            if e_hip <= 2 * e_ref + 5e-7: err, note = 0.0, f" [fwd by the fp64 criterion: hip {e_hip:.1e}, oracle {e_ref:.1e}]"
        # and a short conditioned sampling loop with injected noise (<= 2e-5, the loop tolerance)
        T = rng.randint(3, 12)
        Bl = min(B, 8)
        noise = torch.from_numpy(synth.normal_like(300 + it, "fuzz.noise", (T + 1, Bl, H, td)))
        cond = torch.from_numpy(synth.uniform(300 + it, "fuzz.cond", (1, td), 0.9))
        want_loop = orc.sample_loop(w, orc.schedule_buffers("cosine", 20), noise, T, {0: cond})
        eng = diff._engine(dev)
        xl = noise[0].to(dev).clone()
        xl[:, 0] = cond.to(dev)
        with torch.no_grad():
            eng.sample_loop(xl, T, noise_stack=noise[1:].to(dev).contiguous(), cond0=cond.to(dev))
        torch.cuda.synchronize()
        errl = float((xl.cpu() - want_loop).abs().max())
        plan = eng.small_batch_plan(B)
        # long horizons: the first reverse steps amplify an eps error ~100x; fall back to the fp64 criterion of the
        # parity tests (no farther from the float64 loop than twice the fp32 oracle is)
        if errl > 2e-5 and H >= 64:
            s64 = {k: v.double() for k, v in orc.schedule_buffers("cosine", 20).items()}
            truth = orc.sample_loop(orc.cast_weights(w, torch.float64), s64, noise.double(), T, {0: cond.double()})
            e_hip = float((xl.cpu().double() - truth).abs().max()); e_ref = float((want_loop.double() - truth).abs().max())
            if e_hip <= 2 * e_ref + 5e-7: errl = 0.0
        gtxt = ""
        gerr = 0.0
        shrinking = any(b < a for a, b in zip(mults, mults[1:]))
        if knobs and (grads or it % 3 == 0) and prec == "fp32" and td != dim:
            Bg = min(B, 6)
            x0 = torch.from_numpy(np.clip(synth.normal_like(400 + it, "fuzz.x0", (Bg, H, td)) * 0.5, -1, 1).astype(np.float32))
            tt = torch.from_numpy(np.array([(3 * i + 1) % 20 for i in range(Bg)], dtype=np.int64))
            nz = torch.from_numpy(synth.normal_like(400 + it, "fuzz.nz", (Bg, H, td)))
            for p_ in diff.parameters(): p_.grad = None
            with torch.enable_grad():
                x_t = diff.q_sample(x0.to(dev), tt.to(dev), nz.to(dev)).detach().requires_grad_(True)
                ((diff.model(x_t, tt.to(dev)) - nz.to(dev)) ** 2).mean().backward()
            torch.cuda.synchronize()
            _, og, odx = orc.training_gradients(w, orc.schedule_buffers("cosine", 20), x0, tt, nz)
            gerr = float((x_t.grad.cpu() - odx).abs().max()) / max(float(odx.abs().max()), 1e-12)
            gscale = grad_scales(og)          # (max|g| per tensor; a conv bias in front of a one-channel group is exactly zero)
            for k_, p_ in diff.model.named_parameters():
                gerr = max(gerr, float((p_.grad.cpu() - og[k_]).abs().max()) / gscale[k_])
            if gerr > 2e-5 and os.environ.get("DAD_FUZZ_VERBOSE"):
                worst_ = sorted(((float((p_.grad.cpu() - og[k_]).abs().max()) / gscale[k_], k_) for k_, p_ in diff.model.named_parameters()), reverse=True)[:8]
                print("      d x:", float((x_t.grad.cpu() - odx).abs().max()) / max(float(odx.abs().max()), 1e-12), "worst tensors:", worst_, flush=True)
            if gerr > 2e-5:                  # the same criterion per gradient tensor
                s64 = {k_: v_.double() for k_, v_ in orc.schedule_buffers("cosine", 20).items()}
                _, t64, tdx = orc.training_gradients(orc.cast_weights(w, torch.float64), s64, x0.double(), tt, nz.double())
                sc = grad_scales(t64)
                ok = float((x_t.grad.cpu().double() - tdx).abs().max()) <= 2 * float((odx.double() - tdx).abs().max()) + 2e-6 * float(tdx.abs().max())
                for k_, p_ in diff.model.named_parameters():
                    ok = ok and float((p_.grad.cpu().double() - t64[k_]).abs().max()) <= 2 * float((og[k_].double() - t64[k_]).abs().max()) + 2e-6 * sc[k_]
                if ok: note += f" [grads by the fp64 criterion, vs oracle {gerr:.1e}]"; gerr = 0.0
            gtxt = f" grads {gerr:.2e}"
        flag = "" if err <= 5e-6 and errl <= 2e-5 and gerr <= 2e-5 else "   <<<<<< FAIL"
        if flag: bad += 1
        print(f"{it:3d} dim={dim} mults={mults} H={H} k={ks} td={td} B={B} {prec} t={t} padded={eng.padded} cc(launches, wide)={plan}: fwd {err:.2e} loop(T={T}) {errl:.2e}{gtxt}{note}{flag}", flush=True)
        del unet, diff
    except Exception as e:
        msg = str(e)[:110]
        print(f"{it:3d} dim={dim} mults={mults} H={H} td={td} B={B} {prec}: refused/err: {msg}", flush=True)
print("failures:", bad)
