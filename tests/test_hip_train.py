"""GPU parity of the backward pass: ``loss.backward()`` through the HIP engine
(training forward + dad_unet_backward) against the reference's own autograd — every parameter gradient
and d loss / d x_t.

Fixtures: ``grads_*.npz`` hold, per parameter, a strided sample of the reference's gradient plus its sum,
sum of squares and max |g| (tests/golden/make_golden.py::gen_grads); the oracle's autograd over its
restatement of the forward is pinned to the same fixtures on the CPU (test_oracle_golden.py) and gives
the element-by-element comparison here.  Gate: |g_hip - g_ref| <= 2e-5 * max|g_ref| per tensor.
"""
import numpy as np
import pytest
import torch

from tests.golden import cases
from tests.test_hip_parity import build, dev, injected_noise  # noqa: F401  (dev: fixture)
from tests.util import golden, grad_scales, max_abs, net_weights_torch

pytestmark = pytest.mark.gpu

REL = 2e-5


def _loss_and_backward(diff, name, net, T, B, weighted, devc):
    x0, t, noise, wts = cases.train_inputs(name, net, T, B, weighted)
    tt = torch.from_numpy(t).to(devc)
    for p in diff.parameters():
        p.grad = None
    real_randint = torch.randint
    torch.randint = lambda *a, **k: tt.clone()
    try:
        with injected_noise(noise[None], devc), torch.enable_grad():
            loss = diff.loss(torch.from_numpy(x0).to(devc), None if wts is None else torch.from_numpy(wts).to(devc))
            assert loss.requires_grad
            loss.backward()
    finally:
        torch.randint = real_randint
    torch.cuda.synchronize()
    return loss


@pytest.mark.parametrize("case", cases.GRAD_CASES, ids=lambda c: c[0])
def test_parameter_gradients_vs_reference(case, dev):
    from oracle import denoiser as orc
    name, net, T, B, loss_type, pred_eps, weighted = case
    g = golden(name)
    diff = build(net, T, "cosine", dev, loss_type=loss_type, predict_epsilon=pred_eps)
    loss = _loss_and_backward(diff, name, net, T, B, weighted, dev)
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * max(1.0, abs(float(g["loss"])))
    # the oracle's autograd: every element of every tensor
    x0, t, noise, wts = cases.train_inputs(name, net, T, B, weighted)
    _, og, odx = orc.training_gradients(net_weights_torch(net), orc.schedule_buffers("cosine", T), torch.from_numpy(x0),
                                        torch.from_numpy(t), torch.from_numpy(noise), loss_type, pred_eps,
                                        None if wts is None else torch.from_numpy(wts))
    worst, worst_key = 0.0, None
    params = dict(diff.model.named_parameters())
    assert set(params) == set(og)
    for k, p in params.items():
        assert p.grad is not None, f"no gradient reached {k}"
        got = p.grad.detach().cpu().numpy()
        assert np.isfinite(got).all(), k
        scale = max(float(g["max." + k]), 1e-12)
        flat = got.reshape(-1)
        idx = cases.grad_sample_index(flat.size)
        e_ref = float(np.max(np.abs(flat[idx].astype(np.float64) - g["g." + k]))) / scale      # the reference itself
        e_orc = max_abs(got, og[k].numpy()) / scale                                            # every element
        e_sum = abs(float(flat.astype(np.float64).sum()) - float(g["sum." + k])) / (scale * max(1.0, np.sqrt(flat.size)))
        if max(e_ref, e_orc) > worst:
            worst, worst_key = max(e_ref, e_orc), k
        assert e_ref <= REL and e_orc <= REL, f"{k}: rel err vs reference {e_ref:.2e}, vs oracle {e_orc:.2e}"
        assert e_sum <= REL, f"{k}: sum of the gradient off by {e_sum:.2e} (relative to max|g| sqrt(n))"
    print(f"{name}: worst parameter-gradient error {worst:.2e} x max|g| ({worst_key})")


@pytest.mark.parametrize("case", cases.GRAD_CASES[:2], ids=lambda c: c[0])
def test_input_gradient_and_repeatability(case, dev):
    """d loss / d x_t through a direct model call with a leaf input, per-row timesteps; a second backward
    pass reproduces the first bit for bit (fixed-order reductions, no atomics)."""
    name, net, T, B, loss_type, pred_eps, weighted = case
    g = golden(name)
    diff = build(net, T, "cosine", dev, loss_type=loss_type, predict_epsilon=pred_eps)
    x0, t, noise, wts = cases.train_inputs(name, net, T, B, weighted)
    x0t, tt, nz = torch.from_numpy(x0).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(noise).to(dev)
    runs = []
    for _ in range(2):
        for p in diff.parameters():
            p.grad = None
        with torch.enable_grad():
            x_t = diff.q_sample(x0t, tt, nz).detach().requires_grad_(True)
            out = diff.model(x_t, tt)
            per = diff.loss_fn(out, nz if pred_eps else x0t)
            if wts is not None:
                per = per * torch.from_numpy(wts).to(dev)
            per.mean().backward()
        torch.cuda.synchronize()
        runs.append((x_t.grad.cpu().numpy(), {k: p.grad.cpu().numpy().copy() for k, p in diff.model.named_parameters()}))
    scale = float(np.abs(g["dx"]).max())
    err = max_abs(runs[0][0], g["dx"]) / scale
    print(f"{name}: d loss / d x_t error {err:.2e} x max|g|")
    assert err <= REL
    assert np.array_equal(runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert np.array_equal(runs[0][1][k], runs[1][1][k]), k


def test_composed_loss_backward(dev):
    """ComposedLoss([DiffusionLoss, ProjectionLoss]).backward() — the reference's composed training step
    (losses/__init__.py:189-227, utils/training.py:130-156): the projection term depends on the data
    only, so the parameter gradients equal the diffusion term's."""
    from dynamics_aware_diffusion_amd.dynamics import ProjectionMatrixBuilder, double_integrator
    from dynamics_aware_diffusion_amd.losses import ComposedLoss, DiffusionLoss, ProjectionLoss
    import contextlib
    import io
    name, net, T, B, loss_type, pred_eps, weighted = cases.GRAD_CASES[0]
    g = golden(name)
    diff = build(net, T, "cosine", dev, loss_type=loss_type, predict_epsilon=pred_eps)
    A, Bm = double_integrator(0.1)
    with contextlib.redirect_stdout(io.StringIO()):
        P = ProjectionMatrixBuilder(A, Bm, 4, 2).get_projection_matrix(cases.H)
        terms = ComposedLoss([DiffusionLoss(diff, 1.0),
                              ProjectionLoss(P, cases.NormalizerStub(4, 2), state_dim=4, action_dim=2, observation_dim=4,
                                             horizon=cases.H, weight=0.1, device=str(dev))])
    x0, t, noise, _ = cases.train_inputs(name, net, T, B, weighted)
    tt = torch.from_numpy(t).to(dev)
    for p in diff.parameters():
        p.grad = None
    real_randint = torch.randint
    torch.randint = lambda *a, **k: tt.clone()
    try:
        with injected_noise(noise[None], dev), torch.enable_grad():
            total, parts = terms({"conditions": torch.from_numpy(x0).to(dev)})
            total.backward()
    finally:
        torch.randint = real_randint
    torch.cuda.synchronize()
    assert abs(parts["diffusion"] - float(g["loss"])) <= 2e-6 * max(1.0, abs(float(g["loss"])))
    assert abs(parts["total"] - (parts["diffusion"] + parts["projection"])) <= 1e-6 * max(1.0, abs(parts["total"]))
    for k, p in diff.model.named_parameters():
        flat = p.grad.cpu().numpy().reshape(-1)
        idx = cases.grad_sample_index(flat.size)
        assert float(np.max(np.abs(flat[idx] - g["g." + k]))) <= REL * max(float(g["max." + k]), 1e-12), k


def test_training_refusals(dev):
    """What cannot be trained is refused with a message, not run wrongly: the split-f16 arithmetic."""
    from dynamics_aware_diffusion_amd._engine import DadError
    diff = build("tiny", 20, "cosine", dev)
    diff.model.precision = "f16x3"
    try:
        with torch.enable_grad(), pytest.raises(DadError, match="fp32"):
            diff.model(torch.zeros(2, cases.H, 6, device=dev), torch.zeros(2, dtype=torch.long, device=dev))
    finally:
        diff.model.precision = "fp32"


@pytest.mark.parametrize("net", ["tiny4", "tiny_d48"])   # (tiny_d48: zero-padded widths — the padded copies are rebuilt on the device too)
def test_sgd_steps_track_the_oracle_without_leaving_the_device(net, dev):
    """Three optimiser steps (utils/training.py:152-166: loss, backward, step) on the HIP engine against the
    same three steps of torch autograd on the oracle.  After every step the engine re-derives its packed
    images on the device (dad_model_refresh_weights: forward images, data-gradient images, time tables) —
    the engine object must survive the steps — and the sampler's inference kernels must see the new weights."""
    from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
    from oracle import denoiser as orc
    T, B, lr = 20, 5, 0.05
    od, ad, td, dim, mults = cases.net_dims(net)
    unet = TemporalUnet(td, dim=dim, dim_mults=mults)
    unet.load_state_dict({k: torch.from_numpy(v) for k, v in cases.net_weights(net).items()})
    diff = GaussianDiffusion(unet, cases.H, od, ad, n_timesteps=T).to(dev)
    opt = torch.optim.SGD(diff.model.parameters(), lr=lr)
    w = {k: v.clone() for k, v in net_weights_torch(net).items()}
    sched = orc.schedule_buffers("cosine", T)
    engines = set()
    for step in range(3):
        x0, t, noise, _ = cases.train_inputs(f"sgd.{step}", net, T, B, False)
        tt = torch.from_numpy(t).to(dev)
        real_randint = torch.randint
        torch.randint = lambda *a, **k: tt.clone()
        try:
            with injected_noise(noise[None], dev), torch.enable_grad():
                opt.zero_grad()
                loss = diff.loss(torch.from_numpy(x0).to(dev))
                loss.backward()
                opt.step()
        finally:
            torch.randint = real_randint
        engines.add(id(diff.model._engine))
        ol, og, _ = orc.training_gradients(w, sched, torch.from_numpy(x0), torch.from_numpy(t), torch.from_numpy(noise))
        w = {k: v - lr * og[k] for k, v in w.items()}
        assert abs(float(loss) - float(ol)) <= 5e-6 * max(1.0, abs(float(ol))), (step, float(loss), float(ol))
    torch.cuda.synchronize()
    assert len(engines) == 1, "the engine was rebuilt instead of refreshed"
    for k, p in diff.model.named_parameters():
        assert max_abs(p.detach().cpu().numpy(), w[k].numpy()) <= 2e-5 * max(1.0, float(w[k].abs().max())), k
    # inference kernels (per-timestep tables, small-batch plan) on the updated weights
    x = torch.from_numpy(cases.forward_input("sgd.fwd", net, 3))
    with torch.no_grad():
        want = orc.unet_forward(w, x, torch.full((3,), 7, dtype=torch.long))
        got = diff.model(x.to(dev), 7)
    torch.cuda.synchronize()
    assert len({id(diff.model._engine)} | engines) == 1
    assert max_abs(got.cpu().numpy(), want.numpy()) <= 2e-5


@pytest.mark.parametrize("arch", [
    # (td, dim, mults, horizon, B)
    (5, 256, (1, 8), 8, 3),            # 2048 channels: 256-channel GroupNorm groups (<256,32> direct-B tile), L = 8 / 4
    (9, 128, (1, 8, 4), 16, 4),        # 1024 channels (128-channel groups, <128,64>); shrinking tail 1024 -> 512: ups.0.0 is an identity residual over the concat
    (7, 64, (1, 2, 4, 8), 32, 2),      # four levels, L down to 4: every down / up-sampling conv shape
    (6, 128, (1, 2), 128, 2),          # horizon 128: the <32,128> tiles in the training forward and the data gradients
    (6, 64, (1, 2, 4), 32, 5, 3),      # TemporalUnet(kernel_size=3) (temporal_unet.py:139): 3-tap forward, flipped 3-tap data gradient, wgrad<3>
    (7, 32, (1, 4), 16, 3, 7),         # kernel_size=7: halo of three rows per sample side, wgrad<7>
    (5, 256, (1, 8), 8, 2, 3),         # kernel_size=3 on 2048 channels: the LDS-staged <256,32> tile (the direct-B kernel is 5-tap only)
    (6, 64, (1, 2, 4), 24, 5),         # horizon 24 (zero-padded to 32): masked GroupNorm backward, zero-padded data gradients,
    (9, 32, (1, 2, 2, 4), 40, 3),      #   the trajectory and d loss / d out in their real shape; 40 on four levels
    (7, 32, (1, 2, 4, 8), 16, 6),      # the reference's default dim_mults at train.py's default horizon 16 (2 positions at the bottom)
    (6, 128, (1, 2), 100, 2, 3),       # horizon 100 with kernel_size 3
    (6, 96, (1, 2, 4), 32, 4),         # --dim 96: groups of 12 / 24 / 48 channels run zero-padded to 16 / 32 / 64 (utils/padding.py)
    (7, 40, (1, 3), 24, 3, 3),         # padded widths (40 -> 64, 120 -> 128) AND a padded horizon (24 -> 32), kernel_size 3
    (5, 8, (1, 2, 4, 8), 16, 5),       # dim 8: one real channel per group at level 0
    (6, 32, (1, 4, 2, 1), 32, 3),      # shrinking twice: two decoder blocks whose residual is the identity over [x | skip]
    (7, 64, (1, 2, 1), 24, 2, 3),      #   ... with a padded horizon and kernel_size 3
], ids=lambda a: f"td{a[0]}_d{a[1]}_m{'x'.join(map(str, a[2]))}_H{a[3]}_B{a[4]}" + (f"_k{a[5]}" if len(a) > 5 else ""))
def test_gradients_on_other_architectures_vs_oracle(arch, dev):
    """The backward pass beyond the fixture nets: wide GroupNorm groups (the direct-B forward tile keeps the
    pre-activation and statistics too), four levels, horizon 128 — every parameter gradient and dL/dx against
    the oracle's autograd (pinned to the reference's own gradients in test_oracle_golden.py)."""
    from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
    from dynamics_aware_diffusion_amd.utils import synth
    from oracle import denoiser as orc
    td, dim, mults, H, B = arch[:5]
    ks = arch[5] if len(arch) > 5 else 5
    T = 20
    state = synth.synth_unet_state(td, dim, mults, seed=19, affine_jitter=0.3, kernel_size=ks)
    w = {k: torch.from_numpy(v) for k, v in state.items()}
    unet = TemporalUnet(td, dim=dim, dim_mults=mults, kernel_size=ks)
    unet.load_state_dict(w)
    diff = GaussianDiffusion(unet, H, td - 1, 1, n_timesteps=T).to(dev)
    x0 = torch.from_numpy(np.clip(synth.normal_like(20, f"garch.x.{arch}", (B, H, td)) * 0.5, -1, 1).astype(np.float32))
    t = torch.from_numpy(np.array([(3 * i + 1) % T for i in range(B)], dtype=np.int64))
    noise = torch.from_numpy(synth.normal_like(20, f"garch.n.{arch}", (B, H, td)))
    with torch.enable_grad():
        x_t = diff.q_sample(x0.to(dev), t.to(dev), noise.to(dev)).detach().requires_grad_(True)
        out = diff.model(x_t, t.to(dev))
        ((out - noise.to(dev)) ** 2).mean().backward()
    torch.cuda.synchronize()
    _, og, odx = orc.training_gradients(w, orc.schedule_buffers("cosine", T), x0, t, noise)
    assert max_abs(x_t.grad.cpu().numpy(), odx.numpy()) <= REL * float(odx.abs().max())
    worst = 0.0
    scales = grad_scales(og)
    for k, p in diff.model.named_parameters():
        scale = scales[k]
        e = max_abs(p.grad.cpu().numpy(), og[k].numpy()) / scale
        worst = max(worst, e)
        assert e <= REL, f"{k}: {e:.2e} x max|g|"
    print(f"{arch}: worst parameter-gradient error {worst:.2e} x max|g|")
