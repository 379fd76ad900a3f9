"""``GaussianDiffusion`` — API mirror of ``m_diffuser.models.diffusion.GaussianDiffusion``
(/root/reference/m_diffuser/models/diffusion.py:51-294) for the sampling path.

Schedules are computed with the same fp32 torch ops, in the same order, as the reference
(diffusion.py:32-48, 104-128), so the 12 registered buffers are bit-identical and a
reference checkpoint's buffers load unchanged.  ``p_mean_variance`` / ``p_sample`` /
``p_sample_loop`` run on the HIP engine; the per-step tail (x0 prediction, clamp, posterior
mean, noise, inpainting) is one fused kernel behind ``dad_denoise_step``.  ``loss`` evaluates the
training objective on the same kernels and is differentiable (explicit backward pass of the engine).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from .temporal_unet import TemporalUnet

_ENGINE_BUFFERS = ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                   "posterior_mean_coef1", "posterior_mean_coef2",
                   "posterior_log_variance_clipped")


def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> torch.Tensor:
    """Nichol & Dhariwal cosine schedule, clipped to [1e-4, 0.9999] (diffusion.py:32-41)."""
    grid = torch.linspace(0, timesteps, timesteps + 1)
    abar = torch.cos(((grid / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    abar = abar / abar[0]
    return torch.clip(1 - (abar[1:] / abar[:-1]), 0.0001, 0.9999)


def linear_beta_schedule(timesteps: int, beta_start: float = 1e-4,
                         beta_end: float = 0.02) -> torch.Tensor:
    """DDPM linear schedule (diffusion.py:44-48)."""
    return torch.linspace(beta_start, beta_end, timesteps)


def make_schedule(beta_schedule: str, n_timesteps: int) -> Dict[str, torch.Tensor]:
    """All 12 schedule buffers, in registration order (diffusion.py:96-128)."""
    if beta_schedule == "linear":
        betas = linear_beta_schedule(n_timesteps)
    elif beta_schedule == "cosine":
        betas = cosine_beta_schedule(n_timesteps)
    else:
        raise ValueError(f"Unknown beta schedule: {beta_schedule}")
    alphas = 1.0 - betas
    abar = torch.cumprod(alphas, dim=0)
    abar_prev = torch.cat([torch.ones(1), abar[:-1]])
    post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
    bufs = {"betas": betas, "alphas": alphas, "alphas_cumprod": abar,
            "alphas_cumprod_prev": abar_prev}
    bufs["sqrt_alphas_cumprod"] = torch.sqrt(abar)
    bufs["sqrt_one_minus_alphas_cumprod"] = torch.sqrt(1.0 - abar)
    bufs["sqrt_recip_alphas_cumprod"] = torch.sqrt(1.0 / abar)
    bufs["sqrt_recipm1_alphas_cumprod"] = torch.sqrt(1.0 / abar - 1)
    bufs["posterior_variance"] = post_var
    bufs["posterior_log_variance_clipped"] = torch.log(torch.clamp(post_var, min=1e-20))
    bufs["posterior_mean_coef1"] = betas * torch.sqrt(abar_prev) / (1.0 - abar)
    bufs["posterior_mean_coef2"] = (1.0 - abar_prev) * torch.sqrt(alphas) / (1.0 - abar)
    return bufs


def extract(a: torch.Tensor, t: torch.Tensor, x_shape: tuple) -> torch.Tensor:
    """Per-row schedule lookup broadcastable to ``x_shape`` (diffusion.py:15-29); raises
    ``RuntimeError`` when ``t`` exceeds the schedule, like the reference's gather."""
    picked = a.gather(-1, t)
    return picked.reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


class GaussianDiffusion(nn.Module):
    """DDPM over trajectories (batch, horizon, observation_dim + action_dim).

    Constructor, attributes (``horizon``, ``observation_dim``, ``action_dim``,
    ``transition_dim``, mutable ``n_timesteps``, ``betas`` ...) and method names follow the
    reference (diffusion.py:62-136) because ``scripts/evaluate.py`` and the policies read
    them directly.  Extra, build-specific knob: ``sampler_rng``:

    * ``"torch"`` (default) draws x_T and every z with ``torch.randn`` in the reference's call
      order (diffusion.py:241,218), so a seeded run consumes the device generator exactly
      like the reference would on the same device;
    * ``"philox"`` uses the engine's in-kernel counter-based generator (no noise tensors,
      sharding-invariant) — the throughput path.

    ``use_graph = True`` replays the whole T-step loop as one cached hipGraph (persistent
    input buffers, result returned as a copy): removes the per-launch host cost that dominates
    small batches such as the B=1 plans of ``get_action``.
    """

    def __init__(self, model: TemporalUnet, horizon: int, observation_dim: int, action_dim: int,
                 n_timesteps: int = 1000, loss_type: str = "l2", clip_denoised: bool = True,
                 predict_epsilon: bool = True, beta_schedule: str = "cosine"):
        super().__init__()
        self.model = model
        self.horizon = horizon
        self.observation_dim = observation_dim
        self.action_dim = action_dim
        self.transition_dim = observation_dim + action_dim
        self.n_timesteps = n_timesteps
        self.clip_denoised = clip_denoised
        self.predict_epsilon = predict_epsilon
        self.beta_schedule = beta_schedule
        for name, buf in make_schedule(beta_schedule, n_timesteps).items():
            self.register_buffer(name, buf)
        if loss_type == "l1":
            self.loss_fn = nn.L1Loss(reduction="none")
        elif loss_type == "l2":
            self.loss_fn = nn.MSELoss(reduction="none")
        else:
            raise ValueError(f"Unknown loss type: {loss_type}")
        self.sampler_rng = "torch"
        self.seed = 0
        self.use_graph = False
        # "torch" RNG: the fused loop takes the whole z stack (T, B, H, td) up to this many bytes;
        # beyond it z is drawn step by step (same draws, same order, O(1) memory in T)
        self.max_noise_stack_bytes = 256 << 20

    @staticmethod
    def noise_stack_bytes(shape, n_steps: int) -> int:
        n = 4 * int(n_steps)
        for d in shape:
            n *= int(d)
        return n

    # ------------------------------------------------------------------ engine access
    def _engine(self, device: torch.device):
        sched = {k: getattr(self, k) for k in _ENGINE_BUFFERS}
        self.model.bind_diffusion(sched, int(self.betas.shape[0]), self.predict_epsilon,
                                  self.clip_denoised)
        return self.model.engine(self.horizon, device)

    def _check_step(self, t: int) -> None:
        if t < 0 or t >= int(self.betas.shape[0]):
            # reference: RuntimeError raised by gather in extract() (diffusion.py:28)
            raise RuntimeError(f"index {t} is out of bounds for dimension 0 with size "
                               f"{int(self.betas.shape[0])}")

    # ------------------------------------------------------------------ closed forms
    def q_sample(self, x_start: torch.Tensor, t: torch.Tensor,
                 noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """q(x_t | x_0) (diffusion.py:138-157)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        return (extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def predict_start_from_noise(self, x_t, t, noise):
        """x0 = sqrt(1/abar_t) x_t - sqrt(1/abar_t - 1) eps (diffusion.py:159-166)."""
        return (extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def q_posterior(self, x_start, x_t, t) -> Tuple[torch.Tensor, torch.Tensor]:
        """(posterior mean, clipped log variance) (diffusion.py:168-180)."""
        mean = (extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return mean, extract(self.posterior_log_variance_clipped, t, x_t.shape)

    # ------------------------------------------------------------------ reverse process
    @torch.no_grad()
    def p_mean_variance(self, x: torch.Tensor, t: Union[int, torch.Tensor]
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(model mean, (B,1,1) log variance) of p(x_{t-1} | x_t) (diffusion.py:182-203)."""
        step = TemporalUnet.shared_timestep(t)
        self._check_step(step)
        eng = self._engine(x.device)
        xc = x.contiguous().float()
        mean = torch.empty_like(xc)
        eng.denoise_step(xc, step, mean_out=mean, update_x=False)
        logvar = self.posterior_log_variance_clipped[step].reshape(1, 1, 1).expand(
            x.shape[0], 1, 1)
        return mean, logvar

    @torch.no_grad()
    def p_sample(self, x: torch.Tensor, t: Union[int, torch.Tensor]) -> torch.Tensor:
        """x_{t-1} ~ p(. | x_t); z is drawn even at t == 0 and masked (diffusion.py:205-223)."""
        step = TemporalUnet.shared_timestep(t)
        self._check_step(step)
        eng = self._engine(x.device)
        out = x.contiguous().float().clone()
        noise = torch.randn_like(out)
        eng.denoise_step(out, step, noise=noise)
        return out

    @torch.no_grad()
    def p_sample_loop(self, shape: tuple, verbose: bool = False, row_offset: int = 0
                      ) -> torch.Tensor:
        """Full ancestral sampling t = n_timesteps-1 .. 0 from pure noise (diffusion.py:225-251).

        ``n_timesteps`` may have been lowered after construction (evaluate.py:350-353): the
        loop then walks the first ``n_timesteps`` entries of the trained schedule.
        """
        device = self.betas.device
        eng = self._engine(device)
        n_steps = int(self.n_timesteps)
        self._check_step(n_steps - 1)
        if self.sampler_rng == "philox":
            x = eng.persistent("x", shape) if self.use_graph else \
                torch.empty(shape, device=device, dtype=torch.float32)
            eng.fill_normal(x, self.seed, row_offset=row_offset, draw=0)
            eng.sample_loop(x, n_steps, seed=self.seed, row_offset=row_offset,
                            use_graph=self.use_graph)
            return x.clone() if self.use_graph else x
        x = torch.randn(shape, device=device)
        if self.noise_stack_bytes(shape, n_steps) > self.max_noise_stack_bytes:
            # O(1) in T: z drawn per step with the reference's own call order (diffusion.py:218)
            for i in reversed(range(n_steps)):
                eng.denoise_step(x, i, noise=torch.randn(tuple(shape), device=device))
            return x
        if self.use_graph:
            x = eng.persistent("x", shape).copy_(x)
        stack = eng.persistent("z", (n_steps,) + tuple(shape)) if self.use_graph else \
            torch.empty((n_steps,) + tuple(shape), device=device)
        for j in range(n_steps):
            torch.randn(tuple(shape), out=stack[j])
        eng.sample_loop(x, n_steps, noise_stack=stack, use_graph=self.use_graph)
        return x.clone() if self.use_graph else x

    # ------------------------------------------------------------------ training objective
    def loss(self, x_start: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The diffusion training objective (diffusion.py:253-290): t ~ randint per trajectory,
        noise ~ randn, x_t = q_sample, denoiser on the HIP engine (per-row time embedding),
        elementwise L1 / L2 against the noise (or x_0), optional weights, mean.

        Same random draws in the same order as the reference (``torch.randint`` then
        ``torch.randn_like`` on ``x_start``'s device).  With gradients enabled the result carries an
        autograd graph whose denoiser node is the engine's explicit backward pass
        (``dad_unet_backward``): ``loss.backward()`` fills ``.grad`` of every parameter, as the
        reference's training step expects (utils/training.py:152-156).  Under ``torch.no_grad()`` it
        is the forward-only validation loss."""
        batch = x_start.shape[0]
        self._engine(x_start.device)              # bind schedule / options before the model call
        t = torch.randint(0, self.n_timesteps, (batch,), device=x_start.device).long()
        noise = torch.randn_like(x_start)
        x_noisy = self.q_sample(x_start, t, noise)
        model_output = self.model(x_noisy, t)
        target = noise if self.predict_epsilon else x_start
        loss = self.loss_fn(model_output, target)
        if weights is not None:
            loss = loss * weights
        return loss.mean()

    def forward(self, x, *args, **kwargs):
        return self.loss(x, *args, **kwargs)
