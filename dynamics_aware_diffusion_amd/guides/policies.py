"""Planning policies — API mirror of ``m_diffuser.guides.policies``
(/root/reference/m_diffuser/guides/policies.py:13-485) driving the HIP sampler.

Behaviour kept from the reference, quirks included:
  * conditions overwrite the whole transition (action channels too) at their horizon step,
    in place, before the loop and after every step (policies.py:48-63,109-110,137-138);
  * guidance differentiates the guide at x_t and scales by the posterior VARIANCE
    (policies.py:87-97); the gradient itself stays in PyTorch autograd (the value model is an
    arbitrary ``nn.Module``), only the axpy is fused into the posterior kernel;
  * ``action_horizon = a`` buffers ``min(a + 1, H)`` actions starting at horizon step 0
    (policies.py:181-191);
  * ``DynamicsAwarePolicy`` does NOT project during ``sample_loop`` as shipped
    (SURVEY.md F5).  ``project_during_sampling=True`` is this build's opt-in for the
    README's x_{i-1} = project(denoise(x_i)).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .._engine import ProjectionState
from ..models.temporal_unet import TemporalUnet


class GuidedPolicy(nn.Module):
    """Conditioned (inpainting) sampler with optional guide and an action FIFO."""

    def __init__(self, diffusion_model, normalizer, guide_fn: Optional[Callable] = None,
                 guide_weight: float = 1.0, action_horizon: Optional[int] = None):
        super().__init__()
        self.diffusion = diffusion_model
        self.normalizer = normalizer
        self.guide_fn = guide_fn
        self.guide_weight = guide_weight
        self.horizon = diffusion_model.horizon
        self.observation_dim = diffusion_model.observation_dim
        self.action_dim = diffusion_model.action_dim
        self.transition_dim = diffusion_model.transition_dim
        self.action_horizon = 1 if action_horizon is None else action_horizon
        self.action_buffer: List[np.ndarray] = []

    # ------------------------------------------------------------------ conditioning
    def apply_conditions(self, x: torch.Tensor, conditions: Dict[int, torch.Tensor]) -> torch.Tensor:
        """In-place ``x[:, k] = value`` for every ``{k: value}`` (policies.py:48-63)."""
        for step, value in conditions.items():
            x[:, step] = value
        return x

    def _guided(self) -> bool:
        return self.guide_fn is not None and self.guide_weight > 0

    def _guide_gradient(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        probe = x.detach().requires_grad_(True)
        with torch.enable_grad():
            score = self.guide_fn(probe, t)
            (grad,) = torch.autograd.grad(score.sum(), probe)
        return grad.detach().contiguous()

    @staticmethod
    def _split_conditions(conditions, engine_ok_only: bool = False):
        """Horizon-step-0 condition goes into the fused kernel; any other step is applied
        with torch indexing afterwards."""
        cond0, rest = None, {}
        if conditions:
            for k, v in conditions.items():
                if k == 0:
                    cond0 = v
                else:
                    rest[k] = v
        return cond0, rest

    # ------------------------------------------------------------------ one reverse step
    @torch.no_grad()
    def p_sample_with_guidance(self, x: torch.Tensor, t: torch.Tensor,
                               conditions: Optional[Dict[int, torch.Tensor]] = None) -> torch.Tensor:
        """Mean (+ guidance) + noise, then inpainting (policies.py:65-112)."""
        step = TemporalUnet.shared_timestep(t)
        self.diffusion._check_step(step)
        eng = self.diffusion._engine(x.device)
        out = x.contiguous().float().clone()
        grad = None
        if self._guided():
            tt = t if torch.is_tensor(t) else torch.full((x.shape[0],), step, device=x.device,
                                                         dtype=torch.long)
            grad = self._guide_gradient(x, tt).float()
        noise = torch.randn_like(out)
        cond0, rest = self._split_conditions(conditions)
        c0 = None if cond0 is None else cond0.to(out.device, torch.float32).contiguous()
        eng.denoise_step(out, step, noise=noise, cond0=c0, guide_grad=grad,
                         guide_weight=float(self.guide_weight) if grad is not None else 0.0)
        if rest:
            self.apply_conditions(out, rest)
        return out

    # ------------------------------------------------------------------ full loop
    @torch.no_grad()
    def sample_loop(self, batch_size: int = 1, conditions: Optional[Dict[int, torch.Tensor]] = None,
                    verbose: bool = False, row_offset: int = 0) -> torch.Tensor:
        """x_T ~ N(0, I) -> apply conditions -> T guided/inpainted reverse steps
        (policies.py:114-149)."""
        diff = self.diffusion
        device = diff.betas.device
        shape = (batch_size, self.horizon, self.transition_dim)
        n_steps = int(diff.n_timesteps)
        diff._check_step(n_steps - 1)
        eng = diff._engine(device)
        philox = diff.sampler_rng == "philox"
        # the replayed-graph path needs persistent (address-stable) buffers; it is taken only for
        # the fused loop below, so decide it before anything is allocated
        _, rest_probe = self._split_conditions(conditions)
        graph = bool(diff.use_graph) and not self._guided() and not rest_probe
        if philox:
            x = eng.persistent("x", shape) if graph else \
                torch.empty(shape, device=device, dtype=torch.float32)
            eng.fill_normal(x, diff.seed, row_offset=row_offset, draw=0)
        else:
            x = torch.randn(shape, device=device)
            if graph:
                x = eng.persistent("x", shape).copy_(x)
        if conditions is not None:
            conditions = {k: v.to(device, torch.float32) for k, v in conditions.items()}
            x = self.apply_conditions(x, conditions)
        cond0, rest = self._split_conditions(conditions)
        c0 = None if cond0 is None else cond0.contiguous()
        if graph and c0 is not None:                       # frozen pointer: stage the condition
            c0 = eng.persistent("cond0", tuple(c0.shape)).copy_(c0)
        projector = self._loop_projector()

        if not self._guided() and not rest:
            alphas = None
            if projector is not None:
                alphas = [self._get_projection_alpha(i) for i in range(int(diff.betas.shape[0]))]
            if philox:
                eng.sample_loop(x, n_steps, seed=diff.seed, row_offset=row_offset, cond0=c0,
                                projection=projector, proj_alphas=alphas, use_graph=graph)
            elif diff.noise_stack_bytes(shape, n_steps) > diff.max_noise_stack_bytes:
                # the whole z stack would not be O(1) in T (1.1 GB for Door at B=128, T=1000): draw z
                # step by step instead — the same torch.randn calls in the same order, one engine
                # step per iteration
                for i in reversed(range(n_steps)):
                    eng.denoise_step(x, i, noise=torch.randn(shape, device=device), cond0=c0)
                    if projector is not None:
                        projector.apply(x, alphas[i])
                return x.clone() if graph else x
            else:
                stack = eng.persistent("z", (n_steps,) + shape) if graph else \
                    torch.empty((n_steps,) + shape, device=device)
                for j in range(n_steps):
                    torch.randn(shape, out=stack[j])
                eng.sample_loop(x, n_steps, noise_stack=stack, cond0=c0, projection=projector,
                                proj_alphas=alphas, use_graph=graph)
            return x.clone() if graph else x

        # guided (or multi-step-conditioned) path: one engine step per iteration, the guide
        # gradient comes from PyTorch autograd on the user's value model.
        for j, i in enumerate(reversed(range(n_steps))):
            grad, gw = None, 0.0
            if self._guided():
                tt = torch.full((batch_size,), i, device=device, dtype=torch.long)
                grad = self._guide_gradient(x, tt).float()
                gw = float(self.guide_weight)
            if philox:
                eng.denoise_step(x, i, seed=diff.seed, row_offset=row_offset, draw=j + 1, cond0=c0,
                                 guide_grad=grad, guide_weight=gw)
            else:
                eng.denoise_step(x, i, noise=torch.randn_like(x), cond0=c0, guide_grad=grad,
                                 guide_weight=gw)
            if rest:
                self.apply_conditions(x, rest)
            if projector is not None:
                projector.apply(x, self._get_projection_alpha(i))
        return x

    def _loop_projector(self) -> Optional[ProjectionState]:
        return None

    # ------------------------------------------------------------------ planner glue
    def _process_observation(self, observation) -> np.ndarray:
        """Flatten gym observations, goal-conditioned dicts included (policies.py:151-179)."""
        if isinstance(observation, dict):
            if "observation" in observation and "desired_goal" in observation:
                state, goal = observation["observation"], observation["desired_goal"]
                wants = self.normalizer.obs_mean.shape[0]
                observation = np.concatenate([state, goal]) if wants == len(state) + len(goal) \
                    else state
            elif "observation" in observation:
                observation = observation["observation"]
            elif "achieved_goal" in observation:
                observation = observation["achieved_goal"]
            else:
                observation = np.concatenate([np.asarray(v).flatten() for v in observation.values()])
        return np.asarray(observation).reshape(1, -1)

    def _fill_action_buffer(self, trajectory: torch.Tensor) -> None:
        """Queue actions of horizon steps 0..min(action_horizon, H-1) of plan 0, un-normalised
        (policies.py:181-191)."""
        plan = trajectory[0].cpu().numpy()
        lo, hi = self.observation_dim, self.observation_dim + self.action_dim
        for step in range(min(self.action_horizon + 1, self.horizon)):
            action = self.normalizer.unnormalize_actions(plan[step, lo:hi].reshape(1, -1))
            self.action_buffer.append(action.flatten())

    def get_action(self, observation, **kwargs) -> np.ndarray:
        """Pop a buffered action, replanning (one B=1 sample_loop) when the FIFO is empty
        (policies.py:193-223)."""
        if self.action_buffer:
            return self.action_buffer.pop(0)
        device = self.diffusion.betas.device
        obs = self.normalizer.normalize_observations(self._process_observation(observation))
        start = torch.zeros(1, self.transition_dim, device=device)
        start[:, :self.observation_dim] = torch.as_tensor(obs, dtype=torch.float32).to(device)
        plan = self.sample_loop(batch_size=1, conditions={0: start}, verbose=False)
        self._fill_action_buffer(plan)
        return self.action_buffer.pop(0)


    # ------------------------------------------------------------------ batched planner glue
    def get_actions(self, observations) -> np.ndarray:
        """``get_action`` for N environments at once (not in the
        reference): observations ``(N, observation_dim)`` -> actions ``(N, action_dim)``.

        One B=N sampling loop with a per-row inpainting condition replaces N separate B=1 loops;
        normalisation, the condition build and the action un-normalisation stay on the device, and
        only the ``(N, k, action_dim)`` block of buffered actions crosses PCIe.  Every environment
        follows the single-environment semantics: actions of horizon steps
        0..min(action_horizon, H-1) are queued and served before the next replan
        (policies.py:181-223).  All N queues are refilled together.
        """
        obs = np.asarray(observations, dtype=np.float32)
        if obs.ndim != 2 or obs.shape[1] != self.observation_dim:
            raise ValueError(f"observations must be (N, {self.observation_dim}), got {obs.shape}")
        n = obs.shape[0]
        queue = getattr(self, "_batched_actions", None)
        if queue is None or queue.shape[0] != n or self._batched_cursor >= queue.shape[1]:
            device = self.diffusion.betas.device
            mean = torch.as_tensor(self.normalizer.obs_mean, dtype=torch.float32, device=device)
            std = torch.as_tensor(self.normalizer.obs_std, dtype=torch.float32, device=device)
            a_mean = torch.as_tensor(self.normalizer.action_mean, dtype=torch.float32, device=device)
            a_std = torch.as_tensor(self.normalizer.action_std, dtype=torch.float32, device=device)
            start = torch.zeros(n, self.transition_dim, device=device)
            start[:, :self.observation_dim] = (torch.from_numpy(obs).to(device) - mean) / std
            plans = self.sample_loop(batch_size=n, conditions={0: start}, verbose=False)
            keep = min(self.action_horizon + 1, self.horizon)
            lo, hi = self.observation_dim, self.observation_dim + self.action_dim
            self._batched_actions = (plans[:, :keep, lo:hi] * a_std + a_mean).cpu().numpy()
            self._batched_cursor = 0
        out = self._batched_actions[:, self._batched_cursor].copy()
        self._batched_cursor += 1
        return out


class MPCPolicy(GuidedPolicy):
    """Plan once, execute ``action_horizon`` actions, replan (policies.py:226-240)."""

    def __init__(self, diffusion_model, normalizer, action_horizon: int = 8):
        super().__init__(diffusion_model, normalizer, action_horizon=action_horizon)


class ValueGuidedPolicy(GuidedPolicy):
    """Guide = sum over the horizon of value_model(observations) (policies.py:243-271)."""

    def __init__(self, diffusion_model, normalizer, value_model: nn.Module,
                 guide_weight: float = 1.0, action_horizon: Optional[int] = None):
        obs_dim = diffusion_model.observation_dim

        def guide_fn(x, t):
            return value_model(x[:, :, :obs_dim]).sum(dim=1)

        super().__init__(diffusion_model, normalizer, guide_fn, guide_weight, action_horizon)
        self.value_model = value_model


class DynamicsAwarePolicy(GuidedPolicy):
    """Sampler with the dynamics projector x <- a * (x P) + (1 - a) * x available as
    ``apply_projection`` (policies.py:274-485)."""

    def __init__(self, diffusion_model, projection_matrix: Optional[torch.Tensor] = None,
                 normalizer=None, state_dim: int = 4, observation_dim: int = 4,
                 action_dim: int = 2, horizon: int = 16, projection_schedule: str = "constant",
                 projection_strength: float = 1.0, action_horizon: Optional[int] = None,
                 project_during_sampling: bool = False):
        if action_horizon is None:
            action_horizon = horizon
        super().__init__(diffusion_model=diffusion_model, normalizer=normalizer, guide_fn=None,
                         guide_weight=0.0, action_horizon=action_horizon)
        self.projection_matrix = projection_matrix
        self.state_dim = state_dim
        self.observation_dim = observation_dim
        self.action_dim = action_dim
        self.horizon = horizon
        self.projection_schedule = projection_schedule
        self.projection_strength = projection_strength
        self.project_during_sampling = project_during_sampling
        self.n_timesteps = diffusion_model.n_timesteps
        self.device = diffusion_model.betas.device
        if normalizer is not None:
            self.obs_mean = torch.from_numpy(np.asarray(normalizer.obs_mean)).float().to(self.device)
            self.obs_std = torch.from_numpy(np.asarray(normalizer.obs_std)).float().to(self.device)
            self.action_mean = torch.from_numpy(np.asarray(normalizer.action_mean)).float().to(self.device)
            self.action_std = torch.from_numpy(np.asarray(normalizer.action_std)).float().to(self.device)
        else:
            self.obs_mean = self.obs_std = self.action_mean = self.action_std = None
        self._projector: Optional[ProjectionState] = None

    def _get_projection_alpha(self, t: int) -> float:
        """Annealing of the projection strength (policies.py:358-383)."""
        progress = t / self.n_timesteps
        kind = self.projection_schedule
        if kind == "constant":
            return self.projection_strength
        if kind == "linear":
            return self.projection_strength * (1 - progress)
        if kind == "quadratic":
            return self.projection_strength * (1 - progress) ** 2
        if kind == "noise_schedule":
            # same fp32 arithmetic as the reference's torch.sqrt(1 - betas[t]).item(), evaluated
            # on a host copy of the schedule (one device read per schedule, not one sync per t)
            betas = self.diffusion.betas
            key = (betas.data_ptr(), betas._version, int(betas.shape[0]))
            if getattr(self, "_betas_host_key", None) != key:
                b = betas.detach().to("cpu", torch.float32).numpy()
                # numpy's float32 sqrt is the IEEE instruction (correctly rounded, like the 0-d
                # torch op of the reference); torch's vectorised CPU sqrt is not on every host
                self._betas_host = np.sqrt(np.float32(1.0) - b)
                self._betas_host_key = key
            return float(self._betas_host[t]) * self.projection_strength
        raise ValueError(f"Unknown projection schedule: {kind}")

    def _projection_state(self) -> ProjectionState:
        if self._projector is None:
            if self.obs_std.shape[0] != self.state_dim or self.observation_dim != self.state_dim:
                # the reference broadcasts length-od statistics against an n-slice and fails
                # the same way (policies.py:436-439,389)
                raise RuntimeError(
                    f"The size of tensor a ({self.state_dim}) must match the size of tensor b "
                    f"({self.obs_std.shape[0]}) at non-singleton dimension 2")
            self._projector = ProjectionState(
                self.projection_matrix, self.obs_mean, self.obs_std, self.action_mean,
                self.action_std, self.state_dim, self.observation_dim, self.action_dim,
                self.diffusion.betas.device)
        return self._projector

    def _loop_projector(self) -> Optional[ProjectionState]:
        if not self.project_during_sampling:
            return None
        if self.projection_matrix is None or self.normalizer is None:
            return None
        return self._projection_state()

    def unnormalize_states(self, s):
        return s if self.obs_mean is None else s * self.obs_std + self.obs_mean

    def unnormalize_actions(self, a):
        return a if self.action_mean is None else a * self.action_std + self.action_mean

    def normalize_states(self, s):
        return s if self.obs_mean is None else (s - self.obs_mean) / self.obs_std

    def normalize_actions(self, a):
        return a if self.action_mean is None else (a - self.action_mean) / self.action_std

    def apply_projection(self, x: torch.Tensor, t: int) -> torch.Tensor:
        """Project a normalised trajectory batch onto the dynamics-consistent subspace in
        physical units (policies.py:409-485); returns a new tensor."""
        if self.projection_matrix is None or self.normalizer is None:
            return x
        alpha = self._get_projection_alpha(t)
        if alpha <= 0:
            return x
        out = x.contiguous().float().clone()
        self._projection_state().apply(out, alpha)
        return out
