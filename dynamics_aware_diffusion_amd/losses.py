"""Loss functions — API mirror of ``m_diffuser.losses``
(/root/reference/m_diffuser/losses/__init__.py:13-236).

``DiffusionLoss`` evaluates ``GaussianDiffusion.loss`` (the denoiser runs on the HIP engine with a
per-row time embedding; differentiable through the engine's explicit backward pass);
``ProjectionLoss`` measures the dynamics violation ``mean((v - vP)^2)`` of the DATA batch in physical
units with the projection kernel of the sampler — as in the reference it depends on no parameter, so
it contributes a value and no gradient; ``ComposedLoss`` adds weighted terms, and
``ComposedLoss(...)(batch)[0].backward()`` is the reference's composed training step.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn

from ._engine import ProjectionState


class BaseLoss(ABC, nn.Module):
    """Weighted scalar term over a batch dict with a ``'conditions'`` entry (losses:13-34)."""

    def __init__(self, weight: float = 1.0):
        super().__init__()
        self.weight = weight

    @abstractmethod
    def compute(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        ...

    def forward(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        return self.weight * self.compute(batch)


class DiffusionLoss(BaseLoss):
    """``diffusion.loss(batch['conditions'])`` (losses:37-47)."""

    def __init__(self, diffusion_model, weight: float = 1.0):
        super().__init__(weight)
        self.diffusion = diffusion_model

    def compute(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        return self.diffusion.loss(batch["conditions"])


class ProjectionLoss(BaseLoss):
    """Distance of (normalised) trajectories from the dynamics-consistent subspace, measured in
    physical units: ``mean((v - v P)^2)`` over the concatenated vector
    ``[s_0..s_{H-1}, s_{H-1}, a_0..a_{H-1}]`` (losses:50-186)."""

    def __init__(self, projection_matrix: torch.Tensor, normalizer, state_dim: int, action_dim: int,
                 observation_dim: int, horizon: int, weight: float = 0.1, device: str = "cuda"):
        super().__init__(weight)
        self.P = projection_matrix.to(device)
        self.normalizer = normalizer
        self.state_dim = state_dim
        self.action_dim = action_dim
        self.observation_dim = observation_dim
        self.horizon = horizon
        self.device = device
        self.obs_mean = torch.from_numpy(np.asarray(normalizer.obs_mean)).float().to(device)
        self.obs_std = torch.from_numpy(np.asarray(normalizer.obs_std)).float().to(device)
        self.action_mean = torch.from_numpy(np.asarray(normalizer.action_mean)).float().to(device)
        self.action_std = torch.from_numpy(np.asarray(normalizer.action_std)).float().to(device)
        if observation_dim != state_dim:
            # the reference concatenates ALL observation channels as the state (losses:93-96,
            # 161-175), so P must have been built for n = observation_dim
            state_dim = observation_dim
        self._state = ProjectionState(self.P, self.obs_mean, self.obs_std, self.action_mean,
                                      self.action_std, state_dim, observation_dim, action_dim, device)
        self._D = (horizon + 1) * state_dim + horizon * action_dim
        if tuple(self.P.shape) != (self._D, self._D):
            raise ValueError(f"projection matrix is {tuple(self.P.shape)}, expected ({self._D}, {self._D})")

    def extract_state_actions(self, trajectory: torch.Tensor):
        return trajectory[:, :, :self.observation_dim], trajectory[:, :, self.observation_dim:]

    def unnormalize_states(self, s):
        return s * self.obs_std + self.obs_mean

    def unnormalize_actions(self, a):
        return a * self.action_std + self.action_mean

    def to_concatenated(self, state: torch.Tensor, actions: torch.Tensor) -> torch.Tensor:
        ext = torch.cat([state, state[:, -1:, :]], dim=1)
        return torch.cat([ext.reshape(state.shape[0], -1), actions.reshape(actions.shape[0], -1)], dim=1)

    @torch.no_grad()
    def compute(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        x = batch["conditions"].contiguous().float()
        per_row = self._state.violation(x)                # sum_d (v - vP)^2 per trajectory
        return per_row.sum() / (x.shape[0] * self._D)


class ComposedLoss(nn.Module):
    """Sum of weighted terms + a per-term breakdown for logging (losses:189-226)."""

    def __init__(self, losses: List[BaseLoss]):
        super().__init__()
        self.losses = nn.ModuleList(losses)

    def forward(self, batch: Dict[str, torch.Tensor]):
        total, parts = 0.0, {}
        for term in self.losses:
            value = term(batch)
            total = total + value
            parts[term.__class__.__name__.replace("Loss", "").lower()] = value.detach().item()
        parts["total"] = total.detach().item()
        return total, parts


__all__ = ["BaseLoss", "DiffusionLoss", "ProjectionLoss", "ComposedLoss"]
