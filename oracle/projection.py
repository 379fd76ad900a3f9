"""Oracle (test infrastructure — see oracle/__init__.py): dynamics projection on the CPU.

Follows /root/reference/m_diffuser/dynamics/projection.py:43-120 (lifted map F and
P = F F^+) and /root/reference/m_diffuser/guides/policies.py:358-485 (annealing alpha
and the gather -> de-normalise -> x@P -> blend -> re-normalise -> scatter step).
Parity pinned by tests/golden/projection_*.npz.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

Tensor = torch.Tensor


def lifted_map(A: np.ndarray, B: np.ndarray, horizon: int) -> np.ndarray:
    """F with rows [x_0..x_H, u_0..u_{H-1}] and columns [x_0, u_0..u_{H-1}]
    (dynamics/projection.py:53-83): top-left A^t, top-right A^{t-tau-1} B (tau < t),
    bottom-right identity.  float64."""
    n, m = B.shape
    assert A.shape == (n, n)                                   # projection.py:35-36
    H = horizon
    F = np.zeros(((H + 1) * n + H * m, n + H * m))
    Apow = np.eye(n)
    for t in range(H + 1):
        F[t * n:(t + 1) * n, :n] = Apow
        if t < H:
            Apow = Apow @ A
    AkB = [B]
    for _ in range(H - 1):
        AkB.append(A @ AkB[-1])
    for t in range(1, H + 1):
        for tau in range(t):
            F[t * n:(t + 1) * n, n + tau * m:n + (tau + 1) * m] = AkB[t - tau - 1]
    F[(H + 1) * n:, n:] = np.eye(H * m)
    return F


def projection_matrix(A: np.ndarray, B: np.ndarray, horizon: int) -> Tensor:
    """P = F pinv(F) in float64, returned as fp32 torch (projection.py:98-120)."""
    F = lifted_map(np.asarray(A, dtype=np.float64), np.asarray(B, dtype=np.float64), horizon)
    P = F @ np.linalg.pinv(F)
    return torch.from_numpy(P).float()


def double_integrator(dt: float):
    """The analytical PointMaze (A, B) the reference writes down
    (dynamics/projection.py:143-156; dynamics/extractor.py:93-133)."""
    A = np.array([[1, 0, dt, 0], [0, 1, 0, dt], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    B = np.array([[0.5 * dt ** 2, 0], [0, 0.5 * dt ** 2], [dt, 0], [0, dt]], dtype=np.float64)
    return A, B


def projection_alpha(schedule: str, strength: float, t: int, n_timesteps: int,
                     betas: Optional[Tensor] = None) -> float:
    """policies.py:358-383."""
    progress = t / n_timesteps
    if schedule == "constant":
        return strength
    if schedule == "linear":
        return strength * (1 - progress)
    if schedule == "quadratic":
        return strength * (1 - progress) ** 2
    if schedule == "noise_schedule":
        return torch.sqrt(1 - betas[t]).item() * strength
    raise ValueError(f"Unknown projection schedule: {schedule}")


def apply_projection(x: Tensor, P: Tensor, alpha: float, state_dim: int, observation_dim: int,
                     obs_mean: Tensor, obs_std: Tensor, act_mean: Tensor, act_std: Tensor
                     ) -> Tensor:
    """policies.py:409-485 with alpha already evaluated.  x (B,H,od+ad) normalised.

    v = [s_0..s_{H-1}, s_{H-1}, a_0..a_{H-1}] in physical units; v <- a*(v@P)+(1-a)*v;
    rows s_0..s_{H-1} and the actions are re-normalised and written back; the duplicated
    last state is dropped.  Like the reference, the de-normalise multiplies an n-slice by
    the full length-od statistics, so od != n raises.
    """
    if alpha <= 0:
        return x
    Bsz, H, _ = x.shape
    obs = x[:, :, :observation_dim]
    act = x[:, :, observation_dim:]
    s = obs[:, :, :state_dim] * obs_std + obs_mean
    a = act * act_std + act_mean
    s_ext = torch.cat([s, s[:, -1:, :]], dim=1)
    v = torch.cat([s_ext.reshape(Bsz, -1), a.reshape(Bsz, -1)], dim=1)
    v = alpha * (v @ P) + (1 - alpha) * v
    n_s = (H + 1) * state_dim
    s = v[:, :n_s].reshape(Bsz, H + 1, state_dim)[:, :-1, :]
    a = v[:, n_s:].reshape(Bsz, H, -1)
    s = (s - obs_mean) / obs_std
    a = (a - act_mean) / act_std
    if observation_dim != state_dim:
        pad = torch.zeros(Bsz, H, observation_dim - state_dim, dtype=s.dtype)
        s = torch.cat([s, pad], dim=-1)
    return torch.cat([s, a], dim=-1)


def projection_violation(x: Tensor, P: Tensor, observation_dim: int, obs_mean: Tensor, obs_std: Tensor,
                         act_mean: Tensor, act_std: Tensor) -> Tensor:
    """ProjectionLoss.compute (m_diffuser/losses/__init__.py:93-186): de-normalise, concatenate
    [s_0..s_{H-1}, s_{H-1}, a_0..a_{H-1}] (ALL observation channels are the state there), project,
    mean squared distance."""
    s = x[:, :, :observation_dim] * obs_std + obs_mean
    a = x[:, :, observation_dim:] * act_std + act_mean
    ext = torch.cat([s, s[:, -1:, :]], dim=1)
    v = torch.cat([ext.reshape(x.shape[0], -1), a.reshape(x.shape[0], -1)], dim=1)
    return torch.mean((v - v @ P) ** 2)
