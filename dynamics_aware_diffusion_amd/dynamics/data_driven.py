"""Least-squares system identification — API mirror of
``m_diffuser.dynamics.data_driven`` (/root/reference/m_diffuser/dynamics/data_driven.py:75-165)
for transitions that are already in memory.

The reference pulls transitions out of a Minari dataset (``extract_transitions``,
data_driven.py:11-72); datasets, minari and gymnasium are outside this build (out of
scope), so the entry points here take the ``(states, actions, next_states)`` arrays directly.
``fit_linear_dynamics`` solves  x_{t+1} = A x_t + B u_t  for Theta = [A B]^T exactly as the
reference does (``lstsq`` on Phi = [X U]), in float64; given a ROCm device it runs the solve there
through the normal equations' QR (torch.linalg.lstsq, one-off).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch


def fit_linear_dynamics(states, actions, next_states, state_dim: Optional[int] = None,
                        device=None, return_quality: bool = False):
    """(N, sd), (N, m), (N, sd) -> A (n, n), B (n, m) float64 numpy (data_driven.py:75-134).

    ``state_dim`` keeps the first n observation columns (PointMaze: drop the goal).
    ``return_quality`` adds {"r_squared", "mean_prediction_error"} — what the reference prints.
    """
    dev = torch.device(device) if device is not None else torch.device("cpu")
    X = torch.as_tensor(np.asarray(states), dtype=torch.float64, device=dev)
    U = torch.as_tensor(np.asarray(actions), dtype=torch.float64, device=dev)
    Y = torch.as_tensor(np.asarray(next_states), dtype=torch.float64, device=dev)
    if X.ndim != 2 or U.ndim != 2 or Y.shape != X.shape or U.shape[0] != X.shape[0]:
        raise ValueError(f"transition arrays do not line up: states {tuple(X.shape)}, actions "
                         f"{tuple(U.shape)}, next_states {tuple(Y.shape)}")
    if state_dim is not None and X.shape[1] > state_dim:
        X, Y = X[:, :state_dim], Y[:, :state_dim]
    n, m = X.shape[1], U.shape[1]
    Phi = torch.cat([X, U], dim=1)                                   # (N, n+m)
    if Phi.shape[0] < n + m:
        raise ValueError(f"{Phi.shape[0]} transitions cannot determine {n + m} regressors")
    if dev.type == "cpu":
        Theta = torch.from_numpy(np.linalg.lstsq(Phi.numpy(), Y.numpy(), rcond=None)[0])
    else:
        Q, R = torch.linalg.qr(Phi, mode="reduced")                  # tall-skinny, well within fp64
        Theta = torch.linalg.solve_triangular(R, Q.T @ Y, upper=True)
    A = Theta[:n].T.contiguous().cpu().numpy()
    B = Theta[n:].T.contiguous().cpu().numpy()
    if not return_quality:
        return A, B
    res = Y - Phi @ Theta.to(dev)
    ss_res = float((res ** 2).sum())
    ss_tot = float(((Y - Y.mean(dim=0)) ** 2).sum())
    quality = {"r_squared": 1.0 - ss_res / ss_tot if ss_tot > 0 else float("nan"),
               "mean_prediction_error": float(res.norm(dim=1).mean())}
    return A, B, quality


def transitions_from_episodes(observations, actions, state_dim: Optional[int] = None):
    """Episodes [(T_i+1, sd) observations, (T_i, m) actions] -> stacked (s_t, a_t, s_{t+1})
    (the loop of data_driven.py:51-62 without the dataset object)."""
    S, A, S1 = [], [], []
    for obs, act in zip(observations, actions):
        obs, act = np.asarray(obs), np.asarray(act)
        if obs.shape[0] != act.shape[0] + 1:
            raise ValueError(f"an episode needs T+1 observations for T actions, got {obs.shape[0]} / {act.shape[0]}")
        S.append(obs[:-1]); A.append(act); S1.append(obs[1:])
    S, A, S1 = np.concatenate(S), np.concatenate(A), np.concatenate(S1)
    if state_dim is not None:
        S, S1 = S[:, :state_dim], S1[:, :state_dim]
    return S, A, S1


def identify_dynamics_from_arrays(states, actions, next_states, state_dim: Optional[int] = None,
                                  device=None) -> Tuple[np.ndarray, np.ndarray, int, int]:
    """(A, B, state_dim, action_dim) like ``identify_dynamics_from_data`` (data_driven.py:137-165)."""
    A, B = fit_linear_dynamics(states, actions, next_states, state_dim, device=device)
    return A, B, A.shape[0], B.shape[1]


def identify_dynamics_from_data(dataset_name: str, state_dim: Optional[int] = None,
                                max_trajectories: int = 1000):
    raise ImportError(
        "identify_dynamics_from_data needs the minari dataset package, which is not part of this "
        "build; load the episodes yourself and call identify_dynamics_from_arrays / "
        "transitions_from_episodes")
