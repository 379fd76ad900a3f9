"""Environment name -> physical state size and (A, B) — mirror of
``m_diffuser.dynamics.registry`` (/root/reference/m_diffuser/dynamics/registry.py:11-111) without
gym / minari: data comes in as arrays, the analytical PointMaze model is the reference's double
integrator (dynamics/extractor.py:93-133).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .data_driven import identify_dynamics_from_arrays

# registry.py:11-27
DYNAMICS_REGISTRY = {"pointmaze": "data_driven", "maze": "data_driven", "halfcheetah": "data_driven",
                     "hopper": "data_driven", "walker": "data_driven"}
STATE_DIM_REGISTRY = {"pointmaze": 4, "maze": 4, "halfcheetah": 17, "hopper": 11, "walker": 17}


def state_dim_for_env(env_name: str) -> Optional[int]:
    """Physical state size (goal etc. excluded), None when the name matches no pattern — the
    reference then uses the full observation (registry.py:84-88; Door: n = 39)."""
    low = env_name.lower()
    for pattern, dim in STATE_DIM_REGISTRY.items():
        if pattern in low:
            return dim
    return None


def double_integrator(dt: float = 0.1) -> Tuple[np.ndarray, np.ndarray]:
    """[x, y, vx, vy] / [ax, ay] (extractor.py:93-133)."""
    A = np.array([[1, 0, dt, 0], [0, 1, 0, dt], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    B = np.array([[0.5 * dt * dt, 0], [0, 0.5 * dt * dt], [dt, 0], [0, dt]], dtype=np.float64)
    return A, B


def get_dynamics_for_env(env_name: str, transitions=None, method: Optional[str] = None,
                         device=None) -> Tuple[np.ndarray, np.ndarray, int, int]:
    """(A, B, state_dim, action_dim) (registry.py:38-111).

    ``transitions`` = (states, actions, next_states) arrays for the data-driven fit; without them
    the maze family falls back to the analytical model like the reference does when it cannot find
    a dataset (registry.py:71-75), anything else is an error (the reference's simulator-based
    fallbacks need gymnasium)."""
    low = env_name.lower()
    if method is None:
        method = next((kind for pat, kind in DYNAMICS_REGISTRY.items() if pat in low), "numerical")
    if method == "data_driven" and transitions is not None:
        s, a, s1 = transitions
        return identify_dynamics_from_arrays(s, a, s1, state_dim_for_env(env_name), device=device)
    if method in ("data_driven", "analytical") and "maze" in low:
        A, B = double_integrator()
        return A, B, 4, 2
    raise ValueError(f"no dynamics for {env_name!r} with method {method!r}: pass transitions=(states, "
                     f"actions, next_states); simulator-based extraction is outside this build")
