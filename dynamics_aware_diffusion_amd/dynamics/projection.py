"""``ProjectionMatrixBuilder`` — API mirror of
``m_diffuser.dynamics.projection.ProjectionMatrixBuilder``
(/root/reference/m_diffuser/dynamics/projection.py:11-133).

One-off (once per policy): builds the lifted map F of the linear system
x_{t+1} = A x_t + B u_t over a horizon and the orthogonal projector P = F F^+ onto
dynamically consistent trajectories, in float64, returned as an fp32 torch tensor that
``DynamicsAwarePolicy`` uploads for the projection kernel (``dad_project``).
Trajectory ordering is the reference's concatenated one: [x_0..x_H, u_0..u_{H-1}].

``get_projection_matrix(horizon)`` is the reference's host path (numpy ``pinv``, the golden
vectors pin it).  ``get_projection_matrix(horizon, device="cuda")`` builds the same P on the GPU
(SURVEY 8(f) rank 3): the pseudo-inverse is O(D^3) and D = (H+1) n + H m reaches 2183 for the Door
task (n=39, m=28, H=32), seconds of numpy against milliseconds of rocSOLVER.  F has full column
rank whenever it is built from this recursion (its last H m rows and first n rows hold identity
blocks), so P = Q Q^T with Q from the thin QR of F — no SVD; float64 throughout, cast once.
"""
from __future__ import annotations

import numpy as np
import torch


class ProjectionMatrixBuilder:
    def __init__(self, A: np.ndarray, B: np.ndarray, state_dim: int, action_dim: int):
        A = np.asarray(A, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        assert A.shape == (state_dim, state_dim), f"A shape mismatch: {A.shape}"
        assert B.shape == (state_dim, action_dim), f"B shape mismatch: {B.shape}"
        self.A, self.B = A, B
        self.state_dim, self.action_dim = state_dim, action_dim

    def _build_F_matrix(self, horizon: int) -> np.ndarray:
        """tau = F [x_0; u_0; ...; u_{H-1}]  with  x_t = A^t x_0 + sum_{s<t} A^{t-s-1} B u_s.

        Shape ((H+1) n + H m, n + H m).  Built row-block by row-block with the recursion
        row_{t+1} = A row_t (+ B in the column block of u_t), then the identity for the inputs.
        """
        n, m, H = self.state_dim, self.action_dim, horizon
        cols = n + H * m
        F = np.zeros(((H + 1) * n + H * m, cols))
        block = np.zeros((n, cols))
        block[:, :n] = np.eye(n)
        F[:n] = block
        for t in range(H):
            block = self.A @ block
            block[:, n + t * m:n + (t + 1) * m] += self.B
            F[(t + 1) * n:(t + 2) * n] = block
        F[(H + 1) * n:, n:] = np.eye(H * m)
        return F

    def get_projection_matrix(self, horizon: int, device=None) -> torch.Tensor:
        if device is not None and torch.device(device).type != "cpu":
            return self.projection_matrix_on_device(horizon, device)
        F = self._build_F_matrix(horizon)
        P = F @ np.linalg.pinv(F)
        return torch.from_numpy(P).float()

    def projection_matrix_on_device(self, horizon: int, device) -> torch.Tensor:
        """P = Q Q^T, Q = thin-QR(F), in float64 on ``device``; returns fp32 on that device."""
        dev = torch.device(device)
        n, m, H = self.state_dim, self.action_dim, horizon
        A = torch.as_tensor(self.A, dtype=torch.float64, device=dev)
        B = torch.as_tensor(self.B, dtype=torch.float64, device=dev)
        cols = n + H * m
        F = torch.zeros(((H + 1) * n + H * m, cols), dtype=torch.float64, device=dev)
        block = torch.zeros((n, cols), dtype=torch.float64, device=dev)
        block[:, :n] = torch.eye(n, dtype=torch.float64, device=dev)
        F[:n] = block
        for t in range(H):
            block = A @ block
            block[:, n + t * m:n + (t + 1) * m] += B
            F[(t + 1) * n:(t + 2) * n] = block
        F[(H + 1) * n:, n:] = torch.eye(H * m, dtype=torch.float64, device=dev)
        Q, R = torch.linalg.qr(F, mode="reduced")
        d = R.diagonal().abs()
        if float(d.min()) <= 1e-12 * float(d.max()):      # cannot happen for this F; be loud if it does
            raise RuntimeError("lifted dynamics map is rank deficient; use the host pinv path")
        return (Q @ Q.T).float()

    def verify_projection(self, P: torch.Tensor) -> bool:
        """P is idempotent to 1e-4 (projection.py:122-133)."""
        return torch.allclose(P @ P, P, atol=1e-4)
