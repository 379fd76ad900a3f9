"""Case definitions shared by ``make_golden.py`` (runs the REAL reference, build container
only) and by the parity tests (oracle here, HIP path on the GPU box).

Only inputs are defined here, all from the portable generator
(``dynamics_aware_diffusion_amd/utils/synth.py``) — nothing in this file imports the
reference, so it travels to the GPU box.  Expected outputs live in the ``*.npz`` files next
to it.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np

from dynamics_aware_diffusion_amd.utils import synth

H = 32

SCHEDULE_CASES = [("cosine", 100), ("cosine", 500), ("cosine", 1000), ("linear", 100)]
SINUSOID_DIMS = [32, 128, 256]
SINUSOID_T = [0, 1, 50, 999]

# (name, kind, c_in, c_out, L, B)
UNIT_CASES = [
    ("cb_16_32_L8", "conv_block", 16, 32, 8, 3),
    ("cb_32_32_L32", "conv_block", 32, 32, 32, 2),
    ("rtb_16_32_L8", "res_block", 16, 32, 8, 3),
    ("rtb_32_32_L4", "res_block", 32, 32, 4, 5),
    ("rtb_6_32_L32", "res_block", 6, 32, 32, 2),
    ("down_32_L32", "down", 32, 32, 32, 2),
    ("down_16_L8", "down", 16, 16, 8, 3),
    ("up_32_L4", "up", 32, 32, 4, 3),
    ("up_16_L8", "up", 16, 16, 8, 2),
]
UNIT_TIME_DIM = 32


def unit_weights(name: str, kind: str, ci: int, co: int, seed: int = 11) -> "OrderedDict[str, np.ndarray]":
    """Weights of one isolated layer, keyed like the corresponding reference sub-module."""
    w: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def conv(base: str, a: int, b: int, k: int) -> None:
        bound = 1.0 / np.sqrt(a * k)
        w[base + ".weight"] = synth.uniform(seed, name + base + ".w", (b, a, k), bound)
        w[base + ".bias"] = synth.uniform(seed, name + base + ".b", (b,), bound)

    def norm(base: str, c: int) -> None:
        w[base + ".weight"] = 1.0 + synth.uniform(seed, name + base + ".g", (c,), 0.3)
        w[base + ".bias"] = synth.uniform(seed, name + base + ".be", (c,), 0.3)

    if kind == "conv_block":
        conv("block.0", ci, co, 5)
        norm("block.1", co)
    elif kind == "res_block":
        conv("blocks.0.block.0", ci, co, 5)
        norm("blocks.0.block.1", co)
        conv("blocks.1.block.0", co, co, 5)
        norm("blocks.1.block.1", co)
        bound = 1.0 / np.sqrt(UNIT_TIME_DIM)
        w["time_mlp.1.weight"] = synth.uniform(seed, name + "tw", (co, UNIT_TIME_DIM), bound)
        w["time_mlp.1.bias"] = synth.uniform(seed, name + "tb", (co,), bound)
        if ci != co:
            conv("residual_conv", ci, co, 1)
    elif kind == "down":
        conv("conv", ci, co, 3)
    elif kind == "up":
        bound = 1.0 / np.sqrt(co * 4)
        w["conv.weight"] = synth.uniform(seed, name + "conv.w", (ci, co, 4), bound)
        w["conv.bias"] = synth.uniform(seed, name + "conv.b", (co,), bound)
    else:
        raise ValueError(kind)
    return w


def unit_inputs(name: str, ci: int, L: int, B: int, seed: int = 12):
    x = synth.normal_like(seed, name + ".x", (B, ci, L))
    temb = synth.normal_like(seed, name + ".temb", (B, UNIT_TIME_DIM))
    return x, temb


# ---------------------------------------------------------------------------- full nets
# name: (obs_dim, act_dim, dim, mults, T_train, weight_seed, affine_jitter)
NETS = {
    "tiny": (4, 2, 32, (1, 2, 4), 20, 3, 0.25),
    "tiny4": (5, 3, 32, (1, 2, 2, 4), 20, 4, 0.25),      # 4 levels, L down to 4, td=8
    "tiny_td64": (4, 2, 32, (1, 2, 4), 20, 6, 0.25),     # time_dim = 64 (see TIME_DIMS)
    "pointmaze": (4, 2, 128, (1, 2, 4), 100, 0, 0.0),
    "halfcheetah": (17, 6, 256, (1, 4, 8), 1000, 0, 0.0),
    "door": (39, 28, 256, (1, 2, 4, 8), 1000, 0, 0.0),
    # the same three architectures with non-trivial GroupNorm affine parameters (gamma = 1 + 0.25 U,
    # beta = 0.25 U): the single-forward goldens use these, so that a gamma / beta indexing slip in the
    # 128- and 256-channel-group tiles cannot hide behind gamma = 1, beta = 0
    "pointmaze_j": (4, 2, 128, (1, 2, 4), 100, 8, 0.25),
    "halfcheetah_j": (17, 6, 256, (1, 4, 8), 1000, 8, 0.25),
    "door_j": (39, 28, 256, (1, 2, 4, 8), 1000, 8, 0.25),
    # TemporalUnet(kernel_size=3 / 7) (temporal_unet.py:139; see KERNEL_SIZES)
    "tiny_k3": (4, 2, 32, (1, 2, 4), 20, 9, 0.25),
    "tiny_k7": (5, 3, 64, (1, 2), 20, 10, 0.25),
    # widths GroupNorm(8, C) accepts and the conv tiles do not (temporal_unet.py:71: only C % 8 == 0 is needed):
    # 48 / 96 channels -> groups of 6 / 12, 24 / 48 / 96 -> 3 / 6 / 12; the engine runs them zero-padded
    "tiny_d48": (4, 2, 48, (1, 2), 20, 11, 0.25),
    "tiny_d24": (5, 3, 24, (1, 2, 4), 20, 12, 0.25),
}

# (case, net, B, t)  — single U-Net forward
FORWARD_CASES = [
    ("fwd_tiny", "tiny", 3, 7),
    ("fwd_tiny4", "tiny4", 5, 13),
    ("fwd_pointmaze", "pointmaze_j", 2, 63),
    ("fwd_halfcheetah", "halfcheetah_j", 2, 500),
    ("fwd_door", "door_j", 2, 999),
    ("fwd_tiny_k3", "tiny_k3", 3, 11),
    ("fwd_tiny_k7", "tiny_k7", 5, 2),
    ("fwd_tiny_d48", "tiny_d48", 3, 9),
    ("fwd_tiny_d24", "tiny_d24", 6, 15),
]

# Horizons every level can halve but that are not a power of two (temporal_unet.py:35-54 accepts any such length;
# the engine runs them zero-padded to the next power of two).  (case, net, horizon, B, t): a forward + its fp64 run,
# and a conditioned T-step loop of the net's own schedule length with injected noise.
HORIZON_CASES = [
    ("hz_tiny_H24", "tiny", 24, 3, 5),
    ("hz_tiny_H12", "tiny", 12, 5, 17),
    ("hz_tiny4_H40", "tiny4", 40, 2, 9),
]


def horizon_inputs(case: str, net: str, horizon: int, B: int, T: int):
    _, _, td, _, _ = net_dims(net)
    x = synth.normal_like(27, case + ".x", (B, horizon, td))
    noise = synth.normal_like(27, case + ".noise", (T + 1, B, horizon, td))
    return x, noise


# (case, net, T_train, n_sample_steps, B, conditioned, schedule)
LOOP_CASES = [
    ("loop_tiny_T20_B1", "tiny", 20, 20, 1, False, "cosine"),
    ("loop_tiny_T20_B4_cond", "tiny", 20, 20, 4, True, "cosine"),
    ("loop_tiny_T100_B4_cond", "tiny", 100, 100, 4, True, "cosine"),
    ("loop_tiny_T100_trunc50_B4", "tiny", 100, 50, 4, False, "cosine"),
    ("loop_tiny_linear_T20_B4", "tiny", 20, 20, 4, True, "linear"),
    ("loop_tiny4_T20_B3_cond", "tiny4", 20, 20, 3, True, "cosine"),
    ("loop_pointmaze_T100_B4_cond", "pointmaze", 100, 100, 4, True, "cosine"),
    ("loop_pointmaze_T100_B1_cond", "pointmaze", 100, 100, 1, True, "cosine"),   # get_action's B=1 plan
    ("loop_tiny_d48_T20_B3_cond", "tiny_d48", 20, 20, 3, True, "cosine"),        # zero-padded GroupNorm groups
]

# The T=1000 loops of BASELINE configs 4 and 5 (two plans each; the reference takes minutes on
# them).  Besides the final plans the fixtures hold x after LONG_TRACE iterations, so the CPU
# suite can pin the oracle on the first and the last few steps instead of all thousand.
LONG_LOOP_CASES = [
    ("loop_halfcheetah_T1000_B2", "halfcheetah", 1000, 1000, 2, False, "cosine"),
    ("loop_door_T1000_B2", "door", 1000, 1000, 2, True, "cosine"),
]
LONG_TRACE = (8, 500, 992)

# BASELINE config 3: the README's x_{i-1} = project(denoise(x_i)) — the reference's own
# p_sample_with_guidance and apply_projection alternated by the harness (the shipped sample_loop
# never calls apply_projection).  (case, net, T, B, schedule, strength)
PROJ_LOOP_CASES = [
    ("loop_pointmaze_T500_B4_proj", "pointmaze", 500, 4, "noise_schedule", 1.0),
    ("loop_tiny_T20_B3_proj_linear", "tiny", 20, 3, "linear", 0.7),
]

# GaussianDiffusion / TemporalUnet options off their defaults (diffusion.py:192-200;
# temporal_unet.py:139,152).  (case, net, T, B, predict_epsilon, clip_denoised)
OPTION_CASES = [
    ("opt_tiny_x0_clip", "tiny", 20, 3, False, True),
    ("opt_tiny_eps_noclip", "tiny", 20, 3, True, False),
    ("opt_tiny_x0_noclip", "tiny", 20, 3, False, False),
    ("opt_tinytd_eps_clip", "tiny_td64", 20, 3, True, True),       # time_dim = 64 != dim = 32
]

# (case, net, T, B, guide_weight)
GUIDE_CASES = [
    ("guide_tiny_w0p1", "tiny", 20, 3, 0.1),
    ("guide_tiny_w1", "tiny", 20, 3, 1.0),
    ("guide_pointmaze_w1", "pointmaze", 100, 3, 1.0),     # BASELINE architecture, full T=100 guided loop
]
# Guided loops truncated to a few steps (evaluate.py:350-353 semantics) on the widest transition: the
# guide axpy at td = 67, where the posterior kernel spreads the columns over gridDim.y.
# (case, net, T_train, n_steps, B, guide_weight)
GUIDE_SHORT_CASES = [
    ("guide_door_w0p5_T4", "door_j", 1000, 4, 2, 0.5),
]
VALUE_HIDDEN = 16


TIME_DIMS = {"tiny_td64": 64}        # nets whose time embedding is wider than `dim`
KERNEL_SIZES = {"tiny_k3": 3, "tiny_k7": 7}      # nets whose Conv1dBlocks are not 5 taps wide


# Training objective, forward only (diffusion.py:253-290; losses/__init__.py:37-186).
# (case, net, T, B, loss_type, predict_epsilon, weighted)
TRAIN_CASES = [
    ("train_tiny_l2", "tiny", 20, 6, "l2", True, False),
    ("train_tiny_l1_w", "tiny", 20, 6, "l1", True, True),
    ("train_tiny_x0_l2", "tiny", 20, 5, "l2", False, False),
    ("train_pointmaze_l2", "pointmaze", 100, 9, "l2", True, False),
]


# Backward pass (utils/training.py:152-156: loss.backward() through the denoiser): every parameter
# gradient + d loss / d x_t of the reference, draws injected as above.  Large tensors are pinned by a
# strided sample + their sum and sum of squares (the GPU test also compares EVERY element with the
# oracle's autograd, which test_oracle_golden.py holds to these samples).
# (case, net, T, B, loss_type, predict_epsilon, weighted)
GRAD_CASES = [
    ("grads_tiny", "tiny", 20, 6, "l2", True, False),
    ("grads_tiny4", "tiny4", 20, 5, "l1", True, True),
    ("grads_pointmaze_B9", "pointmaze", 100, 9, "l2", True, False),
    ("grads_tiny_k3", "tiny_k3", 20, 5, "l2", True, False),
    ("grads_tiny_k7", "tiny_k7", 20, 4, "l2", True, True),
    ("grads_tiny_d48", "tiny_d48", 20, 5, "l2", True, False),     # zero-padded GroupNorm groups (utils/padding.py)
    ("grads_tiny_d24", "tiny_d24", 20, 4, "l1", True, True),
]
GRAD_SAMPLE = 2048


def grad_sample_index(numel: int) -> np.ndarray:
    if numel <= 2 * GRAD_SAMPLE:
        return np.arange(numel)
    return np.arange(GRAD_SAMPLE) * (numel // GRAD_SAMPLE)


def train_inputs(case: str, net: str, T: int, B: int, weighted: bool):
    """(x_start, per-row timesteps, noise, weights or None): what loss() draws, made portable."""
    _, _, td, _, _ = net_dims(net)
    x0 = np.clip(synth.normal_like(24, case + ".x0", (B, H, td)) * 0.5, -1, 1).astype(np.float32)
    u = synth.uniform(24, case + ".t", (B,), 1.0)
    t = np.minimum(((u + 1.0) * 0.5 * T).astype(np.int64), T - 1)
    t[0], t[-1] = 0, T - 1                                   # both ends of the schedule
    noise = synth.normal_like(24, case + ".noise", (B, H, td))
    w = (1.0 + synth.uniform(24, case + ".w", (1, H, td), 0.5)).astype(np.float32) if weighted else None
    return x0, t, noise, w


def net_time_dim(net: str):
    return TIME_DIMS.get(net)


def net_kernel_size(net: str) -> int:
    return KERNEL_SIZES.get(net, 5)


def net_dims(net: str):
    od, ad, dim, mults, T, seed, jitter = NETS[net]
    return od, ad, od + ad, dim, mults


def net_weights(net: str) -> "OrderedDict[str, np.ndarray]":
    od, ad, dim, mults, T, seed, jitter = NETS[net]
    return synth.synth_unet_state(od + ad, dim, mults, seed=seed, affine_jitter=jitter,
                                  time_dim=net_time_dim(net), kernel_size=net_kernel_size(net))


def forward_input(case: str, net: str, B: int) -> np.ndarray:
    _, _, td, _, _ = net_dims(net)
    return synth.normal_like(21, case + ".x", (B, H, td))


def loop_noise(case: str, net: str, n_steps: int, B: int) -> np.ndarray:
    """Noise stack [x_T, z_{T-1}, ..., z_0] in the reference's randn call order."""
    _, _, td, _, _ = net_dims(net)
    return synth.normal_like(22, case + ".noise", (n_steps + 1, B, H, td))


def loop_condition(case: str, net: str) -> np.ndarray:
    """(1, td) condition as get_action builds it: normalised obs, action part zero
    (guides/policies.py:210-214)."""
    od, ad, td, _, _ = net_dims(net)
    c = np.zeros((1, td), np.float32)
    c[0, :od] = synth.uniform(23, case + ".cond", (od,), 0.9)
    return c


def value_net_weights(od: int, seed: int = 31) -> Dict[str, np.ndarray]:
    """Tiny fixed value MLP: V(obs) = W2 tanh(W1 obs + b1) + b2, per horizon step."""
    return {
        "w1": synth.uniform(seed, "value.w1", (VALUE_HIDDEN, od), 0.7),
        "b1": synth.uniform(seed, "value.b1", (VALUE_HIDDEN,), 0.2),
        "w2": synth.uniform(seed, "value.w2", (1, VALUE_HIDDEN), 0.7),
        "b2": synth.uniform(seed, "value.b2", (1,), 0.2),
    }


# ---------------------------------------------------------------------------- projection
# (case, dt, horizon)
PROJ_MATRIX_CASES = [("P_dt0p1_H8", 0.1, 8), ("P_dt0p01_H8", 0.01, 8), ("P_dt0p1_H32", 0.1, 32)]
PROJ_SCHEDULES = ["constant", "linear", "quadratic", "noise_schedule"]
PROJ_T = [0, 10, 99]
PROJ_STRENGTH = 0.8
PROJ_B = 5


class NormalizerStub:
    """The 6-attribute duck type the planner consumes (guides/policies.py:159,190,209,
    334-337); the reference's real normalizer package is absent from the snapshot."""

    def __init__(self, od: int, ad: int, seed: int = 41):
        self.obs_mean = synth.normal_like(seed, "norm.obs_mean", (od,))
        self.obs_std = (1.0 + synth.uniform(seed, "norm.obs_std", (od,), 0.5)).astype(np.float32)
        self.action_mean = synth.normal_like(seed, "norm.act_mean", (ad,))
        self.action_std = (1.0 + synth.uniform(seed, "norm.act_std", (ad,), 0.5)).astype(np.float32)

    def normalize_observations(self, obs):
        return ((np.asarray(obs, np.float32) - self.obs_mean) / self.obs_std).astype(np.float32)

    def unnormalize_actions(self, a):
        return (np.asarray(a, np.float32) * self.action_std + self.action_mean).astype(np.float32)


def projection_input(case: str) -> np.ndarray:
    return synth.normal_like(42, case + ".x", (PROJ_B, H, 6))


# --------------------------------------------------------------------------- planner glue
ACTION_HORIZONS = [1, 8, 32]
N_GET_ACTION_CALLS = 12


def glue_observations() -> np.ndarray:
    return synth.normal_like(51, "glue.obs", (N_GET_ACTION_CALLS, 4))


# --------------------------------------------------------------------------- system identification
def sysid_transitions(case: str = "sysid", n_obs: int = 6, n: int = 4, m: int = 2, N: int = 300):
    """(states, actions, next_states) of a noisy double integrator inside 6-column observations
    (columns 4..5 play PointMaze's goal: unrelated to the dynamics) — data_driven.py:75-134."""
    dt = 0.1
    A = np.array([[1, 0, dt, 0], [0, 1, 0, dt], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    B = np.array([[0.5 * dt * dt, 0], [0, 0.5 * dt * dt], [dt, 0], [0, dt]], dtype=np.float64)
    S = synth.normal_like(71, case + ".s", (N, n_obs)).astype(np.float64)
    U = synth.normal_like(71, case + ".u", (N, m)).astype(np.float64)
    S1 = synth.normal_like(71, case + ".s1", (N, n_obs)).astype(np.float64)
    S1[:, :n] = S[:, :n] @ A.T + U @ B.T + 1e-3 * synth.normal_like(71, case + ".e", (N, n)).astype(np.float64)
    return S, U, S1
