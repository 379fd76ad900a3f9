"""ctypes binding of ``libdad_hip.so`` (C ABI: ``include/dad.h``) and the thin engine object
the API mirror classes drive.

There is deliberately NO fallback: if the library is missing or a tensor is not on a ROCm
device the calls raise.  torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Mapping, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DAD_LIB selects an alternative build of the same ABI (timing-only ablation builds).
LIB_PATH = os.environ.get("DAD_LIB") or os.path.join(_HERE, "libdad_hip.so")

DAD_MAX_LEVELS = 8


class DadCfg(C.Structure):
    _fields_ = [
        ("transition_dim", C.c_int32), ("dim", C.c_int32), ("time_dim", C.c_int32),
        ("n_levels", C.c_int32), ("channels", C.c_int32 * DAD_MAX_LEVELS),
        ("kernel_size", C.c_int32), ("horizon", C.c_int32), ("n_timesteps", C.c_int32),
        ("predict_epsilon", C.c_int32), ("clip_denoised", C.c_int32),
    ]


class DadStepArgs(C.Structure):
    _fields_ = [
        ("noise", C.c_void_p), ("seed", C.c_uint64), ("row_offset", C.c_uint64),
        ("draw", C.c_uint64), ("cond0", C.c_void_p), ("cond_per_row", C.c_int32),
        ("guide_grad", C.c_void_p), ("guide_weight", C.c_float),
        ("mean_out", C.c_void_p), ("eps_out", C.c_void_p),
    ]


class DadProjectArgs(C.Structure):
    _fields_ = [
        ("P", C.c_void_p), ("obs_mean", C.c_void_p), ("obs_std", C.c_void_p),
        ("act_mean", C.c_void_p), ("act_std", C.c_void_p),
        ("state_dim", C.c_int32), ("observation_dim", C.c_int32), ("action_dim", C.c_int32),
        ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t),
    ]


# name -> (restype, argtypes); also the list of symbols include/dad.h declares.
ABI = {
    "dad_last_error": (C.c_char_p, []),
    "dad_version": (C.c_char_p, []),
    "dad_model_create": (C.c_int, [C.POINTER(DadCfg), C.POINTER(C.c_void_p)]),
    "dad_model_destroy": (None, [C.c_void_p]),
    "dad_model_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p,
                                        C.POINTER(C.c_int64), C.c_int32]),
    "dad_model_load_schedule": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5),
    "dad_model_load_time_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "dad_model_set_precision": (C.c_int, [C.c_void_p, C.c_int32]),
    "dad_model_set_group_channels": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]),
    "dad_model_set_horizon": (C.c_int, [C.c_void_p, C.c_int32]),
    "dad_model_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dad_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_size_t)]),
    "dad_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "dad_unet_forward_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "dad_projection_violation": (C.c_int, [C.POINTER(DadProjectArgs), C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_void_p]),
    "dad_denoise_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                   C.POINTER(DadStepArgs), C.c_int32, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "dad_sample_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_uint64, C.c_uint64, C.c_void_p, C.c_int32,
                                  C.POINTER(DadProjectArgs), C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    "dad_project": (C.c_int, [C.POINTER(DadProjectArgs), C.c_float, C.c_void_p, C.c_int32,
                              C.c_int32, C.c_void_p]),
    "dad_fill_normal": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64,
                                  C.c_uint64, C.c_void_p]),
    "dad_debug_set_tile": (C.c_int, [C.c_void_p, C.c_int32]),
    "dad_debug_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "dad_debug_read_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                       C.POINTER(C.c_int32)]),
    "dad_debug_mish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dad_debug_kernel_table_consistent": (C.c_int, []),
    "dad_debug_small_batch_plan": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int32)]),
    "dad_model_set_training": (C.c_int, [C.c_void_p, C.c_int32]),
    "dad_model_refresh_weights": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p),
                                            C.c_void_p]),
    "dad_train_grad_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "dad_train_grad_info": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64)]),
    "dad_train_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_size_t)]),
    "dad_unet_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dad_unet_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_void_p), C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dad_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "dad_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_double)]),
}

_lib = None


def load_library() -> C.CDLL:
    """Load libdad_hip.so and type every entry point.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with dynamics_aware_diffusion_amd/csrc/build.sh "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in ABI.items():
        if os.environ.get("DAD_LIB") and not hasattr(lib, name):
            continue                       # an older timing-only build (A/B of two libraries on one box)
        fn = getattr(lib, name)            # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


class DadError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libdad_hip error {code}: {message}")
        self.code = code


def _check(lib, rc: int) -> None:
    if rc != 0:
        # DadError is a RuntimeError: DAD_E_RANGE (timestep outside the schedule) therefore surfaces
        # as the same exception type the reference's gather raises (diffusion.py:28)
        raise DadError(rc, lib.dad_last_error().decode("utf-8", "replace"))


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a ROCm device (got {t.device}); the HIP engine "
                           "has no CPU path")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous float32 (got {t.dtype}, "
                           f"contiguous={t.is_contiguous()})")


PRECISIONS = {"fp32": 0, "f16x3": 1}      # DAD_PREC_* of include/dad.h
TABLES = {"sinusoid": 0, "time_mlp": 1, "blocks": 2}      # DAD_TABLE_* of include/dad.h


def sinusoid_table(n_timesteps: int, dim: int) -> torch.Tensor:
    """SinusoidalPosEmb for t = 0 .. n_timesteps-1 with the reference's own torch expression
    (/root/reference/m_diffuser/models/temporal_unet.py:27-31), on the host: (n_timesteps, dim)
    fp32, bit-identical to what the reference's module computes on this machine."""
    half = dim // 2
    scale = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half) * -scale)
    arg = torch.arange(n_timesteps)[:, None] * freqs[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1).contiguous()

SCHEDULE_KEYS = ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                 "posterior_mean_coef1", "posterior_mean_coef2",
                 "posterior_log_variance_clipped")


class HipEngine:
    """One ``dad_model`` on one device: packed weights, time tables, workspaces."""

    def __init__(self, *, transition_dim: int, dim: int, channels: Sequence[int], horizon: int,
                 n_timesteps: int, time_dim: Optional[int] = None, kernel_size: int = 5,
                 predict_epsilon: bool = True, clip_denoised: bool = True,
                 device: torch.device | str = "cuda", precision: str = "fp32", training: bool = False):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
        self.precision = precision
        self.lib = load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipEngine needs a ROCm device; there is no CPU path")
        # widths the tiles cannot hold (not a multiple of 32 with a power-of-two C / 8) run zero-padded:
        # utils/padding.py builds the padded tensors, the library is told the real widths
        from .utils import padding
        self.real_dim, self.real_channels = int(dim), [int(c) for c in channels]
        padded = [padding.padded_width(c) for c in self.real_channels]
        self.padded = padded != self.real_channels
        if self.padded:
            if self.real_channels[0] != self.real_dim:
                raise NotImplementedError("zero-padded widths need dim_mults[0] == 1")
            if transition_dim == self.real_dim:
                raise NotImplementedError(
                    f"transition_dim == dim == {dim} makes the first block's residual the trajectory itself; "
                    "with zero-padded GroupNorm groups that identity has no kernel")
        # horizons every level can halve but that are not a power of two (24, 48, 96, 100 ...) run zero-padded to
        # the next power of two (dad_model_set_horizon); the trajectory tensors keep their real shape
        horizon = int(horizon)
        levels = len(self.real_channels)
        # (... and the deepest level keeps at least four padded positions, the tiles' lower bound: horizon 16 on four
        # levels — 16 / 8 / 4 / 2 positions — runs in 32 / 16 / 8 / 4)
        self.padded_horizon = max(1 << max(horizon - 1, 0).bit_length(), 4 << (levels - 1))
        self.rows_padded = self.padded_horizon != horizon
        if self.rows_padded:
            if horizon % (1 << (levels - 1)) != 0:
                raise ValueError(f"horizon {horizon} cannot be halved {levels - 1} times (the reference's U-Net needs "
                                 f"H % 2^(levels-1) == 0, temporal_unet.py:35-54)")
            if transition_dim == self.real_dim:
                raise NotImplementedError("transition_dim == dim with a zero-padded horizon has no kernel")
        self.padded = self.padded or self.rows_padded
        cfg = DadCfg()
        cfg.transition_dim = transition_dim
        cfg.dim = padded[0] if padded != self.real_channels else dim
        cfg.time_dim = time_dim or dim
        cfg.n_levels = len(channels)
        for i, ch in enumerate(padded):
            cfg.channels[i] = int(ch)
        cfg.kernel_size = kernel_size
        cfg.horizon = self.padded_horizon
        cfg.n_timesteps = n_timesteps
        cfg.predict_epsilon = int(predict_epsilon)
        cfg.clip_denoised = int(clip_denoised)
        self.cfg = cfg
        self.horizon = horizon
        self.transition_dim = transition_dim
        self.n_timesteps = n_timesteps
        handle = C.c_void_p()
        _check(self.lib, self.lib.dad_model_create(C.byref(cfg), C.byref(handle)))
        self._h = handle
        _check(self.lib, self.lib.dad_model_set_precision(self._h, PRECISIONS[precision]))
        self.widths_padded = padded != self.real_channels
        if self.widths_padded:
            real = (C.c_int32 * len(self.real_channels))(*self.real_channels)
            _check(self.lib, self.lib.dad_model_set_group_channels(self._h, real, len(self.real_channels)))
        if self.rows_padded:
            _check(self.lib, self.lib.dad_model_set_horizon(self._h, horizon))
        self.training = bool(training)
        if self.training:
            _check(self.lib, self.lib.dad_model_set_training(self._h, 1))
        self._ws: Dict[int, torch.Tensor] = {}
        self._pinned: Dict[tuple, torch.Tensor] = {}
        self.ready = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self.lib.dad_model_destroy(h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------ weights
    def load(self, unet_state: Mapping[str, torch.Tensor],
             schedule: Mapping[str, torch.Tensor]) -> None:
        """Upload every denoiser tensor (reference state_dict keys without ``model.``) and
        the five schedule buffers, then build tables and the launch plan."""
        if self.widths_padded:
            from .utils import padding
            mults = [c // self.real_dim for c in self.real_channels]
            self._pad_plan, _, _ = padding.padding_plan(unet_state.keys(), self.transition_dim, self.real_dim, mults)
            unet_state, _, _ = padding.pad_unet_state(unet_state, self.transition_dim, self.real_dim, mults)
        keep = []
        for key, t in unet_state.items():
            h = t.detach().to("cpu", torch.float32).contiguous()
            keep.append(h)
            shape = (C.c_int64 * h.dim())(*h.shape)
            _check(self.lib, self.lib.dad_model_load_weight(
                self._h, key.encode(), h.data_ptr(), shape, h.dim()))
        bufs = []
        for k in SCHEDULE_KEYS:
            b = schedule[k].detach().to("cpu", torch.float32).contiguous()
            if b.numel() != self.n_timesteps:
                raise ValueError(f"schedule buffer {k} has {b.numel()} entries, expected "
                                 f"{self.n_timesteps}")
            bufs.append(b)
        _check(self.lib, self.lib.dad_model_load_schedule(self._h, *[b.data_ptr() for b in bufs]))
        emb = sinusoid_table(self.n_timesteps, self.real_dim)
        if self.widths_padded:             # the real columns in front, as time_mlp.1.weight is padded
            wide = torch.zeros(self.n_timesteps, int(self.cfg.dim))
            wide[:, :self.real_dim] = emb
            emb = wide.contiguous()
        _check(self.lib, self.lib.dad_model_load_time_embedding(
            self._h, emb.data_ptr(), self.n_timesteps, int(self.cfg.dim)))
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_model_finalize(self._h, self._stream()))
        self.ready = True

    # ------------------------------------------------------------------ helpers
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def workspace(self, batch: int) -> torch.Tensor:
        """Device scratch for `batch` rows (activations + split-K slabs); grown on demand."""
        n = C.c_size_t()
        _check(self.lib, self.lib.dad_workspace_bytes(self._h, batch, C.byref(n)))
        ws = self._ws.get(batch)
        if ws is None or ws.numel() * 4 < n.value:
            ws = torch.empty(max(n.value, 4) // 4 + 4, dtype=torch.float32, device=self.device)
            self._ws[batch] = ws
        return ws

    def persistent(self, tag: str, shape) -> torch.Tensor:
        """A device tensor with a stable address per (tag, shape): hipGraph replays freeze the
        pointers they were captured with."""
        key = (tag, tuple(shape))
        t = self._pinned.get(key)
        if t is None:
            t = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
            self._pinned[key] = t
        return t

    def _traj(self, x: torch.Tensor, name: str = "x") -> int:
        _require_device(x, name)
        if x.dim() != 3 or x.shape[1] != self.horizon or x.shape[2] != self.transition_dim:
            raise RuntimeError(f"{name} must be (B, {self.horizon}, {self.transition_dim}), "
                               f"got {tuple(x.shape)}")
        return int(x.shape[0])

    # ------------------------------------------------------------------ compute
    def unet_forward(self, x: torch.Tensor, t: int) -> torch.Tensor:
        B = self._traj(x)
        out = torch.empty_like(x)
        ws = self.workspace(B)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_unet_forward(
                self._h, x.data_ptr(), int(t), out.data_ptr(), B, ws.data_ptr(),
                ws.numel() * 4, self._stream()))
        return out

    def unet_forward_rows(self, x: torch.Tensor, t_rows: torch.Tensor) -> torch.Tensor:
        """eps_theta(x, t) with one timestep per row (the training objective's call); ``t_rows``
        is range-checked by the caller (``TemporalUnet.forward``)."""
        B = self._traj(x)
        t32 = t_rows.to(device=x.device, dtype=torch.int32).contiguous()
        if t32.numel() != B:
            raise RuntimeError(f"time must have one entry per row: got {t32.numel()} for batch {B}")
        out = torch.empty_like(x)
        ws = self.workspace(B)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_unet_forward_rows(
                self._h, x.data_ptr(), t32.data_ptr(), out.data_ptr(), B, ws.data_ptr(),
                ws.numel() * 4, self._stream()))
        return out

    def denoise_step(self, x: torch.Tensor, t: int, *, noise: Optional[torch.Tensor] = None,
                     seed: int = 0, row_offset: int = 0, draw: int = 0,
                     cond0: Optional[torch.Tensor] = None,
                     guide_grad: Optional[torch.Tensor] = None, guide_weight: float = 0.0,
                     mean_out: Optional[torch.Tensor] = None,
                     eps_out: Optional[torch.Tensor] = None, update_x: bool = True) -> None:
        """In-place reverse step on ``x`` (see dad_denoise_step in include/dad.h)."""
        B = self._traj(x)
        a = DadStepArgs()
        for name, ten in (("noise", noise), ("guide_grad", guide_grad), ("mean_out", mean_out),
                          ("eps_out", eps_out)):
            if ten is not None:
                if self._traj(ten, name) != B:
                    raise RuntimeError(f"{name} batch mismatch")
                setattr(a, name, ten.data_ptr())
        a.seed, a.row_offset, a.draw = int(seed), int(row_offset), int(draw)
        if cond0 is not None:
            _require_device(cond0, "cond0")
            rows = cond0.reshape(-1, self.transition_dim).shape[0]
            if rows not in (1, B):
                raise RuntimeError(f"cond0 must be (1, td) or (B, td), got {tuple(cond0.shape)}")
            a.cond0 = cond0.data_ptr()
            a.cond_per_row = int(rows == B and B > 1)
        a.guide_weight = float(guide_weight)
        ws = self.workspace(B)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_denoise_step(
                self._h, x.data_ptr(), int(t), B, C.byref(a), int(not update_x), ws.data_ptr(),
                ws.numel() * 4, self._stream()))

    def sample_loop(self, x: torch.Tensor, n_steps: int, *,
                    noise_stack: Optional[torch.Tensor] = None, seed: int = 0,
                    row_offset: int = 0, cond0: Optional[torch.Tensor] = None,
                    projection: Optional["ProjectionState"] = None,
                    proj_alphas: Optional[Sequence[float]] = None,
                    use_graph: bool = False) -> None:
        B = self._traj(x)
        cond_ptr, per_row = None, 0
        if cond0 is not None:
            _require_device(cond0, "cond0")
            rows = cond0.reshape(-1, self.transition_dim).shape[0]
            if rows not in (1, B):
                raise RuntimeError(f"cond0 must be (1, td) or (B, td), got {tuple(cond0.shape)}")
            cond_ptr, per_row = cond0.data_ptr(), int(rows == B and B > 1)
        if noise_stack is not None:
            _require_device(noise_stack, "noise_stack")
            if tuple(noise_stack.shape) != (n_steps, B, self.horizon, self.transition_dim):
                raise RuntimeError("noise_stack must be (n_steps, B, H, td)")
        pa, alphas = None, None
        if projection is not None:
            projection._check_x(x)
            pa = C.byref(projection.args)
            alphas = (C.c_float * self.n_timesteps)(*[float(a) for a in proj_alphas])
        ws = self.workspace(B)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_sample_loop(
                self._h, x.data_ptr(), int(n_steps), B, _ptr(noise_stack), int(seed),
                int(row_offset), cond_ptr, per_row, pa, alphas, int(use_graph), ws.data_ptr(),
                ws.numel() * 4, self._stream()))

    def fill_normal(self, x: torch.Tensor, seed: int, row_offset: int = 0, draw: int = 0) -> None:
        _require_device(x, "x")
        B = int(x.shape[0])
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_fill_normal(
                x.data_ptr(), B, x.numel() // B, int(seed), int(row_offset), int(draw),
                self._stream()))

    # ------------------------------------------------------------------ training
    def refresh(self, unet_state: Mapping[str, torch.Tensor]) -> None:
        """Re-derive the engine's packed copies from parameter tensors that already live on this device
        (fp32, contiguous): no host round trip (dad_model_refresh_weights)."""
        keys, ptrs, keep = [], [], []
        if self.widths_padded:             # the padded tensors are rebuilt on the device: one scatter for all of them
            with torch.no_grad():
                names = list(unet_state.keys())
                fp = self.flat_padding(names, [tuple(unet_state[k].shape) for k in names])
                wide = fp.pad_flat([unet_state[k].detach().to(self.device) for k in names])
            base, n = wide.data_ptr(), len(names)        # (FlatPadding's packed offsets are 16-byte aligned)
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.dad_model_refresh_weights(
                    self._h, n, (C.c_char_p * n)(*[k.encode() for k in names]),
                    (C.c_void_p * n)(*[base + 4 * o for o in fp.offsets]), self._stream()))
            self._refresh_keep = [wide]
            return
        for key, t in unet_state.items():
            d = t.detach()
            if d.device != self.device or d.dtype != torch.float32 or not d.is_contiguous():
                d = d.to(self.device, torch.float32).contiguous()
            keep.append(d)
            keys.append(key.encode())
            ptrs.append(d.data_ptr())
        n = len(keys)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_model_refresh_weights(
                self._h, n, (C.c_char_p * n)(*keys), (C.c_void_p * n)(*ptrs), self._stream()))
        self._refresh_keep = keep          # (stream-ordered: the copies run before anything enqueued later)

    def flat_padding(self, keys, shapes, offsets=None, total=None):
        """utils/padding.FlatPadding of the named tensors (real shapes) for this net's zero-padded widths."""
        from .utils import padding
        cache = self.__dict__.setdefault("_flat_paddings", {})
        sig = (tuple(keys), None if offsets is None else tuple(offsets))
        if sig not in cache:
            cache[sig] = padding.FlatPadding(keys, shapes, self._pad_plan, offsets, total)
        return cache[sig]

    def time_projection_index(self) -> torch.Tensor:
        """Column of every real time-projection entry in the padded rows the training forward reads: the blocks'
        projections side by side in launch order, each at its level's padded width."""
        from .utils import padding
        cached = getattr(self, "_temb_index", None)
        if cached is None:
            n = len(self.real_channels)
            widths = [c for c in self.real_channels for _ in (0, 1)] + [self.real_channels[-1]] * 2
            for j in range(n - 1):
                widths += [self.real_channels[n - 2 - j]] * 2
            index, off = [], 0
            for c in widths:
                index.append(padding.channel_index(c) + off)
                off += padding.padded_width(c)
            cached = self._temb_index = (torch.cat(index).to(self.device), off)
        return cached

    def grad_layout(self):
        """[(reference key without 'model.', offset in floats, numel)] of the flat gradient buffer and
        its total length."""
        cached = getattr(self, "_grad_layout", None)      # (a property of the architecture: asked once)
        if cached is not None:
            return cached
        n, total = C.c_int32(), C.c_int64()
        _check(self.lib, self.lib.dad_train_grad_count(self._h, C.byref(n), C.byref(total)))
        out = []
        for i in range(n.value):
            key, off, numel = C.c_char_p(), C.c_int64(), C.c_int64()
            _check(self.lib, self.lib.dad_train_grad_info(self._h, i, C.byref(key), C.byref(off), C.byref(numel)))
            out.append((key.value.decode(), off.value, numel.value))
        self._grad_layout = (out, total.value)
        return self._grad_layout

    def train_forward(self, x: torch.Tensor, temb_rows: torch.Tensor):
        """eps_theta(x) with per-row time projections ``temb_rows`` (B, temb_width), keeping every
        activation: returns (out, saved) — ``saved`` goes to :meth:`train_backward`."""
        B = self._traj(x)
        _require_device(temb_rows, "temb_rows")
        if temb_rows.dim() != 2 or temb_rows.shape[0] != B:
            raise RuntimeError(f"temb_rows must be (B, temb_width), got {tuple(temb_rows.shape)}")
        sv, sc = C.c_size_t(), C.c_size_t()
        _check(self.lib, self.lib.dad_train_workspace_bytes(self._h, B, C.byref(sv), C.byref(sc)))
        saved = torch.empty(max(sv.value, 4) // 4 + 4, dtype=torch.float32, device=self.device)
        rows = torch.arange(B, dtype=torch.int32, device=self.device)
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_unet_forward_train(
                self._h, x.data_ptr(), rows.data_ptr(), temb_rows.data_ptr(), out.data_ptr(), B,
                saved.data_ptr(), saved.numel() * 4, self._stream()))
        return out, saved

    def train_backward(self, x: torch.Tensor, d_out: torch.Tensor, saved: torch.Tensor, temb_width: int, shapes):
        """(d_x, d_temb_rows, [one gradient tensor per entry of grad_layout(), shaped like ``shapes``]) for
        one batch (see dad_unet_backward).  Separate tensors: autograd adopts them as ``.grad`` without copying."""
        B = self._traj(x)
        if self._traj(d_out, "d_out") != B:
            raise RuntimeError("d_out batch mismatch")
        sv, sc = C.c_size_t(), C.c_size_t()
        _check(self.lib, self.lib.dad_train_workspace_bytes(self._h, B, C.byref(sv), C.byref(sc)))
        scratch = torch.empty(max(sc.value, 4) // 4 + 4, dtype=torch.float32, device=self.device)
        layout, total = self.grad_layout()
        if len(shapes) != len(layout):
            raise RuntimeError(f"{len(shapes)} parameter shapes for {len(layout)} gradient tensors")
        real = None
        if self.widths_padded:
            # zero-padded widths: the library fills padded gradients; `shapes` are the REAL parameters' — one gather
            # over the flat buffer picks the real entries (utils/padding.FlatPadding), the padding's are dropped
            real = self.flat_padding([k for k, _, _ in layout], shapes, [o for _, o, _ in layout], total)
            shapes = real.padded_shapes
            for (key, _, numel), n in zip(layout, real.sizes):
                if n != numel:
                    raise RuntimeError(f"{key}: {n} padded elements, the library expects {numel}")
        # ONE allocation per step, split into per-parameter views at the library's own (16-byte aligned) offsets:
        # autograd adopts a contiguous view as `.grad` as it does a tensor of its own (no copy), and the host side of
        # a step loses ~150 allocator calls
        flat = torch.empty(total, dtype=torch.float32, device=self.device)
        base = flat.data_ptr()
        spans = [layout[i + 1][1] - layout[i][1] for i in range(len(layout) - 1)] + [total - layout[-1][1]]
        pieces = flat.split_with_sizes(spans)              # one call: a view per slot (slots are padded to 4 floats)
        grads = [(pc if pc.numel() == numel else pc[:numel]).view(tuple(shape))
                 for pc, (_, _, numel), shape in zip(pieces, layout, shapes)]
        ptrs = (C.c_void_p * len(grads))(*[base + 4 * offset for _, offset, _ in layout])
        d_x = torch.empty_like(x)
        d_temb = torch.empty(B, temb_width, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_unet_backward(
                self._h, x.data_ptr(), d_out.data_ptr(), d_x.data_ptr(), d_temb.data_ptr(), ptrs, len(grads), B,
                saved.data_ptr(), saved.numel() * 4, scratch.data_ptr(), scratch.numel() * 4, self._stream()))
        if real is not None:
            grads = real.gather(flat)
        return d_x, d_temb, grads

    # ------------------------------------------------------------------ test / tuning hooks
    def debug_set_tile(self, cfg: int) -> None:
        """Force a conv tile (0..9), -1 = heuristic, 100+cfg / 99 = same without grid split-K."""
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_debug_set_tile(self._h, int(cfg)))

    def debug_set_option(self, name: str, value: int) -> None:
        # (the entry point synchronises the CURRENT device before dropping captured loops)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_debug_set_option(self._h, name.encode(), int(value)))

    def read_table(self, which: str, t: int) -> torch.Tensor:
        """Row t of a per-timestep table (see DAD_TABLE_* in include/dad.h), as a CPU tensor."""
        out = torch.empty(1 << 16, dtype=torch.float32)
        width = C.c_int32()
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_debug_read_table(self._h, TABLES[which], int(t),
                                                           out.data_ptr(), out.numel(), C.byref(width)))
        return out[:width.value].clone()

    def small_batch_plan(self, batch: int):
        """(conv launches through the consumer-combine kernels, how many of them are the
        streamed-weight form for wide layers); (0, 0) when the batch takes the batch-256 kernels."""
        n, w = C.c_int32(), C.c_int32()
        _check(self.lib, self.lib.dad_debug_small_batch_plan(self._h, int(batch), C.byref(n), C.byref(w)))
        return n.value, w.value

    def mish(self, x: torch.Tensor) -> torch.Tensor:
        """The conv epilogue's Mish applied to a device tensor (test hook)."""
        _require_device(x, "x")
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.dad_debug_mish(x.data_ptr(), out.data_ptr(), x.numel(),
                                                     self._stream()))
        return out

    # ------------------------------------------------------------------ profiling
    def profile_enable(self, on: bool) -> None:
        _check(self.lib, self.lib.dad_profile_enable(self._h, int(on)))

    def profile_read(self):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        _check(self.lib, self.lib.dad_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value


class ProjectionState:
    """Device-resident projector + normaliser statistics for dad_project."""

    def __init__(self, P: torch.Tensor, obs_mean, obs_std, act_mean, act_std, state_dim: int,
                 observation_dim: int, action_dim: int, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("projection kernel needs a ROCm device; there is no CPU path")

        def put(v):
            return torch.as_tensor(v, dtype=torch.float32).contiguous().to(dev)

        self.P = put(P)
        self.obs_mean, self.obs_std = put(obs_mean), put(obs_std)
        self.act_mean, self.act_std = put(act_mean), put(act_std)
        a = DadProjectArgs()
        a.P = self.P.data_ptr()
        a.obs_mean, a.obs_std = self.obs_mean.data_ptr(), self.obs_std.data_ptr()
        a.act_mean, a.act_std = self.act_mean.data_ptr(), self.act_std.data_ptr()
        a.state_dim, a.observation_dim, a.action_dim = state_dim, observation_dim, action_dim
        self.args = a
        self.device = dev
        self._scratch = None
        self.gemm = True          # batches of 32+ trajectories: v @ P as an MFMA GEMM (needs the scratch copy)
        self.D = int(self.P.shape[0])
        if self.P.dim() != 2 or self.P.shape[1] != self.D:
            raise ValueError(f"projection matrix must be square, got {tuple(self.P.shape)}")
        if self.obs_mean.numel() != observation_dim or self.obs_std.numel() != observation_dim \
                or self.act_mean.numel() != action_dim or self.act_std.numel() != action_dim:
            raise ValueError("normaliser statistics do not match observation_dim / action_dim")

    def _check_x(self, x: torch.Tensor) -> None:
        """The kernel derives D = (H+1) n + H m from x's horizon and indexes P[D, D] with it: a batch
        with another horizon or transition width would read P out of bounds (the reference raises a
        matmul shape error, guides/policies.py:451, losses/__init__.py:181)."""
        _require_device(x, "x")
        a = self.args
        if x.dim() != 3 or x.shape[2] != a.observation_dim + a.action_dim:
            raise RuntimeError(f"x must be (B, H, {a.observation_dim + a.action_dim}), got {tuple(x.shape)}")
        D = (int(x.shape[1]) + 1) * a.state_dim + int(x.shape[1]) * a.action_dim
        if D != self.D:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({x.shape[0]}x{D} and "
                               f"{self.D}x{self.D}): the projector was built for another horizon")
        if x.device != self.P.device:
            raise RuntimeError(f"x is on {x.device}, the projector on {self.P.device}")
        # scratch copy of the batch for the GEMM form (grown on demand; stable address while the batch
        # size repeats, so captured loops keep their pointer)
        if self.gemm and (self._scratch is None or self._scratch.numel() < x.numel()):
            self._scratch = torch.empty(x.numel(), dtype=torch.float32, device=self.device)
            self.args.scratch = self._scratch.data_ptr()
            self.args.scratch_bytes = self._scratch.numel() * 4

    def violation(self, x: torch.Tensor) -> torch.Tensor:
        """Per-row squared distance from the dynamics-consistent subspace, physical units
        (ProjectionLoss.compute, losses/__init__.py:161-186, before its mean)."""
        self._check_x(x)
        out = torch.empty(int(x.shape[0]), dtype=torch.float32, device=x.device)
        lib = load_library()
        with torch.cuda.device(self.device):
            _check(lib, lib.dad_projection_violation(C.byref(self.args), x.data_ptr(), out.data_ptr(),
                                                     int(x.shape[0]), int(x.shape[1]),
                                                     torch.cuda.current_stream(self.device).cuda_stream))
        return out

    def apply(self, x: torch.Tensor, alpha: float) -> None:
        self._check_x(x)
        lib = load_library()
        with torch.cuda.device(self.device):
            _check(lib, lib.dad_project(C.byref(self.args), float(alpha), x.data_ptr(),
                                        int(x.shape[0]), int(x.shape[1]),
                                        torch.cuda.current_stream(self.device).cuda_stream))
