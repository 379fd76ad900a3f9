"""``TemporalUnet`` — API mirror of ``m_diffuser.models.temporal_unet.TemporalUnet``
(/root/reference/m_diffuser/models/temporal_unet.py:125-241) whose forward pass runs on the
hand-written HIP engine (``csrc/``) instead of ``torch.nn`` layers.

The module owns ordinary ``nn.Parameter`` tensors under exactly the reference's
``state_dict`` keys, so ``load_state_dict`` of a reference
checkpoint works unchanged; the engine keeps its own packed copy, refreshed whenever the
parameters change.  There is no CPU forward: calling it with CPU tensors raises.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from .._engine import HipEngine
from ..utils.synth import unet_param_shapes


class _Node(nn.Module):
    """Empty container; only exists so parameter paths spell the reference's keys."""


def _attach(root: nn.Module, dotted: str, value: nn.Parameter) -> None:
    parts = dotted.split(".")
    node = root
    for name in parts[:-1]:
        child = node._modules.get(name)
        if child is None:
            child = _Node()
            node.add_module(name, child)
        node = child
    node.register_parameter(parts[-1], value)


def _default_init(key: str, shape: Tuple[int, ...], fan_in: Dict[str, int]) -> torch.Tensor:
    """PyTorch's default layer initialisation (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
    for conv/linear weights and biases; GroupNorm affine = 1 / 0)."""
    if ".block.1." in key:
        return torch.ones(shape) if key.endswith("weight") else torch.zeros(shape)
    stem = key.rsplit(".", 1)[0]
    if key.endswith("weight"):
        if len(shape) == 3:
            fan = shape[1] * shape[2]
            if key.endswith(".2.conv.weight") and shape[2] == 4:     # ConvTranspose1d (in,out,k)
                fan = shape[1] * shape[2]
        else:
            fan = shape[1]
        fan_in[stem] = fan
    bound = 1.0 / math.sqrt(fan_in[stem])
    return torch.empty(shape).uniform_(-bound, bound)


class TemporalUnet(nn.Module):
    """1-D conv U-Net denoiser eps_theta(x_t, t) on the MI355X HIP engine.

    Same constructor as the reference (temporal_unet.py:135-140).  ``forward(x, time)`` takes
    ``x`` (batch, horizon, transition_dim) and ``time`` (batch,) and returns a tensor like
    ``x``.  Every sampling call site passes one shared timestep (diffusion.py:248;
    guides/policies.py:146) — the fast path; rows with different timesteps, as the training
    objective draws them (diffusion.py:265), take the per-row time-embedding lookup.

    Autograd: like the reference's module, a call made with gradients enabled and a parameter (or
    ``x``) that requires grad returns a tensor with a graph — ``loss.backward()`` then fills ``.grad``
    of every parameter (training step of utils/training.py:144-156).  That call runs the training
    forward of the engine (every activation kept) and the explicit backward pass of
    ``dad_unet_backward``; the small time MLPs are evaluated with torch ops so that autograd carries
    them.  Under ``torch.no_grad()`` (every sampling entry point) nothing of this is touched.
    """

    default_precision = "fp32"      # what new instances start with (see ``precision`` below)
    default_small_batch_kernels = True

    def __init__(self, transition_dim: int, dim: int = 128,
                 dim_mults: Sequence[int] = (1, 2, 4, 8), kernel_size: int = 5,
                 time_dim: Optional[int] = None):
        super().__init__()
        self.transition_dim = int(transition_dim)
        self.dim = int(dim)
        self.dim_mults = tuple(int(m) for m in dim_mults)
        self.kernel_size = int(kernel_size)
        self.time_dim = int(time_dim or dim)
        self.channels = [self.dim * m for m in self.dim_mults]
        fan: Dict[str, int] = {}
        for key, shape in unet_param_shapes(self.transition_dim, self.dim, self.dim_mults,
                                            self.kernel_size, self.time_dim).items():
            _attach(self, key, nn.Parameter(_default_init(key, shape, fan)))
        # Conv arithmetic of the engine (include/dad.h DAD_PREC_*): "fp32" = exact fp32 MFMA,
        # "f16x3" = split-f16 operands, three f16 MFMAs per product, fp32 accumulation; both meet
        # the same fp32 parity gates.  Changing it re-packs the weights on the next call.
        self.precision = type(self).default_precision
        # Batches of up to 16 plans (batch * horizon <= 512 rows) run the consumer-combine kernels
        # (csrc/conv_cc.hpp: convs emit partial sums, consumers finish them) — the get_action path.
        # False keeps every batch on the batch-256 kernels with grid-level split-K.
        self.small_batch_kernels = type(self).default_small_batch_kernels
        # engine state (not part of state_dict)
        self._named: Optional[Dict[str, nn.Parameter]] = None
        self._engine: Optional[HipEngine] = None
        self._engine_sig = None
        self._engine_params = None
        self._schedule: Optional[Dict[str, torch.Tensor]] = None
        self._diffusion_opts = {"n_timesteps": 1000, "predict_epsilon": True,
                                "clip_denoised": True}

    # ------------------------------------------------------------------ engine plumbing
    def _params(self) -> Dict[str, nn.Parameter]:
        """name -> Parameter, walked once: the module tree is fixed after __init__ (``.to()`` / ``load_state_dict``
        keep the Parameter objects), and every call of the model needs the list — 0.27 ms per walk for 160 tensors."""
        if self._named is None:
            self._named = dict(self.named_parameters())
        return self._named

    def __getstate__(self):
        """``copy.deepcopy(model)`` / pickling (the reference's trainer keeps its EMA weights in a deep copy,
        utils/training.py:77): the copy gets the parameters and options, not the engine — a ``dad_model`` handle
        belongs to one object; the copy builds its own on its first call."""
        state = dict(self.__dict__)
        state.update(_engine=None, _engine_sig=None, _engine_params=None, _named=None)
        return state

    def _apply(self, fn, *args, **kwargs):
        self._named = None                 # (a conversion may replace Parameter objects)
        return super()._apply(fn, *args, **kwargs)

    def bind_diffusion(self, schedule: Dict[str, torch.Tensor], n_timesteps: int,
                       predict_epsilon: bool, clip_denoised: bool) -> None:
        """Called by ``GaussianDiffusion``: hands over the schedule scalars the fused
        posterior kernel needs (diffusion.py:117-128)."""
        self._schedule = schedule
        self._diffusion_opts = {"n_timesteps": int(n_timesteps),
                                "predict_epsilon": bool(predict_epsilon),
                                "clip_denoised": bool(clip_denoised)}

    def _signature(self, horizon: int, device: torch.device):
        """(what the engine's structure depends on, the parameters' identities and versions)"""
        params = tuple((p.data_ptr(), p._version) for p in self._params().values())
        opts = tuple(sorted(self._diffusion_opts.items()))
        sched = None
        if self._schedule is not None:
            sched = tuple((k, v.data_ptr(), v._version) for k, v in sorted(self._schedule.items()))
        return (horizon, str(device), opts, sched, self.precision, bool(self.small_batch_kernels)), params

    def engine(self, horizon: int, device: torch.device, training: bool = False) -> HipEngine:
        """Return the engine for (horizon, device), (re)building it if weights, schedule or
        options changed since the last call."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(
                "TemporalUnet runs on the HIP engine only: move the model and its inputs to a "
                f"ROCm device (got {device}); there is no CPU fallback")
        sig, params = self._signature(horizon, device)
        if self._engine is not None and sig == self._engine_sig and (self._engine.training or not training):
            if params == self._engine_params:
                return self._engine
            if self.precision == "fp32" and all(p.device == device for p in self._params().values()):
                # only parameter VALUES changed (an optimiser step): the packed copies are re-derived on the
                # device, no engine rebuild and no host round trip
                self._engine.refresh(self._params())
                self._engine_params = params
                return self._engine
        opts = self._diffusion_opts
        T = opts["n_timesteps"]
        eng = HipEngine(transition_dim=self.transition_dim, dim=self.dim, channels=self.channels,
                        horizon=horizon, n_timesteps=T, time_dim=self.time_dim,
                        kernel_size=self.kernel_size, predict_epsilon=opts["predict_epsilon"],
                        clip_denoised=opts["clip_denoised"], device=device,
                        precision=self.precision, training=training)
        if self._schedule is not None:
            sched = self._schedule
        else:
            zero = torch.zeros(T)
            sched = {k: zero for k in ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                                       "posterior_mean_coef1", "posterior_mean_coef2",
                                       "posterior_log_variance_clipped")}
        eng.load(self._params(), sched)
        eng.debug_set_option("cc", int(bool(self.small_batch_kernels)))
        self._engine, self._engine_sig, self._engine_params = eng, sig, params
        return eng

    # ------------------------------------------------------------------ forward
    @staticmethod
    def shared_timestep(time: Union[int, torch.Tensor]) -> int:
        if isinstance(time, int):
            return time
        lo, hi = int(time.min()), int(time.max())
        if lo != hi:
            raise NotImplementedError(
                "the HIP sampler evaluates one timestep per call (all rows share t, as in "
                "p_sample_loop / sample_loop); got timesteps in [%d, %d]" % (lo, hi))
        return lo

    # ------------------------------------------------------------------ differentiable forward
    def _block_order(self):
        """ResidualTemporalBlock prefixes in launch order — the order in which the engine lays their
        time projections side by side (csrc/host_plan.hpp build_plan; temporal_unet.py:199-233)."""
        n = len(self.dim_mults)
        order = []
        for i in range(n):
            order += [f"downs.{i}.0", f"downs.{i}.1"]
        order += ["mid_block1", "mid_block2"]
        for j in range(n - 1):
            order += [f"ups.{j}.0", f"ups.{j}.1"]
        return order

    def _time_projections(self, time: torch.Tensor) -> torch.Tensor:
        """(B, sum C_out) = every block's Linear(Mish(time_mlp(SinusoidalPosEmb(t)))) side by side, with
        torch ops (temporal_unet.py:19-32,97-100,155-160): autograd differentiates these."""
        F = torch.nn.functional
        p = self._params()
        half = self.dim // 2
        scale = math.log(10000) / (half - 1)
        freqs = torch.exp(torch.arange(half, device=time.device) * -scale)
        arg = time[:, None] * freqs[None, :]
        emb = torch.cat((arg.sin(), arg.cos()), dim=-1)
        temb = F.linear(F.mish(F.linear(emb, p["time_mlp.1.weight"], p["time_mlp.1.bias"])),
                        p["time_mlp.3.weight"], p["time_mlp.3.bias"])
        act = F.mish(temb)
        # every block's Linear(time_dim -> C_out) as ONE GEMM over the concatenated weights (the same dot
        # products; a dozen separate 256 x 128 x C_out GEMMs cost ~38 us each in hipBLASLt, forward and backward)
        blocks = self._block_order()
        weight = torch.cat([p[b + ".time_mlp.1.weight"] for b in blocks], dim=0)
        bias = torch.cat([p[b + ".time_mlp.1.bias"] for b in blocks], dim=0)
        return F.linear(act, weight, bias).contiguous()

    def _forward_autograd(self, x: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        eng = self.engine(int(x.shape[1]), x.device, training=True)
        layout, _ = eng.grad_layout()
        params = self._params()
        rows = self._time_projections(time.reshape(-1).to(x.device))
        tensors = [params[k] for k, _, _ in layout]
        if eng.widths_padded:
            # zero-padded widths (utils/padding.py): the engine reads padded projection rows (scattered here with a torch
            # op, so autograd gathers their gradient back) and returns the real entries of its padded parameter gradients
            index, width = eng.time_projection_index()
            rows = torch.zeros(rows.shape[0], width, dtype=rows.dtype, device=rows.device).index_copy(1, index, rows)
        return _UnetFunction.apply(eng, layout, x.contiguous().float(), rows, *tensors)

    def forward(self, x: torch.Tensor, time: Union[int, torch.Tensor]) -> torch.Tensor:
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._params().values())):
            if isinstance(time, int):
                time = torch.full((x.shape[0],), time, device=x.device, dtype=torch.long)
            return self._forward_autograd(x, time)        # (any t: the sinusoid is evaluated, not looked up)
        with torch.no_grad():
            return self._forward_inference(x, time)

    def _forward_inference(self, x: torch.Tensor, time: Union[int, torch.Tensor]) -> torch.Tensor:
        if isinstance(time, int):
            lo = hi = time
        else:
            lo, hi = int(time.min()), int(time.max())
        if hi >= self._diffusion_opts["n_timesteps"] and self._schedule is None:
            # bare denoiser: grow the time table on demand
            self._diffusion_opts["n_timesteps"] = max(2 * self._diffusion_opts["n_timesteps"], hi + 1)
        if lo < 0 or hi >= self._diffusion_opts["n_timesteps"]:
            raise RuntimeError(f"index {hi if hi >= 0 else lo} is out of bounds for dimension 0 with size "
                               f"{self._diffusion_opts['n_timesteps']}")
        eng = self.engine(int(x.shape[1]), x.device)
        if lo == hi:
            return eng.unet_forward(x.contiguous().float(), lo)
        return eng.unet_forward_rows(x.contiguous().float(), time.reshape(-1))


class _UnetFunction(torch.autograd.Function):
    """eps_theta(x) on the HIP engine with an explicit backward pass (include/dad.h, training side).
    Inputs: the trajectory, the per-row time projections, then every conv / GroupNorm parameter in the
    order of ``HipEngine.grad_layout()`` (their VALUES live packed inside the engine; they are inputs
    here so that autograd routes their gradients)."""

    @staticmethod
    def forward(ctx, eng, layout, x, rows, *params):
        out, saved = eng.train_forward(x, rows.detach().contiguous().float())
        ctx.eng, ctx.layout, ctx.saved = eng, layout, saved
        ctx.temb_width = int(rows.shape[1])
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        d_x, d_rows, grads = ctx.eng.train_backward(x, d_out.contiguous().float(), ctx.saved, ctx.temb_width, ctx.shapes)
        ctx.saved = None
        return (None, None, d_x, d_rows, *grads)
