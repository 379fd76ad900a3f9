"""Oracle (test infrastructure — see oracle/__init__.py): CPU restatement of the
reverse-diffusion sampler.  Parity pinned by tests/golden (made from the real reference).

Every function cites the reference lines it follows (paths under /root/reference/).
Weights come in as a flat ``{key: tensor}`` dict using the reference's state_dict keys
*without* the ``model.`` prefix.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- schedule
def beta_schedule(name: str, n_timesteps: int) -> Tensor:
    """m_diffuser/models/diffusion.py:32-48 (cosine s=0.008 / linear 1e-4..0.02), fp32."""
    if name == "cosine":
        s = 0.008
        grid = torch.linspace(0, n_timesteps, n_timesteps + 1)
        abar = torch.cos(((grid / n_timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
        abar = abar / abar[0]
        return torch.clip(1 - (abar[1:] / abar[:-1]), 0.0001, 0.9999)
    if name == "linear":
        return torch.linspace(1e-4, 0.02, n_timesteps)
    raise ValueError(f"Unknown beta schedule: {name}")          # diffusion.py:102


def schedule_buffers(name: str, n_timesteps: int) -> Dict[str, Tensor]:
    """The 12 registered buffers, m_diffuser/models/diffusion.py:104-128."""
    betas = beta_schedule(name, n_timesteps)
    alphas = 1.0 - betas
    abar = torch.cumprod(alphas, dim=0)
    abar_prev = torch.cat([torch.ones(1), abar[:-1]])
    post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": abar,
        "alphas_cumprod_prev": abar_prev,
        "sqrt_alphas_cumprod": torch.sqrt(abar),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - abar),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / abar),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / abar - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(torch.clamp(post_var, min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(abar_prev) / (1.0 - abar),
        "posterior_mean_coef2": (1.0 - abar_prev) * torch.sqrt(alphas) / (1.0 - abar),
    }


# ------------------------------------------------------------------------------ denoiser
def sinusoidal_embedding(t: Tensor, dim: int) -> Tensor:
    """m_diffuser/models/temporal_unet.py:19-32.  Always fp32 (arange->exp in fp32)."""
    half = dim // 2
    scale = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half) * -scale)
    arg = t[:, None] * freqs[None, :]
    return torch.cat([arg.sin(), arg.cos()], dim=-1)


def infer_arch(w: Mapping[str, Tensor]) -> Dict[str, object]:
    """Read (transition_dim, dim, per-level channels, kernel) off the weight shapes."""
    n_levels = 1 + max(int(k.split(".")[1]) for k in w if k.startswith("downs."))
    chans = [int(w[f"downs.{i}.0.blocks.0.block.0.weight"].shape[0]) for i in range(n_levels)]
    first = w["downs.0.0.blocks.0.block.0.weight"]
    return {
        "transition_dim": int(first.shape[1]),
        "kernel_size": int(first.shape[2]),
        "dim": int(w["time_mlp.1.weight"].shape[1]),
        "channels": chans,
        "n_levels": n_levels,
    }


def conv_block(w: Mapping[str, Tensor], base: str, x: Tensor) -> Tensor:
    """Conv1d(pad k//2) -> GroupNorm(8) -> Mish, temporal_unet.py:57-76."""
    cw = w[base + ".block.0.weight"]
    y = F.conv1d(x, cw, w[base + ".block.0.bias"], padding=cw.shape[2] // 2)
    y = F.group_norm(y, 8, w[base + ".block.1.weight"], w[base + ".block.1.bias"], eps=1e-5)
    return F.mish(y)


def residual_block(w: Mapping[str, Tensor], base: str, x: Tensor, temb: Tensor) -> Tensor:
    """temporal_unet.py:106-122: B1(B0(x) + Linear(Mish(temb))[:, :, None]) + res(x)."""
    h = conv_block(w, base + ".blocks.0", x)
    tproj = F.linear(F.mish(temb), w[base + ".time_mlp.1.weight"], w[base + ".time_mlp.1.bias"])
    h = conv_block(w, base + ".blocks.1", h + tproj[:, :, None])
    if (base + ".residual_conv.weight") in w:
        res = F.conv1d(x, w[base + ".residual_conv.weight"], w[base + ".residual_conv.bias"])
    else:
        res = x
    return h + res


def unet_forward(w: Mapping[str, Tensor], x: Tensor, t: Tensor,
                 taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """TemporalUnet.forward, temporal_unet.py:199-241.  x (B,H,td), t (B,) int64.

    Reproduces the reference's decoder quirk: every decoder stage upsamples and the
    level-0 skip is pushed but never popped (temporal_unet.py:184-191,221,230).
    ``taps`` (optional dict) receives per-stage intermediates for unit-level parity.
    """
    arch = infer_arch(w)
    n_levels = arch["n_levels"]
    dt = w["time_mlp.1.weight"].dtype
    h = x.transpose(1, 2)
    emb = sinusoidal_embedding(t, arch["dim"]).to(dt)        # fp32 sinusoid, cast for fp64 runs
    temb = F.linear(emb, w["time_mlp.1.weight"], w["time_mlp.1.bias"])
    temb = F.linear(F.mish(temb), w["time_mlp.3.weight"], w["time_mlp.3.bias"])
    if taps is not None:
        taps["temb"] = temb
    skips: List[Tensor] = []
    for i in range(n_levels):
        h = residual_block(w, f"downs.{i}.0", h, temb)
        h = residual_block(w, f"downs.{i}.1", h, temb)
        skips.append(h)
        if taps is not None:
            taps[f"downs.{i}"] = h
        if i < n_levels - 1:
            h = F.conv1d(h, w[f"downs.{i}.2.conv.weight"], w[f"downs.{i}.2.conv.bias"],
                         stride=2, padding=1)
    h = residual_block(w, "mid_block1", h, temb)
    h = residual_block(w, "mid_block2", h, temb)
    if taps is not None:
        taps["mid"] = h
    for j in range(n_levels - 1):
        h = torch.cat([h, skips.pop()], dim=1)
        h = residual_block(w, f"ups.{j}.0", h, temb)
        h = residual_block(w, f"ups.{j}.1", h, temb)
        h = F.conv_transpose1d(h, w[f"ups.{j}.2.conv.weight"], w[f"ups.{j}.2.conv.bias"],
                               stride=2, padding=1)
        if taps is not None:
            taps[f"ups.{j}"] = h
    h = conv_block(w, "final_conv.0", h)
    h = F.conv1d(h, w["final_conv.1.weight"], w["final_conv.1.bias"])
    return h.transpose(1, 2)


# ------------------------------------------------------------------------ reverse process
def _at(buf: Tensor, t: Tensor, ndim: int) -> Tensor:
    """extract(), diffusion.py:15-29 — gather raises if t >= len(buf)."""
    return buf.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def p_mean_variance(w, sched: Mapping[str, Tensor], x: Tensor, t: Tensor,
                    clip_denoised: bool = True, predict_epsilon: bool = True
                    ) -> Tuple[Tensor, Tensor, Tensor]:
    """diffusion.py:182-203 (+159-180).  Returns (mean, log_var (B,1,1), model_out)."""
    out = unet_forward(w, x, t)
    if predict_epsilon:
        x0 = _at(sched["sqrt_recip_alphas_cumprod"], t, x.dim()) * x \
            - _at(sched["sqrt_recipm1_alphas_cumprod"], t, x.dim()) * out
    else:
        x0 = out
    if clip_denoised:
        x0 = torch.clamp(x0, -1.0, 1.0)
    mean = _at(sched["posterior_mean_coef1"], t, x.dim()) * x0 \
        + _at(sched["posterior_mean_coef2"], t, x.dim()) * x
    return mean, _at(sched["posterior_log_variance_clipped"], t, x.dim()), out


def denoise_step(w, sched, x: Tensor, t: Tensor, noise: Tensor,
                 conditions: Optional[Dict[int, Tensor]] = None,
                 guide_grad: Optional[Tensor] = None, guide_weight: float = 0.0,
                 clip_denoised: bool = True, predict_epsilon: bool = True) -> Tensor:
    """One reverse step with injected noise.

    diffusion.py:205-223 (p_sample) and guides/policies.py:65-112
    (p_sample_with_guidance): mean [+ w*exp(logvar)*grad] + [t!=0]*exp(0.5*logvar)*z,
    then x[:, k] = cond_k for every condition (policies.py:48-63; whole transition,
    action channels included).
    """
    mean, logvar, _ = p_mean_variance(w, sched, x, t, clip_denoised, predict_epsilon)
    if guide_grad is not None and guide_weight > 0:
        mean = mean + guide_weight * logvar.exp() * guide_grad
    mask = (t != 0).to(x.dtype).view(-1, *([1] * (x.dim() - 1)))
    x_prev = mean + mask * torch.exp(0.5 * logvar) * noise
    if conditions is not None:
        for k, val in conditions.items():
            x_prev[:, k] = val
    return x_prev


def guide_gradient(guide_fn: Callable[[Tensor, Tensor], Tensor], x: Tensor, t: Tensor) -> Tensor:
    """policies.py:87-94: d sum(guide(x_t, t)) / d x_t, evaluated at x_t (not the mean)."""
    xg = x.detach().requires_grad_(True)
    with torch.enable_grad():
        score = guide_fn(xg, t)
        (grad,) = torch.autograd.grad(score.sum(), xg)
    return grad.detach()


def sample_loop(w, sched, noise: Tensor, n_timesteps: int,
                conditions: Optional[Dict[int, Tensor]] = None,
                guide_fn: Optional[Callable] = None, guide_weight: float = 0.0,
                clip_denoised: bool = True, predict_epsilon: bool = True,
                post_step: Optional[Callable[[Tensor, int], Tensor]] = None,
                trace: Optional[List[Tensor]] = None) -> Tensor:
    """Full ancestral loop with an injected noise stack ``noise[(T+1), B, H, td]``.

    ``noise[0]`` is x_T; ``noise[1+j]`` is the z drawn at loop iteration j (t = T-1-j),
    i.e. the order in which the reference calls ``randn`` (diffusion.py:241,218;
    policies.py:134,100; z is drawn at t=0 too and then masked).  ``n_timesteps`` may be
    smaller than the trained schedule (evaluate.py:350-353 truncation semantics).
    ``post_step(x, i)`` models README's x_{i-1} = project(denoise(x_i)) opt-in.
    """
    with torch.no_grad():
        x = noise[0].clone()
        B = x.shape[0]
        if conditions is not None:
            for k, val in conditions.items():
                x[:, k] = val
        for j, i in enumerate(reversed(range(n_timesteps))):
            t = torch.full((B,), i, dtype=torch.long)
            grad = None
            if guide_fn is not None and guide_weight > 0:
                grad = guide_gradient(guide_fn, x, t)
            x = denoise_step(w, sched, x, t, noise[1 + j], conditions, grad, guide_weight,
                             clip_denoised, predict_epsilon)
            if post_step is not None:
                x = post_step(x, i)
            if trace is not None:
                trace.append(x.clone())
        return x


def training_loss(w, sched: Mapping[str, Tensor], x_start: Tensor, t: Tensor, noise: Tensor,
                  loss_type: str = "l2", predict_epsilon: bool = True,
                  weights: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """GaussianDiffusion.loss (diffusion.py:253-290) with its two random draws (t, noise) passed in:
    x_t = q_sample (diffusion.py:138-157), model on per-row timesteps, elementwise L1 / L2
    (diffusion.py:130-136), optional weights, mean.  Returns (loss, x_t, model output)."""
    with torch.no_grad():
        x_t = _at(sched["sqrt_alphas_cumprod"], t, x_start.dim()) * x_start \
            + _at(sched["sqrt_one_minus_alphas_cumprod"], t, x_start.dim()) * noise
        out = unet_forward(w, x_t, t)
        target = noise if predict_epsilon else x_start
        if loss_type == "l2":
            per = (out - target) ** 2
        elif loss_type == "l1":
            per = (out - target).abs()
        else:
            raise ValueError(f"Unknown loss type: {loss_type}")              # diffusion.py:136
        if weights is not None:
            per = per * weights
        return per.mean(), x_t, out


def training_gradients(w: Mapping[str, Tensor], sched: Mapping[str, Tensor], x_start: Tensor, t: Tensor,
                       noise: Tensor, loss_type: str = "l2", predict_epsilon: bool = True,
                       weights: Optional[Tensor] = None) -> Tuple[Tensor, Dict[str, Tensor], Tensor]:
    """What the reference's training step computes with ``loss.backward()`` (utils/training.py:152-156)
    for GaussianDiffusion.loss (diffusion.py:253-290), its two random draws passed in: torch autograd
    over this file's restatement of the forward.  Returns (loss, {key: d loss / d weight}, d loss / d x_t)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    x_t = (_at(sched["sqrt_alphas_cumprod"], t, x_start.dim()) * x_start
           + _at(sched["sqrt_one_minus_alphas_cumprod"], t, x_start.dim()) * noise).detach().requires_grad_(True)
    with torch.enable_grad():
        out = unet_forward(leaves, x_t, t)
        target = noise if predict_epsilon else x_start
        per = (out - target) ** 2 if loss_type == "l2" else (out - target).abs()
        if weights is not None:
            per = per * weights
        loss = per.mean()
        loss.backward()
    return loss.detach(), {k: v.grad.detach() for k, v in leaves.items()}, x_t.grad.detach()


def cast_weights(w: Mapping[str, Tensor], dtype: torch.dtype) -> Dict[str, Tensor]:
    return {k: v.to(dtype) for k, v in w.items()}


# -------------------------------------------------------------------------- planner glue
def plan_actions(traj0: Tensor, observation_dim: int, action_dim: int, action_horizon: int,
                 unnormalize_actions: Callable) -> List:
    """policies.py:181-191: actions t = 0 .. min(action_horizon, H-1) of trajectory 0,
    un-normalised, in FIFO order (the buffer therefore holds min(a+1, H) entries)."""
    H = traj0.shape[0]
    arr = traj0.cpu().numpy()
    out = []
    for step in range(0, min(action_horizon + 1, H)):
        a = arr[step, observation_dim:observation_dim + action_dim]
        out.append(unnormalize_actions(a.reshape(1, -1)).flatten())
    return out
