"""Checkpoint -> ready-to-sample ``GaussianDiffusion``.

Replaces ``scripts/evaluate.py:64-203`` (``infer_model_config_from_checkpoint`` + ``load_model``)
of the reference for the dict its trainer writes (``m_diffuser/utils/training.py:191-211``):

    {'epoch', 'global_step', 'model_state_dict', 'optimizer_state_dict',
     'config': {'horizon', 'observation_dim', 'action_dim', 'n_timesteps', 'beta_schedule'},
     ['ema_state_dict'], ['scheduler_state_dict']}

Differences from the reference's loader, all deliberate:
  * the architecture is read off tensor SHAPES — ``transition_dim``, ``dim``, the per-level widths
    (hence any ``dim_mults``, e.g. HalfCheetah's (1, 4, 8)), ``time_dim``, ``kernel_size`` — where
    the reference guesses ``dim_mults`` from the level COUNT and mis-builds non-power-of-two nets
    (``evaluate.py:90-99``; SURVEY finding F9);
  * ``observation_dim`` / ``action_dim`` come from the checkpoint's own ``config`` (or arguments),
    not from re-loading the Minari dataset (``evaluate.py:162-168``);
  * ``use_ema=True`` selects ``ema_state_dict`` (``training.py:208``), which the reference's
    ``evaluate.py`` never does (Appendix D.9).
"""
from __future__ import annotations

import os
from typing import Any, Dict, Mapping, Optional, Union

import torch

from ..models.diffusion import GaussianDiffusion
from ..models.temporal_unet import TemporalUnet

_FIRST_CONV = "downs.0.0.blocks.0.block.0.weight"


def _strip_prefix(state: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Denoiser tensors keyed without the ``model.`` prefix (bare TemporalUnet dicts pass
    through; optimizer-style ``module.`` wrappers are unwrapped)."""
    out = {}
    for k, v in state.items():
        while k.startswith("module."):
            k = k[len("module."):]
        out[k] = v
    if any(k.startswith("model.") for k in out):
        return {k[len("model."):]: v for k, v in out.items() if k.startswith("model.")}
    return {k: v for k, v in out.items() if "." in k}          # drops the 12 schedule buffers


def infer_architecture(state: Mapping[str, torch.Tensor]) -> Dict[str, Any]:
    """Architecture of a TemporalUnet from the shapes of its ``state_dict`` (with or without the
    ``model.`` prefix): everything ``TemporalUnet.__init__`` (temporal_unet.py:135-197) needs."""
    w = _strip_prefix(state)
    if _FIRST_CONV not in w:
        raise KeyError(f"not a TemporalUnet state_dict: '{_FIRST_CONV}' is missing "
                       f"(keys start with {sorted(w)[:3]} ...)")
    levels = sorted({int(k.split(".")[1]) for k in w if k.startswith("downs.") and k.split(".")[1].isdigit()})
    if levels != list(range(len(levels))):
        raise KeyError(f"encoder levels are not contiguous: {levels}")
    first = w[_FIRST_CONV]
    channels = []
    for i in levels:
        key = f"downs.{i}.0.blocks.0.block.0.weight"
        if key not in w:
            raise KeyError(f"missing key '{key}'")
        channels.append(int(w[key].shape[0]))
    if "time_mlp.1.weight" not in w or "time_mlp.3.weight" not in w:
        raise KeyError("missing key 'time_mlp.1.weight' / 'time_mlp.3.weight'")
    dim = int(w["time_mlp.1.weight"].shape[1])                 # SinusoidalPosEmb(dim) feeds Linear(dim, 4*time_dim)
    time_dim = int(w["time_mlp.3.weight"].shape[0])
    if any(c % dim for c in channels):
        raise ValueError(f"level widths {channels} are not multiples of dim={dim} "
                         "(the reference builds dims = [transition_dim, dim*m for m in dim_mults])")
    if channels[0] != dim:
        # constructible in the reference (dims = dim * m), but its forward cannot run: final_conv is
        # Conv1dBlock(dim, dim) on a decoder that ends with dim * dim_mults[0] channels
        # (/root/reference/m_diffuser/models/temporal_unet.py:193-197)
        raise ValueError(f"dim_mults[0] = {channels[0] // dim}: the reference's final_conv takes dim={dim} "
                         f"channels, the decoder of this checkpoint ends with {channels[0]}")
    return {
        "transition_dim": int(first.shape[1]),
        "kernel_size": int(first.shape[2]),
        "dim": dim,
        "time_dim": time_dim,
        "channels": channels,
        "dim_mults": tuple(c // dim for c in channels),
    }


def load_checkpoint(checkpoint: Union[str, os.PathLike, Mapping[str, Any]],
                    device: Union[str, torch.device] = "cuda", use_ema: bool = False, *,
                    horizon: Optional[int] = None, observation_dim: Optional[int] = None,
                    action_dim: Optional[int] = None, beta_schedule: Optional[str] = None,
                    precision: Optional[str] = None, strict: bool = True,
                    weights_only: bool = True) -> GaussianDiffusion:
    """Build the sampler for a reference checkpoint and load its weights.

    ``checkpoint``: a path (``torch.load``-ed on the CPU) or the dict itself.  ``use_ema`` picks
    ``ema_state_dict`` instead of ``model_state_dict``.  ``horizon`` / ``observation_dim`` /
    ``action_dim`` / ``beta_schedule`` override the checkpoint's ``config`` (needed for old
    checkpoints saved without one).  Returns the model in eval mode on ``device``.
    ``weights_only`` (default True) refuses to unpickle arbitrary objects: the dict the reference's
    trainer writes (tensors, numbers, strings; utils/training.py:193-211) loads under it; pass
    ``weights_only=False`` for a trusted file that holds anything else (what
    scripts/evaluate.py:140 does unconditionally).
    """
    if not isinstance(checkpoint, Mapping):
        checkpoint = torch.load(os.fspath(checkpoint), map_location="cpu", weights_only=bool(weights_only))
    which = "ema_state_dict" if use_ema else "model_state_dict"
    if which not in checkpoint:
        have = sorted(k for k in checkpoint if k.endswith("state_dict"))
        raise KeyError(f"checkpoint has no '{which}' (state dicts present: {have})")
    state = checkpoint[which]
    arch = infer_architecture(state)
    cfg = dict(checkpoint.get("config") or {})

    def pick(name, override, default=None):
        if override is not None:
            return override
        if name in cfg:
            return cfg[name]
        if default is not None:
            return default
        raise KeyError(f"checkpoint['config'] has no '{name}': pass {name}=... to load_checkpoint")

    n_timesteps = int(state["betas"].shape[0]) if "betas" in state else int(pick("n_timesteps", None))
    horizon = int(pick("horizon", horizon))
    obs = int(pick("observation_dim", observation_dim))
    act = int(pick("action_dim", action_dim))
    if obs + act != arch["transition_dim"]:
        raise ValueError(f"observation_dim {obs} + action_dim {act} != transition_dim "
                         f"{arch['transition_dim']} of the checkpoint's first conv")
    unet = TemporalUnet(arch["transition_dim"], dim=arch["dim"], dim_mults=arch["dim_mults"],
                        kernel_size=arch["kernel_size"], time_dim=arch["time_dim"])
    if precision is not None:
        unet.precision = precision
    diffusion = GaussianDiffusion(unet, horizon, obs, act, n_timesteps=n_timesteps,
                                  beta_schedule=pick("beta_schedule", beta_schedule, "cosine"))
    # accept both the trainer's GaussianDiffusion dict ("model." prefix + schedule buffers) and a
    # bare TemporalUnet dict
    clean = {}
    for k, v in state.items():
        while k.startswith("module."):
            k = k[len("module."):]
        clean[k] = v
    if not any(k.startswith("model.") for k in clean):
        clean = {"model." + k: v for k, v in clean.items()}
        strict_here = False                                     # schedule buffers come from the constructor
    else:
        strict_here = strict
    missing, unexpected = diffusion.load_state_dict(clean, strict=False)
    missing = [k for k in missing if k.startswith("model.") or strict_here]
    if strict and (missing or unexpected):
        raise KeyError(f"checkpoint does not match the inferred architecture {arch['dim_mults']} x "
                       f"dim {arch['dim']}: missing {missing[:4]}{' ...' if len(missing) > 4 else ''}, "
                       f"unexpected {list(unexpected)[:4]}{' ...' if len(unexpected) > 4 else ''}")
    diffusion.loaded_from = {"state": which, "epoch": checkpoint.get("epoch"),
                             "global_step": checkpoint.get("global_step"), **arch}
    return diffusion.to(torch.device(device)).eval()
