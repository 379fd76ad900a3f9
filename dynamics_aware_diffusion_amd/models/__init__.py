from .diffusion import GaussianDiffusion, cosine_beta_schedule, extract, linear_beta_schedule
from .temporal_unet import TemporalUnet

__all__ = ["TemporalUnet", "GaussianDiffusion", "cosine_beta_schedule", "linear_beta_schedule",
           "extract"]
