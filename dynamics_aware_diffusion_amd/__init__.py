"""MI355X-native reverse-diffusion planning sampler.

Drop-in for the sampling path of ``m_diffuser`` (darshangm/dynamics-aware-diffusion):
``models.TemporalUnet`` / ``models.GaussianDiffusion`` / ``guides.*Policy`` /
``dynamics.ProjectionMatrixBuilder`` keep the reference's names and signatures; the
arithmetic runs in ``libdad_hip.so`` (hand-written gfx950 kernels, C ABI in
``include/dad.h``).  No CPU fallback exists.
"""
from . import dynamics, guides, losses, models
from .guides import DynamicsAwarePolicy, GuidedPolicy, MPCPolicy, ValueGuidedPolicy
from .models import GaussianDiffusion, TemporalUnet
from .utils.checkpoint import infer_architecture, load_checkpoint

__version__ = "0.1.0"
__all__ = ["models", "guides", "dynamics", "losses", "TemporalUnet", "GaussianDiffusion", "GuidedPolicy",
           "MPCPolicy", "ValueGuidedPolicy", "DynamicsAwarePolicy", "load_checkpoint",
           "infer_architecture"]
