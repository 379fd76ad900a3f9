"""GPU parity of the split-f16 conv arithmetic (``TemporalUnet.precision = "f16x3"``,
include/dad.h DAD_PREC_F16X3): every fp32 operand is carried as hi + lo*2^-11 halves and a
product block is three v_mfma_f32_32x32x16_f16 with fp32 accumulation.

The gates are the SAME as for the exact-fp32 MFMA path — the tests below are the parity tests of
tests/test_hip_parity.py / tests/test_hip_extra.py re-run with the split kernels: golden vectors
of the real reference (<= 5e-6 per forward / step, <= 2e-5 per loop, and no farther from the
reference's fp64 run than 2x the reference's own fp32 arithmetic + 5e-7), every tile variant,
grid-level split-K, ragged batches, the wide architectures and the full-size B=256 properties.
"""
import pytest
import torch

from tests import test_hip_parity as _parity
from tests.test_hip_parity import (  # noqa: F401  (collected here under the split fixture)
    dev,
    test_unet_forward_vs_reference,
    test_sampling_loops_vs_reference,
    test_graph_replay_matches_eager,
    test_value_guidance_vs_reference,
    test_get_action_glue_vs_reference,
    test_t1000_loops_vs_reference,
    test_projected_loops_vs_reference,
    test_diffusion_options_vs_reference,
    test_training_objective_forward_vs_reference,
)
from tests.test_hip_extra import (  # noqa: F401
    test_every_tile_variant_matches_oracle,
    test_grid_split_k_is_exact_to_rounding_and_deterministic,
    test_wide_group_tiles_on_big_architectures,
    test_wide_nets_at_a_ragged_multi_tile_batch,
    test_ragged_batches_match_oracle,
    test_philox_sampling_is_sharding_invariant_and_deterministic,
    test_full_size_properties,
    test_weights_refresh_after_load_state_dict,
    test_assorted_architectures_match_oracle,
    test_batched_get_actions_matches_per_env_oracle,
)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def split_precision():
    from dynamics_aware_diffusion_amd import TemporalUnet
    _parity._MODELS.clear()
    TemporalUnet.default_precision = "f16x3"
    yield
    TemporalUnet.default_precision = "fp32"
    _parity._MODELS.clear()


def test_split_kernels_are_the_ones_running(dev):
    """The split path is not a silent alias of the fp32 one: same inputs, the two precisions
    differ in the last bits (and only there)."""
    from dynamics_aware_diffusion_amd import TemporalUnet
    torch.manual_seed(0)
    a = TemporalUnet(6, dim=128, dim_mults=(1, 2, 4)).to(dev)
    assert a.precision == "f16x3"
    b = TemporalUnet(6, dim=128, dim_mults=(1, 2, 4)).to(dev)
    b.load_state_dict(a.state_dict())
    b.precision = "fp32"
    x = torch.randn(8, 32, 6, device=dev)
    t = torch.full((8,), 17, device=dev, dtype=torch.long)
    ya, yb = a(x, t), b(x, t)
    d = float((ya - yb).abs().max())
    print(f"|f16x3 - fp32| = {d:.2e}")
    assert 0.0 < d <= 5e-6


def test_saturating_activations_do_not_poison(dev):
    """|x| beyond the f16 range saturates instead of producing inf / NaN."""
    from dynamics_aware_diffusion_amd import TemporalUnet
    torch.manual_seed(1)
    net = TemporalUnet(6, dim=128, dim_mults=(1, 2, 4)).to(dev)
    x = torch.randn(4, 32, 6, device=dev)
    x[0, 3, 2] = 3.0e5
    y = net(x, torch.full((4,), 5, device=dev, dtype=torch.long))
    assert bool(torch.isfinite(y).all())
