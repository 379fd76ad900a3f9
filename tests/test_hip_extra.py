"""GPU tests beyond the golden vectors: every conv tile variant, ragged / edge batch sizes,
per-row conditions, the in-kernel noise generator, sharding invariance, the opt-in projection
inside the loop, and size-independent properties at the benchmark's full batch."""
import numpy as np
import pytest
import torch

from oracle import denoiser as orc
from oracle import philox as ophilox
from oracle import projection as oproj
from tests.golden import cases
from tests.test_hip_parity import TOL_LOOP, TOL_STEP, build, eps_gain
from tests.util import golden, max_abs, net_weights_torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle_eps(net, x, t):
    with torch.no_grad():
        return orc.unet_forward(net_weights_torch(net), x, torch.full((x.shape[0],), t, dtype=torch.long))


@pytest.mark.parametrize("tile", list(range(10)))
def test_every_tile_variant_matches_oracle(tile, dev):
    """Force each (BM, BN, split-K, K-chunk) instantiation wherever it is valid."""
    from dynamics_aware_diffusion_amd.utils import synth
    for net, B in (("tiny", 5), ("pointmaze", 9)):
        diff = build(net, cases.NETS[net][4], "cosine", dev)
        eng = diff._engine(dev)
        try:
            eng.debug_set_option("cc", 0)               # tiles belong to the batch-256 kernels
            eng.debug_set_tile(tile)
            x = torch.from_numpy(synth.normal_like(61, f"tile{tile}.{net}", (B, 32, diff.transition_dim)))
            want = _oracle_eps(net, x, 3)
            got = diff.model(x.to(dev), 3)
            torch.cuda.synchronize()
            assert max_abs(got.cpu().numpy(), want.numpy()) <= TOL_STEP, (tile, net)
        finally:
            eng.debug_set_tile(-1)
            eng.debug_set_option("cc", int(diff.model.small_batch_kernels))


def test_grid_split_k_is_exact_to_rounding_and_deterministic(dev):
    """Small batches split K over several blocks per tile (last-arriver reduction in slice
    order): same answer as the unsplit kernel to fp32 rounding, bit-identical run to run."""
    from dynamics_aware_diffusion_amd.utils import synth
    diff = build("pointmaze", 100, "cosine", dev)
    eng = diff._engine(dev)
    eng.debug_set_option("cc", 0)                       # grid split-K lives in the batch-256 kernels
    for B in (1, 3, 8):
        x = torch.from_numpy(synth.normal_like(65, f"splitk.{B}", (B, 32, 6)))
        want = _oracle_eps("pointmaze", x, 42).numpy()
        xd = x.to(dev)
        try:
            eng.debug_set_tile(-1)
            a = diff.model(xd, 42).cpu().numpy()
            b = diff.model(xd, 42).cpu().numpy()
            eng.debug_set_tile(99)                       # heuristic tiles, split-K off
            c = diff.model(xd, 42).cpu().numpy()
        finally:
            eng.debug_set_tile(-1)
        assert np.array_equal(a, b)
        assert max_abs(a, want) <= TOL_STEP and max_abs(c, want) <= TOL_STEP
        assert max_abs(a, c) <= 1e-5
    eng.debug_set_option("cc", int(diff.model.small_batch_kernels))


def test_wide_group_tiles_on_big_architectures(dev):
    """HalfCheetah / Door exercise the 128- and 256-channel GroupNorm tiles (cfg 2 and 3)."""
    for name, net, B, t in cases.FORWARD_CASES[3:5]:
        g = golden(name)
        diff = build(net, cases.NETS[net][4], "cosine", dev)
        x = torch.from_numpy(cases.forward_input(name, net, B)).to(dev)
        got = diff.model(x, t).cpu().numpy()
        assert max_abs(got, g["eps"]) <= TOL_STEP


@pytest.mark.parametrize("net", ["halfcheetah", "door"])
def test_wide_nets_at_a_ragged_multi_tile_batch(net, dev):
    """B = 33 on the wide nets: several N tiles per layer (the last one partly empty), 128- and
    256-channel GroupNorm tiles, grid split-K on the deepest levels — against the oracle."""
    from dynamics_aware_diffusion_amd.utils import synth
    diff = build(net, cases.NETS[net][4], "cosine", dev)
    B, t = 33, 321
    x = torch.from_numpy(synth.normal_like(68, f"wide.{net}", (B, 32, diff.transition_dim)))
    want = _oracle_eps(net, x, t).numpy()
    got = diff.model(x.to(dev), t).cpu().numpy()
    assert max_abs(got, want) <= TOL_STEP


@pytest.mark.parametrize("net", ["halfcheetah_j", "door_j"])
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_wide_nets_nontrivial_affine_through_the_batch_kernels(net, precision, dev):
    """B = 33 with GroupNorm gamma / beta off their defaults on the 1024 / 2048-channel nets, through the
    batch-256 kernels (<128,64>, <256,32,BDIR>, grid split-K) in both conv arithmetics: a per-channel
    parameter indexing slip in the wide-group tiles would show here."""
    from dynamics_aware_diffusion_amd.utils import synth
    diff = build(net, cases.NETS[net][4], "cosine", dev)
    diff.model.precision = precision
    try:
        B, t = 33, 777
        x = torch.from_numpy(synth.normal_like(81, f"widej.{net}", (B, 32, diff.transition_dim)))
        want = _oracle_eps(net, x, t).numpy()
        got = diff.model(x.to(dev), t).cpu().numpy()
    finally:
        diff.model.precision = "fp32"
    assert max_abs(got, want) <= TOL_STEP


# Frozen draws of tests/fuzz_parity.py (widths >= 1024, batches 5..40, both arithmetics): the manual
# sweep found nothing in 400+ cases; these keep a fixed sample of it inside `-m gpu`.
# (dim, dim_mults, horizon, transition_dim, batch, precision, t)
FROZEN_FUZZ = [
    (128, (1, 8), 32, 11, 5, "fp32", 3), (128, (1, 8), 16, 7, 40, "f16x3", 17), (256, (1, 4), 32, 23, 9, "fp32", 0),
    (256, (1, 4), 8, 5, 33, "f16x3", 19), (256, (1, 8), 16, 14, 6, "fp32", 11), (256, (1, 8), 32, 3, 13, "f16x3", 8),
    (128, (1, 2, 8), 32, 9, 7, "fp32", 5), (128, (1, 4, 8), 16, 20, 21, "f16x3", 2), (256, (1, 2, 4), 32, 17, 16, "fp32", 14),
    (256, (1, 4, 4), 16, 6, 31, "f16x3", 9), (256, (1, 4, 8), 32, 12, 8, "fp32", 1), (256, (1, 8, 4), 32, 4, 5, "f16x3", 16),
    (128, (1, 8, 8), 32, 24, 12, "fp32", 7), (128, (1, 2, 4, 8), 32, 8, 10, "f16x3", 13), (256, (1, 2, 4, 8), 32, 19, 5, "fp32", 18),
    (256, (1, 1, 4, 8), 32, 2, 27, "f16x3", 4), (64, (1, 4, 16), 32, 10, 36, "fp32", 6), (64, (1, 16), 16, 15, 14, "f16x3", 10),
    (256, (1, 8, 8), 16, 21, 11, "fp32", 15), (128, (1, 8, 2), 32, 13, 24, "f16x3", 12),
]


@pytest.mark.parametrize("draw", FROZEN_FUZZ, ids=lambda d: "d%d_m%s_H%d_td%d_B%d_%s" % (d[0], "x".join(map(str, d[1])), d[2], d[3], d[4], d[5]))
def test_frozen_fuzz_draws_match_oracle(draw, dev):
    from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
    from dynamics_aware_diffusion_amd.utils import synth
    dim, mults, H, td, B, prec, t = draw
    tag = "frozen.%d.%s.%d.%d" % (dim, "-".join(map(str, mults)), H, td)
    state = synth.synth_unet_state(td, dim, mults, seed=91, affine_jitter=0.3)
    w = {k: torch.from_numpy(v) for k, v in state.items()}
    unet = TemporalUnet(td, dim=dim, dim_mults=mults)
    unet.load_state_dict(w)
    unet.precision = prec
    diff = GaussianDiffusion(unet, H, td - 1, 1, n_timesteps=20).to(dev)
    x = torch.from_numpy(synth.normal_like(92, tag, (B, H, td)))
    with torch.no_grad():
        want = orc.unet_forward(w, x, torch.full((B,), t, dtype=torch.long))
    got = diff.model(x.to(dev), t)
    torch.cuda.synchronize()
    assert max_abs(got.cpu().numpy(), want.numpy()) <= TOL_STEP


@pytest.mark.parametrize("net,B", [("halfcheetah", 8), ("halfcheetah", 12), ("door", 16)])
def test_wide_nets_at_a_few_environments_default_path(net, B, dev):
    """`get_actions` for 8..16 environments on the wide nets: past `ccw_max_rows` the batch takes the
    batch-256 kernels with grid-level split-K (whole slices rounded so that tiles x slices stays within one
    wave of the 256 CUs) — against the oracle, default settings."""
    from dynamics_aware_diffusion_amd.utils import synth
    diff = build(net, cases.NETS[net][4], "cosine", dev)
    assert diff._engine(dev).small_batch_plan(B) == (0, 0)
    x = torch.from_numpy(synth.normal_like(83, f"mid.{net}.{B}", (B, 32, diff.transition_dim)))
    want = _oracle_eps(net, x, 640).numpy()
    got = diff.model(x.to(dev), 640).cpu().numpy()
    assert max_abs(got, want) <= TOL_STEP


def test_eight_environments_cost_less_per_plan_than_four(dev):
    """on HalfCheetah a denoise step for 8 plans takes at most 1.5x the step for 4
    (measured 720 vs 625 us), i.e. `get_actions(8 envs)` is cheaper per plan than two calls of 4."""
    import time
    from dynamics_aware_diffusion_amd import GuidedPolicy
    diff = build("halfcheetah", 1000, "cosine", dev)
    diff.sampler_rng, diff.seed, diff.use_graph = "philox", 5, True
    keep = diff.n_timesteps
    diff.n_timesteps = 60
    pol = GuidedPolicy(diff, None)
    cond = {0: torch.from_numpy(cases.loop_condition("mid", "halfcheetah")).to(dev)}
    us = {}
    try:
        for B in (4, 8):
            pol.sample_loop(batch_size=B, conditions=cond)             # capture + warm
            best = float("inf")
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pol.sample_loop(batch_size=B, conditions=cond)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e6 / diff.n_timesteps)
            us[B] = best
    finally:
        diff.n_timesteps = keep
        diff.use_graph = False
        diff.sampler_rng = "torch"
    print(f"HalfCheetah denoise step: B=4 {us[4]:.0f} us, B=8 {us[8]:.0f} us")
    assert us[8] <= 1.5 * us[4]


@pytest.mark.parametrize("net,B", [("halfcheetah", 1), ("halfcheetah", 2), ("halfcheetah", 5), ("halfcheetah", 16),
                                   ("door", 1), ("door", 3), ("door", 4), ("door", 16)])
def test_wide_nets_small_batches_take_the_streamed_weight_kernels(net, B, dev):
    """The `get_action` batch on the 1024 / 2048-channel nets: csrc/conv_ccw.hpp (weights streamed
    global -> registers, K slices of whole 128 / 256-channel groups, 16- and 32-row tiles, stand-alone
    1x1 residual convs consumed in pieces).  The plan must really be the consumer-combine one, one
    forward and a short conditioned loop must match the oracle, and the batch-256 kernels must
    agree with it to rounding."""
    from dynamics_aware_diffusion_amd.utils import synth
    diff = build(net, cases.NETS[net][4], "cosine", dev)
    eng = diff._engine(dev)
    assert eng.small_batch_plan(4)[1] >= 10 and eng.small_batch_plan(5) == (0, 0)     # default: up to 128 rows
    eng.debug_set_option("ccw_max_rows", 512)           # (beyond: correct, just slower than the batch-256 kernels)
    outs = []
    try:
        launches, wide = eng.small_batch_plan(B)
        assert launches > 0 and wide >= 10, (launches, wide)
        t = 123
        x = torch.from_numpy(synth.normal_like(69, f"widecc.{net}.{B}", (B, 32, diff.transition_dim)))
        want = _oracle_eps(net, x, t).numpy()
        got = diff.model(x.to(dev), t).cpu().numpy()
        assert max_abs(got, want) <= TOL_STEP
        T = 3
        noise = torch.from_numpy(synth.normal_like(70, f"widecc.noise.{net}.{B}", (T + 1, B, 32, diff.transition_dim)))
        cond = torch.from_numpy(synth.uniform(70, f"widecc.cond.{net}", (1, diff.transition_dim), 0.9))
        for cc in (1, 0):
            eng.debug_set_option("cc", cc)
            xl = noise[0].to(dev).clone()
            xl[:, 0] = cond.to(dev)
            eng.sample_loop(xl, T, noise_stack=noise[1:].to(dev).contiguous(), cond0=cond.to(dev))
            torch.cuda.synchronize()
            outs.append(xl.cpu().numpy())
        assert eng.small_batch_plan(B)[0] == 0
    finally:
        eng.debug_set_option("cc", int(diff.model.small_batch_kernels))
        eng.debug_set_option("ccw_max_rows", 128)
    assert max_abs(outs[0], outs[1]) <= TOL_LOOP
    if B <= 2:                                         # (the CPU oracle takes seconds per step here)
        want_loop = orc.sample_loop(net_weights_torch(net), orc.schedule_buffers("cosine", cases.NETS[net][4]), noise, T, {0: cond})
        assert max_abs(outs[0], want_loop.numpy()) <= TOL_LOOP


@pytest.mark.parametrize("B", [1, 2, 3, 7, 17, 65])
def test_ragged_batches_match_oracle(B, dev):
    from dynamics_aware_diffusion_amd.utils import synth
    diff = build("tiny4", 20, "cosine", dev)          # 4 levels: L = 32,16,8,4
    x = torch.from_numpy(synth.normal_like(62, f"ragged.{B}", (B, 32, diff.transition_dim)))
    got = diff.model(x.to(dev), 11)
    torch.cuda.synchronize()
    assert max_abs(got.cpu().numpy(), _oracle_eps("tiny4", x, 11).numpy()) <= TOL_STEP


def test_per_row_conditions_and_outputs(dev):
    """(B, td) conditions (one per environment) + eps/mean side outputs of one step."""
    from dynamics_aware_diffusion_amd.utils import synth
    net, T, B = "tiny", 20, 6
    diff = build(net, T, "cosine", dev)
    eng = diff._engine(dev)
    w, sched = net_weights_torch(net), orc.schedule_buffers("cosine", T)
    x = torch.from_numpy(synth.normal_like(63, "rowcond.x", (B, 32, 6)))
    z = torch.from_numpy(synth.normal_like(63, "rowcond.z", (B, 32, 6)))
    cond = torch.from_numpy(synth.uniform(63, "rowcond.c", (B, 6), 0.9))
    t = 7
    tt = torch.full((B,), t, dtype=torch.long)
    with torch.no_grad():
        mean, _, eps = orc.p_mean_variance(w, sched, x, tt)
        want = orc.denoise_step(w, sched, x.clone(), tt, z, {0: cond})
    xd = x.to(dev).clone()
    mean_d, eps_d = torch.empty_like(xd), torch.empty_like(xd)
    eng.denoise_step(xd, t, noise=z.to(dev), cond0=cond.to(dev), mean_out=mean_d, eps_out=eps_d)
    torch.cuda.synchronize()
    tol = TOL_STEP * eps_gain(diff, t)
    assert max_abs(eps_d.cpu().numpy(), eps.numpy()) <= TOL_STEP
    assert max_abs(mean_d.cpu().numpy(), mean.numpy()) <= tol
    assert max_abs(xd.cpu().numpy(), want.numpy()) <= tol
    assert torch.equal(xd[:, 0].cpu(), cond)
    # t == 0: noise is masked, result is the mean (+ inpainting)
    x0 = x.to(dev).clone()
    eng.denoise_step(x0, 0, noise=torch.full_like(x0, 1e6))
    assert torch.isfinite(x0).all() and x0.abs().max() < 50


def test_inkernel_philox_matches_its_oracle(dev):
    diff = build("tiny", 20, "cosine", dev)
    eng = diff._engine(dev)
    B, E = 37, 32 * 6
    for draw, seed, off in ((0, 1, 0), (5, 0xDEADBEEF12345678, 11)):
        x = torch.empty(B, 32, 6, device=dev)
        eng.fill_normal(x, seed, row_offset=off, draw=draw)
        idx = np.arange(B * E, dtype=np.uint64) + np.uint64(off * E)
        want = ophilox.normal(idx, draw, seed).reshape(B, 32, 6)
        assert max_abs(x.cpu().numpy(), want) <= 2e-5        # float stage: libm vs device
    big = torch.empty(4096, 32, 6, device=dev)
    eng.fill_normal(big, 7)
    v = big.cpu().numpy()
    assert abs(v.mean()) < 5e-3 and abs(v.std() - 1) < 5e-3 and np.abs(v).max() < 7


def test_philox_sampling_is_sharding_invariant_and_deterministic(dev):
    """Rows depend on (seed, global row) only: an 8-row call equals two 4-row calls at row
    offsets 0 and 4, and a repeat is bit-identical."""
    from dynamics_aware_diffusion_amd import GuidedPolicy
    diff = build("tiny", 20, "cosine", dev)
    diff.sampler_rng, diff.seed = "philox", 4242
    eng = diff._engine(dev)
    try:
        eng.debug_set_tile(101)                        # tile 1, no grid split-K => same summation order
        pol = GuidedPolicy(diff, None)
        cond = {0: torch.from_numpy(cases.loop_condition("inv", "tiny")).to(dev)}
        full = pol.sample_loop(batch_size=8, conditions=cond)
        again = pol.sample_loop(batch_size=8, conditions=cond)
        lo = pol.sample_loop(batch_size=4, conditions=cond, row_offset=0)
        hi = pol.sample_loop(batch_size=4, conditions=cond, row_offset=4)
        torch.cuda.synchronize()
        assert torch.equal(full, again)
        assert torch.equal(full, torch.cat([lo, hi]))
        diff.seed = 4243
        other = pol.sample_loop(batch_size=8, conditions=cond)
        assert not torch.equal(full, other)
        # unconditional p_sample_loop takes the same route
        a = diff.p_sample_loop((4, 32, 6), row_offset=4)
        diff.seed = 4242
        b = diff.p_sample_loop((8, 32, 6))
        c = diff.p_sample_loop((4, 32, 6), row_offset=4)
        assert torch.equal(b[4:], c) and not torch.equal(a, c)
    finally:
        eng.debug_set_tile(-1)
        diff.sampler_rng = "torch"


def test_projection_inside_the_loop_opt_in(dev):
    """README semantics x_{i-1} = project(denoise(x_i)): HIP loop vs the oracle loop with the
    projection as post-step; and the shipped default (no projection) stays untouched."""
    from dynamics_aware_diffusion_amd import DynamicsAwarePolicy
    from dynamics_aware_diffusion_amd.dynamics import ProjectionMatrixBuilder
    from tests.test_hip_parity import injected_noise
    net, T, B = "tiny", 20, 5
    diff = build(net, T, "cosine", dev)
    A, Bm = oproj.double_integrator(0.1)
    P = ProjectionMatrixBuilder(A, Bm, 4, 2).get_projection_matrix(32)
    norm = cases.NormalizerStub(4, 2)
    noise = cases.loop_noise("projloop", net, T, B)
    cond = torch.from_numpy(cases.loop_condition("projloop", net))
    stats = [torch.from_numpy(v) for v in (norm.obs_mean, norm.obs_std, norm.action_mean, norm.action_std)]
    sched = orc.schedule_buffers("cosine", T)

    def post(x, i):
        a = oproj.projection_alpha("noise_schedule", 1.0, i, T, sched["betas"])
        return oproj.apply_projection(x, P, a, 4, 4, *stats)

    want_proj = orc.sample_loop(net_weights_torch(net), sched, torch.from_numpy(noise), T, {0: cond},
                                post_step=post)
    want_plain = orc.sample_loop(net_weights_torch(net), sched, torch.from_numpy(noise), T, {0: cond})
    kw = dict(projection_matrix=P, normalizer=norm, state_dim=4, observation_dim=4, action_dim=2,
              horizon=32, projection_schedule="noise_schedule", projection_strength=1.0)
    for opt_in, want in ((True, want_proj), (False, want_plain)):
        pol = DynamicsAwarePolicy(diff, project_during_sampling=opt_in, **kw)
        with injected_noise(noise, dev):
            got = pol.sample_loop(batch_size=B, conditions={0: cond.to(dev)})
        torch.cuda.synchronize()
        assert max_abs(got.cpu().numpy(), want.numpy()) <= TOL_LOOP, opt_in
    assert max_abs(want_proj.numpy(), want_plain.numpy()) > 1e-3      # the projection matters


FULL_SIZE = [
    # (id, net, T, batch, projected)          BASELINE.json configs 2, 3, 4 and 5 (per-GPU shard)
    ("cfg2_pointmaze_T100_b256", "pointmaze", 100, 256, False),
    ("cfg3_pointmaze_T500_b256_proj", "pointmaze", 500, 256, True),
    ("cfg4_halfcheetah_T1000_b128", "halfcheetah", 1000, 128, False),
    ("cfg5_door_T1000_b128", "door", 1000, 128, False),
]


@pytest.mark.parametrize("cfg", FULL_SIZE, ids=lambda c: c[0])
def test_full_size_properties(cfg, dev):
    """Every BASELINE configuration at its full size, through size-independent checks: rows are
    independent of their batch (rows 0..3 of the full run == a 4-row run), the inpainted step 0 is
    exact and x0-clamping keeps plans in [-1, 1] (unprojected loops), results are deterministic,
    distinct rows are distinct plans."""
    from dynamics_aware_diffusion_amd import GuidedPolicy
    from tests.test_hip_parity import _double_integrator_policy
    name, net, T, B, projected = cfg
    diff = build(net, T, "cosine", dev)
    diff.sampler_rng, diff.seed = "philox", 99
    td = diff.transition_dim
    try:
        if projected:
            pol = _double_integrator_policy(diff, "noise_schedule", 1.0, dev, project_during_sampling=True)
        else:
            pol = GuidedPolicy(diff, None)
        c = cases.loop_condition("full", net)
        cond = {0: torch.from_numpy(c).to(dev)}
        big = pol.sample_loop(batch_size=B, conditions=cond)
        big2 = pol.sample_loop(batch_size=B, conditions=cond)
        small = pol.sample_loop(batch_size=4, conditions=cond)
        torch.cuda.synchronize()
        assert torch.equal(big, big2)
        assert torch.isfinite(big).all()
        if not projected:
            assert np.array_equal(big[:, 0].cpu().numpy(), np.broadcast_to(c, (B, td)))
            assert float(big.abs().max()) <= 1.0 + 1e-3      # last step: sigma = 0, |x0| <= 1
        else:
            # v' = a P v + (1-a) v shrinks the off-subspace part by (1-a) per step: after the last
            # step (a = sqrt(1 - beta_0) ~ 0.9997) the plans obey x_{t+1} = A x_t + B u_t
            norm = pol.normalizer
            xs = big[:, :, :4].double().cpu() * torch.from_numpy(norm.obs_std).double() + torch.from_numpy(norm.obs_mean).double()
            us = big[:, :, 4:].double().cpu() * torch.from_numpy(norm.action_std).double() + torch.from_numpy(norm.action_mean).double()
            A, Bm = oproj.double_integrator(0.1)
            A, Bm = torch.from_numpy(np.asarray(A)).double(), torch.from_numpy(np.asarray(Bm)).double()
            resid = xs[:, 1:] - (xs[:, :-1] @ A.T + us[:, :-1] @ Bm.T)
            assert float(resid.abs().max()) <= 5e-3 * max(1.0, float(xs.abs().max()))
        # batch 4 and the full batch may pick different tiles (summation order): fp32 tolerance
        assert max_abs(big[:4].cpu().numpy(), small.cpu().numpy()) <= TOL_LOOP
        # distinct rows are distinct plans
        assert float((big[1:] - big[:-1]).abs().max()) > 1e-2
    finally:
        diff.sampler_rng = "torch"


def test_projection_kernel_beyond_one_wave_of_blocks_at_wide_dims(dev):
    """dad_project at B > 512 picks four rows per block only when they fit LDS: D = 753
    (HalfCheetah-sized n = 17, m = 6) must fall back to one row per block and still match."""
    from dynamics_aware_diffusion_amd._engine import ProjectionState
    from dynamics_aware_diffusion_amd.utils import synth
    n, m, H, B = 17, 6, 32, 520
    D = (H + 1) * n + H * m
    assert 17 * 4 * D * 4 > 160 * 1024 >= 17 * D * 4
    rng = np.random.default_rng(3)
    Q, _ = np.linalg.qr(rng.normal(size=(D, n + H * m)))
    P = torch.from_numpy((Q @ Q.T).astype(np.float32))
    stats = [torch.from_numpy(v) for v in (synth.normal_like(5, "wp.om", (n,)),
                                           1.0 + synth.uniform(5, "wp.os", (n,), 0.5),
                                           synth.normal_like(5, "wp.am", (m,)),
                                           1.0 + synth.uniform(5, "wp.as", (m,), 0.5))]
    x = torch.from_numpy(synth.normal_like(5, "wp.x", (B, H, n + m)))
    want = oproj.apply_projection(x, P, 0.75, n, n, *stats)
    for gemm in (False, True):             # one trajectory per block (P streamed per trajectory) / MFMA GEMM
        st = ProjectionState(P, *stats, n, n, m, dev)
        st.gemm = gemm
        xd = x.to(dev).clone()
        st.apply(xd, 0.75)
        torch.cuda.synchronize()
        assert max_abs(xd.cpu().numpy(), want.numpy()) <= 2e-5, gemm


@pytest.mark.parametrize("dims", [(17, 6, 128), (39, 28, 128), (39, 28, 45), (12, 4, 64)],
                         ids=lambda d: "n%d_m%d_B%d" % d)
def test_projection_as_a_gemm_at_wide_dims(dims, dev):
    """apply_projection (guides/policies.py:431-483) at HalfCheetah (D = 753) and Door (D = 2183) size for
    a batch of 128 — v @ P as an MFMA GEMM, P read once per 32 trajectories — against the oracle in
    float64; a ragged batch (45) and a mid size (D = 524); a second call reproduces the first bit for bit.
    Door size exists only in this form: one trajectory's partial sums (148 KB) leave no room in LDS."""
    from dynamics_aware_diffusion_amd._engine import ProjectionState
    from dynamics_aware_diffusion_amd.utils import synth
    n, m, B = dims
    H = 32
    D = (H + 1) * n + H * m
    rng = np.random.default_rng(11)
    Q, _ = np.linalg.qr(rng.normal(size=(D, n + H * m)))
    P64 = Q @ Q.T
    P = torch.from_numpy(P64.astype(np.float32))
    stats = [torch.from_numpy(v) for v in (synth.normal_like(6, "pg.om", (n,)),
                                           1.0 + synth.uniform(6, "pg.os", (n,), 0.5),
                                           synth.normal_like(6, "pg.am", (m,)),
                                           1.0 + synth.uniform(6, "pg.as", (m,), 0.5))]
    x = torch.from_numpy(synth.normal_like(6, f"pg.x.{D}", (B, H, n + m)))
    want = oproj.apply_projection(x.double(), P.double(), 0.6, n, n, *[s.double() for s in stats])
    st = ProjectionState(P, *stats, n, n, m, dev)
    outs = []
    for _ in range(2):
        xd = x.to(dev).clone()
        st.apply(xd, 0.6)
        torch.cuda.synchronize()
        outs.append(xd.cpu().numpy())
    err = max_abs(outs[0], want.numpy())
    print(f"projection GEMM D={D} B={B}: max |hip - fp64 oracle| = {err:.2e}")
    assert err <= 2e-5
    assert np.array_equal(outs[0], outs[1])


def test_engine_argument_errors(dev):
    from dynamics_aware_diffusion_amd._engine import DadError
    diff = build("tiny", 20, "cosine", dev)
    eng = diff._engine(dev)
    with pytest.raises(RuntimeError):
        eng.unet_forward(torch.zeros(2, 16, 6, device=dev), 0)            # wrong horizon
    with pytest.raises(RuntimeError):
        eng.unet_forward(torch.zeros(2, 32, 6, device=dev).double(), 0)   # wrong dtype
    with pytest.raises(DadError):
        eng.unet_forward(torch.zeros(2, 32, 6, device=dev), 20)           # t outside schedule
    with pytest.raises(RuntimeError):
        eng.denoise_step(torch.zeros(2, 32, 6, device=dev), 0, cond0=torch.zeros(3, 6, device=dev))


def test_weights_refresh_after_load_state_dict(dev):
    """The packed engine copy follows the module's parameters (checkpoint reload)."""
    from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
    from dynamics_aware_diffusion_amd.utils import synth
    unet = TemporalUnet(6, dim=32, dim_mults=(1, 2, 4))
    diff = GaussianDiffusion(unet, 32, 4, 2, n_timesteps=20).to(dev)
    x = torch.from_numpy(synth.normal_like(64, "refresh.x", (3, 32, 6)))
    for seed in (3, 8):
        state = synth.synth_unet_state(6, 32, (1, 2, 4), seed=seed, affine_jitter=0.25)
        w = {k: torch.from_numpy(v) for k, v in state.items()}
        sd = diff.state_dict()
        sd.update({"model." + k: v for k, v in w.items()})
        diff.load_state_dict(sd)                       # reference checkpoint format
        with torch.no_grad():
            want = orc.unet_forward(w, x, torch.full((3,), 4, dtype=torch.long))
        got = diff.model(x.to(dev), 4)
        assert max_abs(got.cpu().numpy(), want.numpy()) <= TOL_STEP, seed


def test_graph_replay_with_inkernel_noise(dev):
    """hipGraph replay of the whole loop with the Philox key read from device memory: identical
    to eager launches, and a new seed needs no new capture."""
    from dynamics_aware_diffusion_amd import GuidedPolicy
    diff = build("tiny", 20, "cosine", dev)
    diff.sampler_rng = "philox"
    pol = GuidedPolicy(diff, cases.NormalizerStub(4, 2), action_horizon=3)
    cond = {0: torch.from_numpy(cases.loop_condition("graphphilox", "tiny")).to(dev)}
    try:
        outs = {}
        for graph in (False, True):
            diff.use_graph = graph
            for seed in (11, 12, 11):
                diff.seed = seed
                x = pol.sample_loop(batch_size=3, conditions=cond)
                u = diff.p_sample_loop((2, 32, 6))
                torch.cuda.synchronize()
                outs.setdefault((graph, seed), []).append((x.cpu().numpy().copy(), u.cpu().numpy().copy()))
        for seed in (11, 12):
            e, g = outs[(False, seed)][0], outs[(True, seed)][0]
            assert np.array_equal(e[0], g[0]) and np.array_equal(e[1], g[1])
        assert np.array_equal(outs[(True, 11)][0][0], outs[(True, 11)][1][0])      # replayed, same seed
        assert not np.array_equal(outs[(True, 11)][0][0], outs[(True, 12)][0][0])  # new seed, same graph
        # planner glue on top of a replayed B=1 plan
        diff.use_graph = True
        a = pol.get_action(np.zeros(4, np.float32))
        assert a.shape == (2,) and np.isfinite(a).all() and len(pol.action_buffer) == 3
    finally:
        diff.use_graph = False
        diff.sampler_rng = "torch"


@pytest.mark.parametrize("arch", [
    # (td, dim, mults, horizon, B)
    (6, 32, (1, 2), 16, 3),            # 2 levels, short horizon
    (11, 64, (1, 2, 4), 16, 5),        # L = 16, 8, 4
    (23, 64, (1, 1, 2), 64, 2),        # horizon 64: one sample per 64-row tile; repeated width
    (3, 32, (1,), 32, 4),              # single level: no down/up-sampling, no skip is popped
    (8, 128, (1, 4), 8, 9),            # horizon 8 -> L = 8, 4; wide jump 128 -> 512
    (6, 32, (1, 4, 2), 32, 3),         # shrinking widths: decoder block with an IDENTITY residual
                                       # over the channel concat (2*64 == 128)
    (5, 32, (1, 4, 2, 1), 32, 2),      # two such stages
    (32, 32, (1, 2), 32, 3),           # transition_dim == dim: the first block's residual is the
                                       # trajectory itself (nn.Identity, temporal_unet.py:92-94)
    (9, 256, (1, 8), 32, 2),           # 2048 channels at 16 positions: GroupNorm pairs of 4096 elements
                                       # (conv_ccw re-reads LDS between the passes at small batch)
    (6, 128, (1, 2, 4), 128, 1),       # horizon 128 (the reference's QUICKSTART.md:82 recipe): 128-position
    (6, 128, (1, 2, 4), 128, 9),       # tiles <32,128> on level 0, batch 1 and 9
    (23, 256, (1, 4, 8), 128, 3),      # the same on the HalfCheetah widths (64-channel groups at L = 64: <128,64>)
    (14, 64, (1, 2, 4, 8), 64, 5),     # horizon 64 on four levels
    (6, 128, (1, 2, 4), 32, 9, 3),     # TemporalUnet(kernel_size=3) (temporal_unet.py:139) on the PointMaze widths: split-K batch
    (6, 128, (1, 2, 4), 32, 130, 3),   #   ... and a batch with the 1x1 residual conv riding as the fourth tap
    (6, 64, (1, 2, 4), 32, 70, 7),     # kernel_size=7: three halo rows per sample side, ride as the eighth tap
    (11, 32, (1, 2), 16, 1, 7),        #   ... batch 1 (the small-batch kernels are 5-tap only: batch kernels + split-K)
    (23, 256, (1, 4, 8), 32, 4, 3),    # kernel_size=3 on the HalfCheetah widths (LDS-staged <256,32> tile at 2048 channels)
    (5, 32, (1, 4, 2), 32, 3, 7),      # kernel_size=7 with an identity residual over the concat
    (6, 48, (1, 2), 32, 5),            # --dim 48: GroupNorm groups of 6 / 12 channels (temporal_unet.py:71 needs only
    (9, 96, (1, 2, 4), 16, 70),        #   C % 8 == 0) run zero-padded to 8 / 16 (utils/padding.py,
    (6, 24, (1, 2, 4), 32, 130),       #   dad_model_set_group_channels): masked GroupNorm statistics in the epilogue;
    (5, 40, (1, 2, 3), 32, 2),         #   40 / 80 / 120 -> 64 / 128 / 128: two levels share a padded width
    (7, 8, (1, 2), 16, 1),             #   one-channel groups (dim 8) at batch 1: split-K + padding
    (6, 48, (1, 2), 32, 9, 3),         #   ... with kernel_size 3
    (6, 128, (1, 2, 4), 24, 40),       # horizon 24 on the PointMaze widths (24 / 12 / 6 positions, zero-padded to 32 / 16 / 8)
    (11, 64, (1, 2, 4), 100, 3),       # horizon 100: 100 / 50 / 25 positions in 128-position tiles
    (23, 256, (1, 4, 8), 48, 2),       # horizon 48 on the HalfCheetah widths (wide-group tiles, split-K)
    (6, 48, (1, 2), 48, 5, 3),         # zero-padded horizon AND widths AND kernel_size 3
    (9, 32, (1, 2, 2, 4), 8, 70),      # horizon 8 on four levels: ONE position at the deepest level (padded to 32 / 16 / 8 / 4)
    (6, 32, (1, 2, 4, 8), 16, 4),      # the reference's default dim_mults at train.py's default horizon 16: 16 / 8 / 4 / 2
], ids=lambda a: f"td{a[0]}_d{a[1]}_m{'x'.join(map(str, a[2]))}_H{a[3]}_B{a[4]}" + (f"_k{a[5]}" if len(a) > 5 else ""))
def test_assorted_architectures_match_oracle(arch, dev):
    """Shapes outside the three BASELINE architectures, against the oracle on seeded inputs."""
    from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
    from dynamics_aware_diffusion_amd.utils import synth
    td, dim, mults, H, B = arch[:5]
    ks = arch[5] if len(arch) > 5 else 5
    state = synth.synth_unet_state(td, dim, mults, seed=17, affine_jitter=0.3, kernel_size=ks)
    w = {k: torch.from_numpy(v) for k, v in state.items()}
    unet = TemporalUnet(td, dim=dim, dim_mults=mults, kernel_size=ks)
    unet.load_state_dict(w)
    diff = GaussianDiffusion(unet, H, td - 1, 1, n_timesteps=30).to(dev)
    x = torch.from_numpy(synth.normal_like(66, f"arch.{arch}", (B, H, td)))
    with torch.no_grad():
        want = orc.unet_forward(w, x, torch.full((B,), 21, dtype=torch.long))
    got = diff.model(x.to(dev), 21)
    torch.cuda.synchronize()
    assert max_abs(got.cpu().numpy(), want.numpy()) <= TOL_STEP
    # and a short conditioned loop with injected noise
    T = 30
    noise = torch.from_numpy(synth.normal_like(66, f"arch.noise.{arch}", (T + 1, B, H, td)))
    cond = torch.from_numpy(synth.uniform(66, f"arch.cond.{arch}", (1, td), 0.9))
    want_loop = orc.sample_loop(w, orc.schedule_buffers("cosine", T), noise, T, {0: cond})
    eng = diff._engine(dev)
    xl = noise[0].to(dev).clone()
    xl[:, 0] = cond.to(dev)
    eng.sample_loop(xl, T, noise_stack=noise[1:].to(dev).contiguous(), cond0=cond.to(dev))
    torch.cuda.synchronize()
    err = max_abs(xl.cpu().numpy(), want_loop.numpy())
    if err > TOL_LOOP:
        # long horizons: the first reverse step amplifies an eps error ~100x (eps_gain) and the maximum is
        # taken over 4x the elements; fall back to the fp64 criterion of the forward goldens — the HIP loop
        # is no farther from the float64 loop than twice the fp32 oracle is
        sched64 = {k: v.double() for k, v in orc.schedule_buffers("cosine", T).items()}
        truth = orc.sample_loop(orc.cast_weights(w, torch.float64), sched64, noise.double(), T, {0: cond.double()})
        e_hip = max_abs(xl.cpu().double().numpy(), truth.numpy())
        e_ref = max_abs(want_loop.double().numpy(), truth.numpy())
        print(f"{arch}: loop vs fp32 oracle {err:.2e}; vs fp64: hip {e_hip:.2e}, fp32 oracle {e_ref:.2e}")
        assert e_hip <= 2 * e_ref + 5e-7
    else:
        assert err <= TOL_LOOP


@pytest.mark.parametrize("arch", [(6, 32, (1, 2, 4), 70), (9, 64, (1, 2), 64), (6, 128, (1, 2, 4), 100), (16, 128, (1, 4), 65)],
                         ids=lambda a: f"td{a[0]}_d{a[1]}_m{'x'.join(map(str, a[2]))}_B{a[3]}")
def test_level0_chain_matches_oracle_and_the_separate_launches(arch, dev):
    """csrc/conv_chain.hpp: downs.0.0 (+ riding 1x1 residual conv), downs.0.1 and the down-sampling conv as one
    launch per sample (dim 32 / 64 / 128, batches of 64+): against the oracle, and against the same net with the
    chain switched off (five separate launches) to fp32 rounding."""
    from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
    from dynamics_aware_diffusion_amd.utils import synth
    td, dim, mults, B = arch
    state = synth.synth_unet_state(td, dim, mults, seed=23, affine_jitter=0.3)
    w = {k: torch.from_numpy(v) for k, v in state.items()}
    unet = TemporalUnet(td, dim=dim, dim_mults=mults)
    unet.load_state_dict(w)
    diff = GaussianDiffusion(unet, 32, td - 1, 1, n_timesteps=30).to(dev)
    x = torch.from_numpy(synth.normal_like(67, f"chain.{arch}", (B, 32, td)))
    with torch.no_grad():
        want = orc.unet_forward(w, x, torch.full((B,), 17, dtype=torch.long))
    eng = diff._engine(dev)
    try:
        eng.debug_set_option("chain", 1)
        got = diff.model(x.to(dev), 17).cpu().numpy()
        again = diff.model(x.to(dev), 17).cpu().numpy()
        eng.debug_set_option("chain", 0)
        sep = diff.model(x.to(dev), 17).cpu().numpy()
    finally:
        eng.debug_set_option("chain", 0)               # (the default: opt-in)
    assert np.array_equal(got, again)
    assert max_abs(got, want.numpy()) <= TOL_STEP
    assert max_abs(sep, want.numpy()) <= TOL_STEP
    assert not np.array_equal(got, sep) or dim == 0        # (two different summation orders: the switch really switches)
    assert max_abs(got, sep) <= 1e-5


def test_per_row_timesteps_near_the_end_of_the_schedule_on_a_wide_net(dev):
    """`dad_unet_forward_rows` indexes the per-timestep table with t_row[b] only where a conv HAS a
    time embedding (the second conv of a block, the resampling and 1x1 convs fall back to the bias row and must
    not be offset by t * temb_width): rows at t = T-1 on HalfCheetah's widths (temb_width 10 752, T = 1000), a
    NaN-poisoned neighbourhood would show as NaN."""
    from dynamics_aware_diffusion_amd.utils import synth
    diff = build("halfcheetah", 1000, "cosine", dev)
    B = 5
    x = torch.from_numpy(synth.normal_like(84, "rows.hc", (B, 32, diff.transition_dim)))
    t = torch.tensor([999, 0, 998, 500, 999], dtype=torch.long)
    with torch.no_grad():
        want = orc.unet_forward(net_weights_torch("halfcheetah"), x, t).numpy()
        got = diff.model(x.to(dev), t.to(dev)).cpu().numpy()
    assert np.isfinite(got).all()
    assert max_abs(got, want) <= TOL_STEP


def test_projector_refuses_a_batch_of_another_shape(dev):
    """the projection kernels derive D from the batch's horizon; a batch whose horizon or
    transition width differs from what P was built for must raise (the reference: matmul shape error,
    guides/policies.py:451, losses/__init__.py:181), not read P out of bounds."""
    from dynamics_aware_diffusion_amd._engine import ProjectionState
    from dynamics_aware_diffusion_amd.dynamics import ProjectionMatrixBuilder, double_integrator
    import contextlib
    import io
    A, Bm = double_integrator(0.1)
    with contextlib.redirect_stdout(io.StringIO()):
        P = ProjectionMatrixBuilder(A, Bm, 4, 2).get_projection_matrix(32)
    n = cases.NormalizerStub(4, 2)
    st = ProjectionState(P, n.obs_mean, n.obs_std, n.action_mean, n.action_std, 4, 4, 2, dev)
    st.apply(torch.zeros(3, 32, 6, device=dev), 0.5)                       # the shape it was built for
    for shape in ((3, 16, 6), (3, 32, 7), (3, 64, 6)):
        with pytest.raises(RuntimeError):
            st.apply(torch.zeros(*shape, device=dev), 0.5)
        with pytest.raises(RuntimeError):
            st.violation(torch.zeros(*shape, device=dev))
    with pytest.raises(RuntimeError):
        st.apply(torch.zeros(3, 32, 6), 0.5)                               # CPU tensor


def test_unsupported_architectures_are_refused_with_a_message(dev):
    from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
    from dynamics_aware_diffusion_amd._engine import DadError
    for kwargs, H in ((dict(dim=32, dim_mults=(1, 2, 4, 8)), 12),     # 12 cannot be halved three times (nor can the reference)
                      (dict(dim=44, dim_mults=(1, 2)), 32),           # GroupNorm(8, 44) does not exist in the reference either
                      (dict(dim=32, dim_mults=(1, 2), kernel_size=4), 32),   # even kernel: the reference's padding k//2 changes the length
                      (dict(dim=32, dim_mults=(1, 2), kernel_size=9), 32)):
        unet = TemporalUnet(6, **kwargs)
        diff = GaussianDiffusion(unet, H, 4, 2, n_timesteps=10).to(dev)
        with pytest.raises((DadError, ValueError)):
            diff.model(torch.zeros(1, H, 6, device=dev), 0)


def test_batched_get_actions_matches_per_env_oracle(dev):
    """N environments in one loop with per-row conditions: plans equal the oracle's loop with the
    same (N, td) condition, and the FIFO semantics equal N single-environment policies."""
    from dynamics_aware_diffusion_amd import GuidedPolicy
    from dynamics_aware_diffusion_amd.utils import synth
    from tests.test_hip_parity import injected_noise
    net, T, N, ah = "tiny", 20, 5, 2
    diff = build(net, T, "cosine", dev)
    norm = cases.NormalizerStub(4, 2)
    obs = synth.normal_like(67, "batched.obs", (N, 4))
    noise = synth.normal_like(67, "batched.noise", (T + 1, N, 32, 6))
    pol = GuidedPolicy(diff, norm, action_horizon=ah)
    with injected_noise(noise, dev):
        first = pol.get_actions(obs)
    served = [first] + [pol.get_actions(obs) for _ in range(ah)]          # from the queue
    cond = np.zeros((N, 6), np.float32)
    cond[:, :4] = norm.normalize_observations(obs)
    want = orc.sample_loop(net_weights_torch(net), orc.schedule_buffers("cosine", T),
                           torch.from_numpy(noise), T, {0: torch.from_numpy(cond)}).numpy()
    for k, got in enumerate(served):
        ref = want[:, k, 4:6] * norm.action_std + norm.action_mean
        assert got.shape == (N, 2)
        assert max_abs(got, ref) <= TOL_LOOP, k
    assert pol._batched_cursor == ah + 1                                  # queue exhausted: replan next
    with pytest.raises(ValueError):
        pol.get_actions(np.zeros((N, 3), np.float32))


def test_projector_and_system_id_on_the_device(dev):
    """P built on the GPU equals the host pinv path at PointMaze size and is a
    rank-(n + H m) orthogonal projector at Door size (D = 2183); the device least-squares fit
    equals the host fit."""
    from dynamics_aware_diffusion_amd.dynamics import (ProjectionMatrixBuilder, double_integrator,
                                                       fit_linear_dynamics)
    A, B = double_integrator(0.1)
    b = ProjectionMatrixBuilder(A, B, 4, 2)
    Pd = b.get_projection_matrix(32, device=dev)
    assert Pd.device.type == "cuda" and Pd.dtype == torch.float32
    assert max_abs(Pd.cpu().numpy(), b.get_projection_matrix(32).numpy()) <= 1e-6
    rng = np.random.default_rng(9)
    n, m, H = 39, 28, 32
    A = 0.95 * np.linalg.qr(rng.normal(size=(n, n)))[0]
    Bm = rng.normal(size=(n, m)) / np.sqrt(n)
    P = ProjectionMatrixBuilder(A, Bm, n, m).get_projection_matrix(H, device=dev).double()
    D = (H + 1) * n + H * m
    assert P.shape == (D, D)
    assert float((P - P.T).abs().max()) <= 1e-6
    assert float((P @ P - P).abs().max()) <= 1e-5
    assert abs(float(P.trace()) - (n + H * m)) <= 1e-2
    S = rng.normal(size=(5000, n)); U = rng.normal(size=(5000, m))
    S1 = S @ A.T + U @ Bm.T
    Ah, Bh = fit_linear_dynamics(S, U, S1)
    Ad, Bd = fit_linear_dynamics(S, U, S1, device=dev)
    assert np.abs(Ad - Ah).max() <= 1e-9 and np.abs(Bd - Bh).max() <= 1e-9 and np.abs(Ad - A).max() <= 1e-9


def test_large_batch_matches_oracle_rows(dev):
    """2048 plans in one call (plentiful-tile heuristics, 1024 N-tiles per layer): first and last
    rows against the oracle, in both conv arithmetics."""
    from dynamics_aware_diffusion_amd.utils import synth
    diff = build("pointmaze", 100, "cosine", dev)
    keep = diff.model.precision
    x = torch.from_numpy(synth.normal_like(3, "big.x", (2048, 32, 6)))
    rows = [0, 1, 2, 3, 2044, 2045, 2046, 2047]
    want = _oracle_eps("pointmaze", x[rows], 40)
    try:
        for prec in ("fp32", "f16x3"):
            diff.model.precision = prec
            y = diff.model(x.to(dev), torch.full((2048,), 40, device=dev, dtype=torch.long))
            torch.cuda.synchronize()
            assert bool(torch.isfinite(y).all())
            assert max_abs(y[rows].cpu().numpy(), want) <= TOL_STEP, prec
    finally:
        diff.model.precision = keep


def test_bench_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks before
    touching the GPU; they share this box's one GPU (rehearsal mode) and exchange plans and the
    max-over-ranks time over gloo.  Exercises init_process_group, the gather and the MAX reduce of
    bench.py itself."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DAD_BENCH_SHARE_GPU="1", DAD_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
                          "--warmup", "0", "--no-alt", "--no-cpu-baseline", "--no-configs"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 512
    assert out["value"] > 0 and out["roofline"]["frac"] > 0


def test_bench_rehearses_config5_shape_on_four_ranks(dev):
    """BASELINE config 5's workload (Door, T = 1000, 128 plans per rank) through bench.py's own rank
    start-up, gather and max-over-ranks timing on FOUR ranks sharing this box's GPU (a GPU box admits at
    most six GPU processes: the eight-rank run is the driver's; its sharding arithmetic is covered by the
    eight-rank gloo test of tests/test_host_logic.py and by the row-for-row test below)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DAD_BENCH_SHARE_GPU="1", DAD_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--workload", "door_b128",
                          "--steps", "1", "--warmup", "0", "--no-alt", "--no-cpu-baseline", "--no-configs"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["global_batch"] == 512 and out["config"]["batch_per_gpu"] == 128
    assert out["config"]["denoise_steps"] == 1000 and out["value"] > 0


def test_config5_shards_equal_one_1024_row_run(dev):
    """BASELINE config 5 row for row: eight shards of 128 plans (row offsets 0, 128, ..., 896 — what the
    eight ranks compute) equal ONE 1024-plan Philox run on the Door architecture, bit for bit (tiles pinned
    so that the summation order does not follow the per-GPU batch; loop truncated to 6 of the 1000 steps)."""
    from dynamics_aware_diffusion_amd import GuidedPolicy
    diff = build("door", 1000, "cosine", dev)
    diff.sampler_rng, diff.seed = "philox", 777
    keep = diff.n_timesteps
    diff.n_timesteps = 6
    eng = diff._engine(dev)
    try:
        eng.debug_set_tile(101)                        # tile 1 where valid, no grid split-K
        pol = GuidedPolicy(diff, None)
        cond = {0: torch.from_numpy(cases.loop_condition("cfg5", "door")).to(dev)}
        full = pol.sample_loop(batch_size=1024, conditions=cond)
        shards = [pol.sample_loop(batch_size=128, conditions=cond, row_offset=128 * r) for r in range(8)]
        torch.cuda.synchronize()
        assert bool(torch.isfinite(full).all())
        assert torch.equal(full, torch.cat(shards))
    finally:
        eng.debug_set_tile(-1)
        diff.n_timesteps = keep
        diff.sampler_rng = "torch"


@pytest.mark.parametrize("arch", [(23, 32, (1, 4, 8), 17, 6), (8, 32, (1, 2, 4, 8), 5, 3),
                                  (6, 48, (1, 2), 4, 2, 3)],        # --dim 48 (zero-padded groups) with kernel_size 3
                         ids=["mults_1_4_8", "mults_1_2_4_8", "dim48_k3"])
def test_checkpoint_round_trip_on_the_device(arch, dev, tmp_path):
    """load_checkpoint(path) -> sampler on the GPU: the loaded net reproduces the oracle on the
    checkpoint's weights (raw and EMA), with widths inferred from shapes."""
    from dynamics_aware_diffusion_amd import load_checkpoint
    from dynamics_aware_diffusion_amd.utils import synth
    from tests.test_host_logic import _synthetic_checkpoint
    td, dim, mults, od, ad = arch[:5]
    ks = arch[5] if len(arch) > 5 else 5
    ckpt, w, w_ema = _synthetic_checkpoint(td, dim, mults, od, ad, ema_seed=12, kernel_size=ks)
    path = tmp_path / "model.pt"
    torch.save(ckpt, path)
    x = torch.from_numpy(synth.normal_like(69, f"ckpt.{mults}", (3, 32, td)))
    t = torch.full((3,), 11, dtype=torch.long)
    outs = []
    for use_ema, weights in ((False, w), (True, w_ema)):
        diff = load_checkpoint(path, device=dev, use_ema=use_ema)
        assert diff.model.dim_mults == mults and diff.model.kernel_size == ks and diff.betas.device.type == "cuda"
        with torch.no_grad():
            want = orc.unet_forward({k: torch.from_numpy(v) for k, v in weights.items()}, x, t)
        got = diff.model(x.to(dev), 11)
        torch.cuda.synchronize()
        assert max_abs(got.cpu().numpy(), want.numpy()) <= TOL_STEP, use_ema
        outs.append(got.cpu())
        plans = diff.p_sample_loop((2, 32, td))                     # and it samples
        assert torch.isfinite(plans).all()
    assert float((outs[0] - outs[1]).abs().max()) > 1e-3            # EMA weights are different weights


def _random_architectures(seed: int, count: int):
    """The generator of tests/fuzz_parity.py, frozen: (td, dim, mults, H, B, t) draws that the
    library supports (first `count` accepted ones)."""
    import random
    rng = random.Random(seed)
    out = []
    while len(out) < count:
        dim = rng.choice([32, 64, 128])
        nlev = rng.choice([1, 2, 3, 4])
        mults = tuple([1] + [rng.choice([1, 2, 4, 8]) for _ in range(nlev - 1)])
        H = rng.choice([8, 16, 32, 32, 64])
        td = rng.randint(2, 24)
        B = rng.choice([1, 2, 3, 5, 8, 13, 16])
        t = rng.randint(0, 19)
        if H >> (nlev - 1) < 4 or max(mults) * dim > 1024:
            continue
        out.append((td, dim, mults, H, B, t))
    return out


@pytest.mark.parametrize("arch", _random_architectures(2026, 14),
                         ids=lambda a: f"td{a[0]}_d{a[1]}_m{'x'.join(map(str, a[2]))}_H{a[3]}_B{a[4]}")
def test_random_architectures_both_small_batch_families(arch, dev):
    """Frozen draws of the fuzz generator (widths up to 1024, horizons 8..64, batches 1..31):
    one forward and a short conditioned loop against the oracle, through the consumer-combine
    kernels where the batch is small enough AND through the batch-256 kernels."""
    from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet
    from dynamics_aware_diffusion_amd.utils import synth
    td, dim, mults, H, B, t = arch
    state = synth.synth_unet_state(td, dim, mults, seed=500 + td, affine_jitter=0.3)
    w = {k: torch.from_numpy(v) for k, v in state.items()}
    x = torch.from_numpy(synth.normal_like(501, f"rnd.{arch}", (B, H, td)))
    with torch.no_grad():
        want = orc.unet_forward(w, x, torch.full((B,), t, dtype=torch.long))
    T = 6
    noise = torch.from_numpy(synth.normal_like(502, f"rnd.noise.{arch}", (T + 1, B, H, td)))
    cond = torch.from_numpy(synth.uniform(502, f"rnd.cond.{arch}", (1, td), 0.9))
    want_loop = orc.sample_loop(w, orc.schedule_buffers("cosine", 20), noise, T, {0: cond})
    for small in (True, False):
        unet = TemporalUnet(td, dim=dim, dim_mults=mults)
        unet.load_state_dict(w)
        unet.small_batch_kernels = small
        diff = GaussianDiffusion(unet, H, td - 1, 1, n_timesteps=20).to(dev)
        got = diff.model(x.to(dev), t)
        eng = diff._engine(dev)
        xl = noise[0].to(dev).clone()
        xl[:, 0] = cond.to(dev)
        eng.sample_loop(xl, T, noise_stack=noise[1:].to(dev).contiguous(), cond0=cond.to(dev))
        torch.cuda.synchronize()
        assert max_abs(got.cpu().numpy(), want.numpy()) <= TOL_STEP, small
        assert max_abs(xl.cpu().numpy(), want_loop.numpy()) <= TOL_LOOP, small


def test_small_batch_projected_loop_graph_equals_eager(dev):
    """Consumer-combine kernels + in-loop projection + hipGraph replay at batch 3: the replayed
    graph (captured once, replayed with a new seed) equals eager launches bit for bit."""
    from tests.test_hip_parity import _double_integrator_policy
    diff = build("tiny", 20, "cosine", dev)
    diff.sampler_rng = "philox"
    cond = {0: torch.from_numpy(cases.loop_condition("smallproj", "tiny")).to(dev)}
    try:
        outs = {}
        for graph in (False, True, True):
            diff.use_graph = graph
            pol = _double_integrator_policy(diff, "linear", 0.7, dev, project_during_sampling=True)
            for seed in (5, 6):
                diff.seed = seed
                x = pol.sample_loop(batch_size=3, conditions=cond)
                torch.cuda.synchronize()
                outs.setdefault((graph, seed), []).append(x.cpu().numpy().copy())
        for seed in (5, 6):
            assert np.array_equal(outs[(False, seed)][0], outs[(True, seed)][0])
            assert np.array_equal(outs[(True, seed)][0], outs[(True, seed)][1])
        assert not np.array_equal(outs[(True, 5)][0], outs[(True, 6)][0])
    finally:
        diff.use_graph = False
        diff.sampler_rng = "torch"


@pytest.mark.gpu
def test_ema_copy_runs_its_own_engine(dev):
    """``copy.deepcopy(diffusion)`` after the original has run (the reference's EMA model, utils/training.py:77-84): the
    copy builds its own engine, gives the original's output, follows its own EMA update — and the original is unchanged."""
    import copy
    diff = build("tiny", 20, "cosine", dev)
    x = torch.from_numpy(cases.forward_input("ema", "tiny", 3)).to(dev)
    with torch.no_grad():
        first = diff.model(x, 4).clone()
        ema = copy.deepcopy(diff)
        assert ema.model._engine is None
        assert torch.equal(ema.model(x, 4), first) and ema.model._engine is not diff.model._engine
        w = {k: v.clone() for k, v in net_weights_torch("tiny").items()}
        for k, p in ema.model.named_parameters():           # an in-place update, as EMA.update_model_average makes
            p.mul_(1.1)
            w[k] = 1.1 * w[k]
        got = ema.model(x, 4)
        want = orc.unet_forward(w, x.cpu(), torch.full((3,), 4, dtype=torch.long))
        assert max_abs(got.cpu().numpy(), want.numpy()) <= 2e-5
        assert torch.equal(diff.model(x, 4), first)
