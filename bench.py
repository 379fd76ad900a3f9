#!/usr/bin/env python3
"""bench.py — trajectory plans/sec of the MI355X reverse-diffusion sampler.

Contract (driver):  python bench.py --gpus N --steps K --warmup W.  N>1 runs one rank per GPU:
either the driver starts the ranks (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE), or —
when WORLD_SIZE is not set — this script starts them itself as child processes BEFORE anything
touches the GPU and relays rank 0's line.  RCCL only collects finished plans.  Rank 0 prints ONE
JSON line.

Headline workload (BASELINE.json configs[1]): PointMaze umaze-v2, guided policy without a guide
(`GuidedPolicy.sample_loop`, inpainting condition at horizon step 0), horizon 32, dim 128,
dim_mults (1,2,4), T = 100 denoise steps, batch 256 plans PER GPU (weak scaling), synthetic
weights/conditions from the portable generator, in-kernel Philox noise.
One "step" = one full sampling loop (T U-Net evaluations + posterior updates) over the batch.

Extra objects on the JSON line:
  roofline     — the dominant kernel (conv_gemm_f32, all tile variants): algorithmic FLOPs of
                 its launches / their HIP-event duration, measured in a second, instrumented
                 pass of the same loop (events between launches perturb the clean timing).
  cpu_baseline — the CPU oracle (torch-CPU restatement of the reference path, kind "port")
                 timed on this host's cores on a bounded sample of the same workload.
  configs      — the other BASELINE.json configurations (the T=1000 half of the metric, the
                 projected T=500 loop, the batch-1 planning call), one timed loop each in the same
                 invocation, each with its own ms_per_step / roofline / cpu_baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2516.6     # same guide: 16x the f32 MFMA rate (v_mfma_f32_32x32x16_f16), dense
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (arch key, batch per GPU, T, BASELINE.json config, description)
    "pointmaze_b256": ("pointmaze", 256, 100, 2,
                       "PointMaze umaze-v2 guided policy, H=32 dim=128 mults(1,2,4) T=100 batch=256/GPU"),
    "pointmaze_b1": ("pointmaze", 1, 100, 1,
                     "PointMaze guided policy, batch=1 (the get_action planning call; hipGraph replay)"),
    # BASELINE config 3: T=500 and the dynamics projection after every step (opt-in in the build,
    # the shipped reference loop never projects); "..._noproj" is the as-shipped variant
    "pointmaze_proj_t500_b256": ("pointmaze", 256, 500, 3,
                                 "PointMaze dynamics-aware policy, T=500, projection after every step "
                                 "(noise_schedule, strength 1), batch=256/GPU"),
    "pointmaze_noproj_t500_b256": ("pointmaze", 256, 500, 3,
                                   "PointMaze dynamics-aware policy as shipped (no projection), T=500, batch=256/GPU"),
    # the same loop with a value guide (ValueGuidedPolicy, guides/policies.py:243-271): the guide's gradient
    # comes from torch autograd on the user's value module every step, so the loop is driven from Python
    # (one dad_denoise_step per iteration, no hipGraph); value net = Linear(4,16) -> tanh -> Linear(16,1)
    "pointmaze_guided_b256": ("pointmaze", 256, 100, 2,
                              "PointMaze umaze-v2 VALUE-guided policy (tiny MLP value net, guide_weight 0.1), "
                              "H=32 dim=128 T=100 batch=256/GPU"),
    "halfcheetah_b128": ("halfcheetah", 128, 1000, 4,
                         "HalfCheetah medium-v2, H=32 dim=256 mults(1,4,8) T=1000 batch=128/GPU"),
    "door_b128": ("door", 128, 1000, 5,
                  "AdroitHand door expert-v2, H=32 dim=256 mults(1,2,4,8) T=1000 batch=128/GPU "
                  "(config 5 = 1024 plans sharded over 8 GPUs)"),
    # not BASELINE configurations: the same batch-1 planning call (`get_action`) on the two wide
    # architectures, where a denoise step is an HBM weight stream of 1.2 / 1.3 GB
    "halfcheetah_b1": ("halfcheetah", 1, 1000, None,
                       "HalfCheetah architecture, batch=1 (the get_action planning call; hipGraph replay)"),
    "door_b1": ("door", 1, 1000, None,
                "AdroitHand door architecture, batch=1 (the get_action planning call; hipGraph replay)"),
}
# the other BASELINE configurations timed after the headline, in this order (+ the two batch-1
# calls on the wide nets)
EXTRA_CONFIGS = ["pointmaze_b1", "pointmaze_proj_t500_b256", "pointmaze_guided_b256", "halfcheetah_b128", "door_b128",
                 "halfcheetah_b1", "door_b1"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pointmaze_b256", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "f16x3"],
                    help="conv arithmetic: exact fp32 MFMA, or split-f16 operands (3 f16 MFMAs per "
                         "product block, fp32 accumulation; same parity gates)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the sampling loop as one hipGraph (auto: batches <= 32, which are launch-bound)")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the second pass with the other conv arithmetic")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the other BASELINE configurations (T=1000 nets, projected loop, batch 1)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="independent sampling loops kept in flight on separate HIP streams "
                         "(diagnostic: shows how much of a step is dependency bubbles; the "
                         "headline number uses 1)")
    return ap.parse_args()


def self_launch(args) -> int:
    """--gpus N>1 without a launcher: start the N ranks as a child `torch.distributed.run` and
    relay its output.  This parent never initialises the GPU (it does not even import torch) and
    never execs: it waits for the child and exits with its code."""
    with socket.socket() as s:                       # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch                                     # noqa: E402  (after the launch decision)
    sys.path.insert(0, ROOT)
    from dynamics_aware_diffusion_amd import GaussianDiffusion, GuidedPolicy, TemporalUnet
    from dynamics_aware_diffusion_amd.utils import synth

    class BenchNormalizer:
        """Synthetic normaliser statistics (SURVEY 8(d) cfg 3: mean ~ N(0,1), std ~ U(0.5,1.5))."""

        def __init__(self, od: int, ad: int):
            self.obs_mean = synth.normal_like(41, "bench.norm.obs_mean", (od,))
            self.obs_std = 1.0 + synth.uniform(41, "bench.norm.obs_std", (od,), 0.5)
            self.action_mean = synth.normal_like(41, "bench.norm.act_mean", (ad,))
            self.action_std = 1.0 + synth.uniform(41, "bench.norm.act_std", (ad,), 0.5)

        def normalize_observations(self, obs):
            return (obs - self.obs_mean) / self.obs_std

        def unnormalize_actions(self, a):
            return a * self.action_std + self.action_mean

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the sampler has no CPU path")
    # one rank per GPU; DAD_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box) folds ranks onto device 0
    ngpu = torch.cuda.device_count()
    share = os.environ.get("DAD_BENCH_SHARE_GPU") == "1"
    device = torch.device("cuda", (local_rank % ngpu if share else local_rank) if world > 1 else 0)
    torch.cuda.set_device(device)
    dist = None
    host_collectives = os.environ.get("DAD_BENCH_BACKEND", "nccl") != "nccl"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DAD_BENCH_BACKEND", "nccl")        # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def usable_cores() -> int:
        """Host cores this process may really use: min(affinity mask, cgroup CPU quota)."""
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
        return max(1, n)

    def build_policy(workload: str, precision: str):
        arch, batch, T, _, _ = WORKLOADS[workload]
        od, ad, dim, mults, _ = synth.ARCHS[arch]
        td = od + ad
        unet = TemporalUnet(td, dim=dim, dim_mults=mults)
        unet.precision = precision
        state = synth.synth_unet_state(td, dim, mults, seed=0)
        unet.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        diff = GaussianDiffusion(unet, 32, od, ad, n_timesteps=T).to(device)
        diff.sampler_rng = "philox"
        if "proj_t500" in workload:               # config 3: double integrator dt=0.1, n=4, m=2, D=196
            from dynamics_aware_diffusion_amd import DynamicsAwarePolicy
            from dynamics_aware_diffusion_amd.dynamics import ProjectionMatrixBuilder
            import contextlib
            import io
            import numpy as np
            dt = 0.1
            A = np.eye(4); A[0, 2] = A[1, 3] = dt
            Bm = np.zeros((4, 2)); Bm[0, 0] = Bm[1, 1] = 0.5 * dt * dt; Bm[2, 0] = Bm[3, 1] = dt
            with contextlib.redirect_stdout(io.StringIO()):
                Pm = ProjectionMatrixBuilder(A, Bm, 4, 2).get_projection_matrix(32)
                policy = DynamicsAwarePolicy(diff, projection_matrix=Pm, normalizer=BenchNormalizer(od, ad),
                                             state_dim=4, observation_dim=od, action_dim=ad, horizon=32,
                                             projection_schedule="noise_schedule", projection_strength=1.0,
                                             project_during_sampling=not workload.startswith("pointmaze_noproj"))
        elif "guided" in workload:
            from dynamics_aware_diffusion_amd import ValueGuidedPolicy
            value = torch.nn.Sequential(torch.nn.Linear(od, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1)).to(device)
            with torch.no_grad():
                for i, prm in enumerate(value.parameters()):
                    prm.copy_(torch.from_numpy(synth.uniform(31, f"bench.value.{i}", tuple(prm.shape), 0.7)))
            policy = ValueGuidedPolicy(diff, None, value, guide_weight=0.1)
        else:
            policy = GuidedPolicy(diff, normalizer=None)
        cond = torch.zeros(1, td)
        cond[0, :od] = torch.from_numpy(synth.uniform(1, "bench.cond", (od,), 0.9))
        return policy, diff, {0: cond.to(device)}, state

    def cpu_baseline(workload: str, state, budget_s: float, min_steps: int = 3):
        """Time the oracle's denoise step on the host cores (bounded sample, extrapolated x T)."""
        from oracle import denoiser as od_
        arch, batch, T, _, _ = WORKLOADS[workload]
        od, ad, dim, mults, _ = synth.ARCHS[arch]
        td = od + ad
        cores = usable_cores()
        torch.set_num_threads(cores)
        w = {k: torch.from_numpy(v) for k, v in state.items()}
        sched = od_.schedule_buffers("cosine", T)
        x = torch.from_numpy(synth.normal_like(2, "bench.cpu.x", (batch, 32, td)))
        z = torch.from_numpy(synth.normal_like(2, "bench.cpu.z", (batch, 32, td)))
        t = torch.full((batch,), T // 2, dtype=torch.long)
        with torch.no_grad():
            od_.denoise_step(w, sched, x, t, z)                 # warm-up
            n, t0 = 0, time.perf_counter()
            while True:
                od_.denoise_step(w, sched, x, t, z)
                n += 1
                el = time.perf_counter() - t0
                if (el >= budget_s and n >= min_steps) or n >= 4 * T:
                    break
        step_s = el / n
        return {
            "value": batch / (step_s * T), "unit": "plans/s", "cores": cores, "kind": "port",
            "sample": f"{n} denoise steps of batch {batch} ({el:.1f} s), extrapolated x{T} steps/plan"
                      + (" (+ projection, negligible on the CPU)" if "proj_t500" in workload else ""),
            "ms_per_denoise_step": step_s * 1e3,
        }

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds: float) -> float:
        if dist is None:
            return seconds
        tmax = torch.tensor([seconds], device="cpu" if host_collectives else device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    def roofline_of(workload, policy_, diff_, cond_, precision, loop_s):
        """Rank-local pass (NO collective in here): HIP events bracket each denoiser evaluation's
        conv-GEMM launches on the launch stream.  Long loops are sampled on their first 100
        denoise steps (launch durations do not depend on t)."""
        arch, batch, T, _, _ = WORKLOADS[workload]
        od, ad, dim, mults, _ = synth.ARCHS[arch]
        td = od + ad
        f = synth.unet_flops_per_sample(td, dim, mults, 32)
        P = synth.count_params(synth.unet_param_shapes(td, dim, mults))
        eng = diff_._engine(device)
        graph, steps_keep = diff_.use_graph, diff_.n_timesteps
        diff_.use_graph = False               # events are recorded on eager launches
        diff_.n_timesteps = min(T, 100)       # evaluate.py:350-353 truncation
        eng.profile_enable(True)
        diff_.seed = 10_000
        policy_.sample_loop(batch_size=batch, conditions=cond_, row_offset=rank * batch)
        torch.cuda.synchronize()
        conv_ms, launches, conv_flops = eng.profile_read()
        eng.profile_enable(False)
        diff_.use_graph, diff_.n_timesteps = graph, steps_keep
        # The event-bracketed pass runs a few per cent slower than the timed loop (eager launches, event
        # records between denoiser evaluations), and a kernel cannot take longer than the step that
        # contains it: the conv time per denoise step is the SMALLER of the bracketed figure and the
        # clean step time of the timed loop (then a lower bound on the rate: the step also holds the
        # posterior / projection kernels).
        step_ms_clean = loop_s * 1e3 / T
        conv_ms_step_events = conv_ms / diff_min(T)
        conv_ms_step = min(conv_ms_step_events, step_ms_clean)
        achieved = (conv_flops / diff_min(T)) / (conv_ms_step * 1e-3) / 1e12
        traffic, traffic_note = None, ""
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                table = json.load(open(tpath))
                key = workload if precision == "fp32" else workload + ":" + precision
                if key not in table and workload.endswith("noproj_t500_b256"):
                    # config 3 launches exactly the conv kernels of config 2 (same net, same batch);
                    # the projection kernel is not a conv launch
                    key = "pointmaze_b256" if precision == "fp32" else "pointmaze_b256:" + precision
                    traffic_note = " — the conv launches of pointmaze_b256, which this workload repeats"
                traffic = table.get(key)
            except Exception:
                traffic = None
        hbm_floor = 4 * P + 12 * batch * 32 * td              # SURVEY 8(d) floor per denoise step
        step_s = loop_s / T
        mfma = {
            "bound": "mfma", "kernel": "dad::conv_gemm_f32<*> (all tile variants)",
            "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
            "traffic_source": ("profiles/traffic.json (rocprofv3 PMC pass of this workload; not re-measured in this run)"
                               + traffic_note) if traffic is not None else None,
            "launches_per_denoise_step": launches / diff_min(T),
            "avg_launch_us": conv_ms_step * 1e3 / max(launches / diff_min(T), 1),
            "flops_per_launch": conv_flops / max(launches, 1),
            "conv_ms_per_denoise_step": conv_ms_step,
            "conv_ms_per_denoise_step_event_pass": conv_ms_step_events,
            "achieved_source": ("HIP events around each denoiser evaluation's conv launches"
                                if conv_ms_step_events <= step_ms_clean else
                                "clean step time of the timed loop (the event pass ran slower than the step: lower bound)"),
            "whole_step_tflops": f * batch / step_s / 1e12,
            "hbm_model_bytes_per_denoise_step": hbm_floor,
            "hbm_frac": hbm_floor / step_s / (PEAK_HBM_GBS * 1e9),
        }
        if precision == "f16x3":
            # split-f16 executes 3 f16 MFMAs per product block: its roofline is the f16 matrix peak
            # (executed FLOPs = 3 x algorithmic); the ratio to the fp32 peak is kept as information
            mfma["algorithmic_vs_fp32_mfma_peak"] = mfma["frac"]
            mfma["achieved"] = 3 * achieved
            mfma["peak"] = PEAK_F16_MFMA_TFLOPS
            mfma["frac"] = 3 * achieved / PEAK_F16_MFMA_TFLOPS
            mfma["note"] = "executed f16 MFMA FLOPs (3 per fp32 product) against the dense f16 peak"
        if batch < 5:                          # below the ridge point the weight stream binds
            return {
                "bound": "hbm", "kernel": "whole denoise step (weight stream, launch-latency bound)",
                "achieved": hbm_floor / step_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": hbm_floor / step_s / (PEAK_HBM_GBS * 1e9),
                # HBM bytes per denoise step from the committed PMC pass (per conv launch x launches)
                "traffic": None if traffic is None else traffic * launches / diff_min(T),
                "traffic_source": mfma["traffic_source"],
                "algorithmic_bytes_per_denoise_step": hbm_floor, "us_per_denoise_step": step_s * 1e6,
                "conv_gemm": mfma,
            }
        return mfma

    def diff_min(T):
        return min(T, 100)

    def run_workload(workload: str, steps: int, warmup: int, *, headline: bool):
        arch, batch, T, cfg_no, desc = WORKLOADS[workload]
        od, ad, dim, mults, _ = synth.ARCHS[arch]
        td = od + ad
        policy, diff, cond, state = build_policy(workload, args.precision)
        # small batches are launch-bound: replay one hipGraph
        diff.use_graph = batch <= 32 if args.graph == "auto" else args.graph == "on"
        gathered = torch.empty(world * batch, 32, td, device=device) if world > 1 else None

        def one_step(k: int):
            diff.seed = 1000 + k                                  # fresh noise every loop
            plans = policy.sample_loop(batch_size=batch, conditions=cond, row_offset=rank * batch)
            if dist is not None:                                   # collect finished plans (RCCL)
                if host_collectives:                               # gloo rehearsal: stage through host
                    parts = [torch.empty(plans.shape) for _ in range(world)]
                    dist.all_gather(parts, plans.cpu())
                else:
                    dist.all_gather_into_tensor(gathered, plans)
            return plans

        extra = []
        if headline and args.inflight > 1:        # one policy/engine + stream per loop in flight
            extra = [(build_policy(workload, args.precision), torch.cuda.Stream(device))
                     for _ in range(args.inflight - 1)]
        inflight = args.inflight if headline else 1

        def run_steps(first: int, count: int):
            out = None
            for k in range(count):
                slot = k % inflight
                if slot == 0:
                    out = one_step(first + k)
                else:
                    (pol_k, diff_k, cond_k, _), st = extra[slot - 1]
                    with torch.cuda.stream(st):
                        diff_k.seed = 1000 + first + k
                        pol_k.sample_loop(batch_size=batch, conditions=cond_k, row_offset=rank * batch)
            return out

        if headline:
            run_steps(0, max(warmup, inflight if inflight > 1 else 0))
        else:
            # warm-up of the long loops: a truncated loop touches every kernel, workspace and table
            keep = diff.n_timesteps
            if T > 200 and not diff.use_graph:
                diff.n_timesteps = 20
            for k in range(max(1, warmup)):
                one_step(k)
            diff.n_timesteps = keep
        fence()
        t0 = time.perf_counter()
        plans = run_steps(warmup, steps)
        fence()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        assert torch.isfinite(plans).all()
        loop_s = elapsed / steps

        roof = None
        if rank == 0:
            roof = roofline_of(workload, policy, diff, cond, args.precision, loop_s)

        # ---- the other conv arithmetic, same workload, same run (one GPU only; reported beside
        # the headline, never as `value`)
        def run_alt():
            other = "f16x3" if args.precision == "fp32" else "fp32"
            pol2, diff2, cond2, _ = build_policy(workload, other)
            diff2.use_graph = diff.use_graph

            def alt_steps(first, count):
                out_ = None
                for k in range(count):
                    diff2.seed = 1000 + first + k
                    out_ = pol2.sample_loop(batch_size=batch, conditions=cond2, row_offset=0)
                return out_

            alt_steps(0, max(1, warmup))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            plans2 = alt_steps(warmup, steps)
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            assert torch.isfinite(plans2).all()
            # same seeds as the headline loop: the two arithmetics must agree to the fp32 tolerance
            diff.seed = diff2.seed = 4242
            pa = policy.sample_loop(batch_size=batch, conditions=cond, row_offset=0)
            pb = pol2.sample_loop(batch_size=batch, conditions=cond2, row_offset=0)
            return {
                "conv_arithmetic": other, "value": batch * steps / el2, "unit": "plans/s",
                "ms_per_step": el2 / steps * 1e3,
                "max_abs_diff_vs_headline_plans": float((pa - pb).abs().max()),
                "roofline": roofline_of(workload, pol2, diff2, cond2, other, el2 / steps),
            }

        alt = None
        if headline and rank == 0 and world == 1 and args.inflight == 1 and not args.no_alt:
            try:                                  # a failure here must not cost the headline line
                alt = run_alt()
            except Exception as exc:              # noqa: BLE001
                alt = {"error": repr(exc)}

        base = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            try:
                base = cpu_baseline(workload, state, args.cpu_seconds if headline else 3.0)
            except Exception as exc:              # noqa: BLE001
                base = {"error": repr(exc)}

        out = {
            "metric": "trajectory plans/sec (H=32, T=%d)" % T,
            "value": world * batch * steps / elapsed, "unit": "plans/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": loop_s * 1e3,
            "ms_per_denoise_step": loop_s * 1e3 / T,
            "config": {"workload": desc, "baseline_config": cfg_no, "arch": arch, "batch_per_gpu": batch,
                       "global_batch": world * batch, "horizon": 32, "denoise_steps": T,
                       "rng": "in-kernel philox",
                       "conv_arithmetic": args.precision + (" (exact fp32 MFMA)" if args.precision == "fp32"
                                                            else " (split-f16 operands, fp32 accumulate)"),
                       "hipgraph": bool(diff.use_graph), "sharding": f"batch x{world}, gather at end",
                       "loops_in_flight": inflight},
            "roofline": roof, "cpu_baseline": base,
        }
        if headline:
            out["alt_precision"] = alt
        del policy, diff, extra
        torch.cuda.empty_cache()
        return out

    def training_step():
        """one training step's device work on the headline architecture — the
        differentiable GaussianDiffusion.loss (training forward: every activation kept) and loss.backward()
        through the engine's explicit backward pass — next to the oracle's autograd on the host cores."""
        arch, batch = "pointmaze", 256
        od, ad, dim, mults, T = synth.ARCHS[arch]
        td = od + ad
        unet = TemporalUnet(td, dim=dim, dim_mults=mults)
        state = synth.synth_unet_state(td, dim, mults, seed=0)
        unet.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        diff = GaussianDiffusion(unet, 32, od, ad, n_timesteps=T).to(device)
        x0 = torch.from_numpy(synth.normal_like(3, "train.x0", (batch, 32, td))).to(device).clamp(-1, 1)
        fwd, bwd = [], []
        for rep in range(4):
            for prm in diff.parameters():
                prm.grad = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.enable_grad():
                loss = diff.loss(x0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if rep:
                fwd.append(t1 - t0)
                bwd.append(t2 - t1)
        # steady state: steps issued back to back as a training loop does (the host side of step i + 1 runs under
        # the kernels of step i); forward_ms / backward_ms above are each bracketed by a synchronisation
        n_steady = 10
        steady = None
        for timed in (False, True):                # (two untimed steps first: the allocator settles on the pattern
            torch.cuda.synchronize()               #  of several steps in flight)
            t0 = time.perf_counter()
            for _ in range(n_steady if timed else 2):
                for prm in diff.parameters():
                    prm.grad = None
                with torch.enable_grad():
                    diff.loss(x0).backward()
            torch.cuda.synchronize()
            steady = (time.perf_counter() - t0) / n_steady
        # ... and with an optimiser step in the loop: every parameter changes, the next forward re-derives the
        # engine's packed copies on the device (dad_model_refresh_weights: one repack launch + one copy launch)
        opt = torch.optim.SGD(diff.parameters(), lr=1e-6)
        steady_opt = None
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_steady if timed else 2):
                opt.zero_grad(set_to_none=True)
                with torch.enable_grad():
                    diff.loss(x0).backward()
                opt.step()
            torch.cuda.synchronize()
            steady_opt = (time.perf_counter() - t0) / n_steady
        f = synth.unet_flops_per_sample(td, dim, mults, 32) * batch
        out = {"workload": f"{arch} batch {batch}: GaussianDiffusion.loss + loss.backward() (fp32)",
               "forward_ms": min(fwd) * 1e3, "backward_ms": min(bwd) * 1e3,
               "steady_ms_per_step": steady * 1e3,
               "steady_ms_per_step_with_sgd_and_weight_refresh": steady_opt * 1e3,
               "samples_per_s": batch / steady,
               "samples_per_s_synchronised": batch / (min(fwd) + min(bwd)),
               "conv_tflops_algorithmic_steady": 3 * f / steady / 1e12,
               "backward_conv_tflops_algorithmic": 2 * f / min(bwd) / 1e12,
               "note": "steady_* = 10 steps issued back to back, one synchronisation at the end (forward_ms / "
                       "backward_ms are bracketed by synchronisations each); host work included (time MLPs and loss "
                       "in torch, workspace allocation)"}
        if not args.no_cpu_baseline:
            from oracle import denoiser as od_
            cores = usable_cores()
            torch.set_num_threads(cores)
            cb = 32
            w = {k: torch.from_numpy(v) for k, v in state.items()}
            sched = od_.schedule_buffers("cosine", T)
            xs = x0[:cb].cpu()
            tt = torch.arange(cb) % T
            nz = torch.from_numpy(synth.normal_like(3, "train.noise", (cb, 32, td)))
            od_.training_gradients(w, sched, xs, tt, nz)
            t0 = time.perf_counter()
            od_.training_gradients(w, sched, xs, tt, nz)
            el = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": cb / el, "unit": "samples/s", "cores": cores, "kind": "port",
                                   "sample": f"one oracle forward + autograd backward of batch {cb} ({el:.2f} s)"}
        return out

    head = run_workload(args.workload, args.steps, args.warmup, headline=True)
    configs = None
    if not args.no_configs and args.inflight == 1:
        configs = {}
        for name in EXTRA_CONFIGS:
            if name == args.workload:
                continue
            try:                                   # an extra configuration must not cost the headline
                # loops timed: the T=1000 nets take seconds per loop (one is enough), the short ones
                # are repeated so that a single disturbed loop does not become the number
                _, b_, T_, _, _ = WORKLOADS[name]
                loops = (5 if T_ <= 100 else 2) if b_ == 1 else (3 if T_ <= 500 else 1)
                configs[name] = run_workload(name, loops, 1, headline=False)
            except Exception as exc:               # noqa: BLE001
                if dist is not None:
                    raise                          # ranks must stay in step: fail loudly
                configs[name] = {"error": repr(exc)}

    training = None
    if rank == 0 and world == 1 and configs is not None:
        try:
            training = training_step()
        except Exception as exc:                   # noqa: BLE001
            training = {"error": repr(exc)}

    if rank == 0:
        line = {
            "metric": head["metric"], "value": head["value"], "unit": "plans/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f16x3 (split-f16 operand pairs, f32 accumulate)",
            "data": "synthetic", "config": head["config"], "roofline": head["roofline"],
            "cpu_baseline": head["cpu_baseline"], "alt_precision": head.get("alt_precision"),
            "configs": configs, "training_step": training,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
