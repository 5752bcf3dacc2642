#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native NFOPP inner loop.

Metric (BASELINE.json): waypoint-evals/sec = trajectories x waypoints x planner steps / wall seconds.
Workload at N GPUs (BASELINE configs[2] per GPU, i.e. configs[3] at 8 GPUs): 4096 trajectories x 256 waypoints per
GPU, random-obstacle map (300 discs on 100 m x 100 m), SE(2) constrained planner with the bench-mr hyper-parameters
(reference scripts/run_bench_mr.py:45-63), ONF F=220 pre-fitted on the map and then FROZEN.  One "step" is one
planner `.step()` for every trajectory: fused collision sampling + ONF fwd/bwd (MFMA kernel), loss terms + H^-1 +
Adam + multiplier ascent (stencil kernel), and the arc-length reparametrisation every 10th step.  Inputs are
resident in HBM before the timed region; the per-step interpolation draws come from the in-kernel Philox stream.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Trajectories shard contiguously across ranks (weak scaling, 4096 per GPU); the frozen-ONF step has no data-path
collective.  Rank 0 prints ONE JSON line.  `roofline` = fused ONF kernel, algorithmic fp32 FLOPs (131 400 per
collision sample) over its HIP-event duration vs the 157.3 TFLOP/s fp32 MFMA peak.  `cpu_baseline` =
the oracle (oracle/nfopp_oracle.py, a numpy port with analytic gradients) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pytorch-motion-planner_amd"))
sys.path.insert(0, ROOT)

import nfopp  # noqa: E402

FLOP_PER_SAMPLE = 131400.0      # 65 700 FMA per collision sample (fwd 32 740 + input-bwd 32 960)
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
SPLIT_PRODUCTS = 6              # bf16x3 split path: partial products issued per fp32 multiply (csrc/onf_split.hip)
B_PER_GPU, N_WAYPOINTS = 4096, 256
BOUNDS = (0.0, 100.0, 0.0, 100.0)


def make_environment():
    rng = np.random.default_rng(1234)
    return rng.uniform(5, 95, (300, 2)), 1.5


def in_collision(xy, obstacles, radius):
    d2 = ((xy[:, None, :] - obstacles[None]) ** 2).sum(-1)
    out = (d2 < radius * radius).any(1)
    out |= (xy[:, 0] < BOUNDS[0]) | (xy[:, 0] > BOUNDS[1]) | (xy[:, 1] < BOUNDS[2]) | (xy[:, 1] > BOUNDS[3])
    return out


def free_poses(rng, n, obstacles, radius):
    out = np.zeros((0, 3))
    while len(out) < n:
        c = np.concatenate([rng.uniform(2, 98, (2 * n, 2)), rng.uniform(-np.pi, np.pi, (2 * n, 1))], 1)
        out = np.concatenate([out, c[~in_collision(c[:, :2], obstacles, radius + 0.5)]])
    return out[:n].astype(np.float32)


def make_onf(device, obstacles, radius, fit_iters, fit_points):
    """ONF of scripts/run_bench_mr.py:28-36 (sigma=10), fitted to the map with the HIP training kernel, then frozen."""
    torch.random.manual_seed(100)
    onf = nfopp.ONF(0, 10, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to(device)
    fitter = nfopp.OnfFitter(onf, lr=2e-2, betas=(0.9, 0.9), distributed=False)   # every rank fits the same field
    rng = np.random.default_rng(77)
    for _ in range(fit_iters):
        x = np.concatenate([rng.uniform(0, 100, (fit_points, 2)), rng.uniform(0, 2 * np.pi, (fit_points, 1))], 1)
        y = in_collision(x[:, :2], obstacles, radius).astype(np.float32)
        fitter.step(torch.tensor(x.astype(np.float32), device=device), torch.tensor(y, device=device), global_count=fit_points)
    return onf, float(fitter.last_loss) if fit_iters else float("nan")


def bench_hyper():
    # reference scripts/run_bench_mr.py:37-63
    return nfopp.TrajectoryHyper(collision_weight=100, angle_weight=5, constraint_deltas_weight=100, multipliers_lr=0.1,
                                 collision_multipliers_lr=1e-3, boundary_weight=1, collision_beta=10,
                                 direction_delta_weight=100, lr=5e-2, betas=(0.9, 0.9), eps=1e-8, bounds=BOUNDS)


def cpu_baseline(onf_flat, starts, goals, sample_b, steps):
    """Oracle (numpy port of the reference's algorithm) on the first `sample_b` trajectories of the workload."""
    from oracle import nfopp_oracle as orc
    try:
        import threadpoolctl
        threads = max([i["num_threads"] for i in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = 1
    cfg = orc.OnfConfig(0, 10, True, True, True)
    hp = orc.Hyper(100, 5, 100, 0.1, 1e-3, 1, 10, 100, 5e-2, 0.9, 0.9, 1e-8, BOUNDS)
    n = N_WAYPOINTS
    s = dict(traj=nfopp.straight_line_init(starts[:sample_b], goals[:sample_b], n), start=starts[:sample_b],
             goal=goals[:sample_b], lam=np.zeros((sample_b, n + 1), np.float32), cm=np.zeros((sample_b, n), np.float32),
             adam_m=np.zeros((sample_b, n, 3), np.float32), adam_v=np.zeros((sample_b, n, 3), np.float32),
             adam_step=0, step_count=0)
    hinv = orc.calculate_inv_hessian(n, 0.5)
    rng = np.random.default_rng(5)
    draw = lambda: rng.uniform(0, 1, (sample_b, n - 1)).astype(np.float32)  # noqa: E731
    orc.planner_step(s, draw(), onf_flat, cfg, hp, hinv)  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.planner_step(s, draw(), onf_flat, cfg, hp, hinv)
    dt = time.perf_counter() - t0
    return {"value": sample_b * n * steps / dt, "unit": "waypoint-evals/s", "cores": int(threads), "kind": "port",
            "sample": "%d trajectories x %d waypoints x %d steps of the same workload (numpy oracle, %.1f s)"
                      % (sample_b, n, steps, dt)}


def pmc_traffic(batch, n):
    """HBM bytes per launch of the fused kernel from the committed PMC passes (profiles/r01_traffic.json, collected
    with tools/gpu_pmc.sh on this workload); None when the workload differs or the file is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            d = json.load(f)
        return float(d["bytes_per_launch"]) if (batch, n) == (B_PER_GPU, N_WAYPOINTS) else None
    except Exception:
        return None


def roofline(matrix_path, achieved, k1_ms, samples, traffic):
    """`achieved` is ALGORITHMIC fp32 FLOP/s of the fused ONF kernel in both cases.  fp32 path: against the fp32 MFMA
    peak.  Split path: the kernel runs on the bf16 matrix pipe and issues 6 bf16 partial products per fp32 multiply,
    so the peak that bounds it is the dense bf16 peak / 6 (fp32-equivalent); the fp32-MFMA ratio is given beside it."""
    out = {"bound": "mfma", "achieved": achieved, "unit": "TFLOP/s", "traffic": traffic, "kernel_ms": k1_ms,
           "algorithmic_flop_per_launch": samples * FLOP_PER_SAMPLE}
    if matrix_path == "split":
        peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS
        out.update({"peak": peak, "frac": achieved / peak, "kernel": "onf_split_kernel<14,2,0>",
                    "pipe": "bf16 MFMA 16x16x32, %d partial products per fp32 multiply (exact 3-level operand split)" % SPLIT_PRODUCTS,
                    "pipe_peak": PEAK_BF16_MFMA_TFLOPS, "executed_tflops": achieved * SPLIT_PRODUCTS,
                    "vs_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS})
    else:
        out.update({"peak": PEAK_FP32_MFMA_TFLOPS, "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                    "kernel": "onf_fwd_bwd_kernel<14,2,0>"})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=B_PER_GPU)
    ap.add_argument("--fit-iters", type=int, default=300)
    ap.add_argument("--cpu-sample", type=int, default=256, help="trajectories in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--workload", choices=("cfg3", "cfg5"), default="cfg3",
                    help="cfg3 (default, the headline): 4096x256 frozen field.  cfg5: 4096x512 per GPU with continuous "
                         "ONF learning every step (device sampling + checker + MFMA fit + gradient all-reduce)")
    ap.add_argument("--spin-up", type=int, default=300, help="throw-away steps during setup (clock ramp), state reset after")
    ap.add_argument("--matrix-path", choices=("split", "fp32"), default="split",
                    help="fused ONF kernel: bf16x3 split-precision MFMA (default, fp32-faithful) or fp32 MFMA")
    args = ap.parse_args()
    from nfopp import _lib
    _lib.check(_lib.load().nfopp_set_matrix_path(1 if args.matrix_path == "split" else 0))
    global N_WAYPOINTS
    if args.workload == "cfg5":
        N_WAYPOINTS = 512

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # one process per GPU; the modulo only matters for a rehearsal of the N-rank flow on a box with fewer GPUs
    # (NFOPP_DIST_BACKEND=gloo, several ranks sharing a card) -- on the benchmark node it is the identity
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = os.environ.get("NFOPP_DIST_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(backend)

    B, N = args.batch_per_gpu, N_WAYPOINTS
    obstacles, radius = make_environment()
    onf, fit_loss = make_onf(device, obstacles, radius, args.fit_iters, 4096)
    rng = np.random.default_rng(4321)
    starts = free_poses(rng, world * B, obstacles, radius)
    goals = free_poses(rng, world * B, obstacles, radius)
    lo, hi = nfopp.shard_range(world * B, rank, world)
    checker = None
    if args.workload == "cfg5":   # continuous learning: ground truth on the device, forward-only constraints on
        checker = nfopp.DeviceCircleChecker(obstacles, radius, BOUNDS, device=device)
    planner = nfopp.BatchPlanner(onf, hi - lo, N, bench_hyper(), velocity_hessian_weight=0.5, device=device, seed=100,
                                 traj_index_offset=lo, checker=checker, fit_lr=2e-2, angle_offset=0.3)
    planner.init(starts[lo:hi], goals[lo:hi], BOUNDS)
    eng = planner.engine

    def one_step(ev=None):
        if checker is not None and planner.step_count % planner.fit_freq == 0:
            planner.fit_field()
        if ev is not None:
            ev[0].record()
        eng.collision_eval()
        if ev is not None:
            ev[1].record()
        eng.update(False)
        if planner.step_count % planner.reparam_freq == 0:
            eng.reparametrize()
        planner.step_count += 1

    # setup, not measurement: a fresh box starts at idle clocks and the first process pays one-off costs (code-object
    # load, DVFS ramp).  Spin the device up on throw-away steps, then restore the initial planner state.
    for _ in range(args.spin_up):
        one_step()
    torch.cuda.synchronize()
    if args.spin_up:
        planner.init(starts[lo:hi], goals[lo:hi], BOUNDS)
    for _ in range(args.warmup):
        one_step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(events[k])
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        k1_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
        samples = (hi - lo) * (N - 1)
        achieved = samples * FLOP_PER_SAMPLE / (k1_ms * 1e-3) / 1e12
        paths = planner.get_paths()
        finite = bool(np.isfinite(paths).all())
        out = {
            "metric": "waypoint-evals/sec", "value": world * B * N * args.steps / elapsed, "unit": "waypoint-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2] per GPU: %d trajectories x %d waypoints, random-obstacle map "
                                    "(300 discs r=1.5 on 100x100), SE(2) constrained planner, frozen pre-fitted ONF "
                                    "(F=220, fit loss %.3f)" % (B, N, fit_loss)) if checker is None else
                                   ("BASELINE configs[4] per GPU: %d trajectories x %d waypoints, forward-only constraints, "
                                    "continuous ONF learning every step on %d device-sampled poses per GPU, gradient "
                                    "all-reduce over ranks (last fit loss %.3f)"
                                    % (B, N, planner.sampler.B * planner.sampler.S, float(planner.fitter.last_loss))),
                       "trajectories_per_gpu": B, "waypoints": N, "global_batch": world * B,
                       "parallelism": "trajectory shards, no data-path collective" if checker is None else "trajectory shards + one all-reduce of the 33163-float ONF gradient buffer per step", "paths_finite": finite, "matrix_path": args.matrix_path,
                       "planner_steps_per_s": args.steps / elapsed},
            "roofline": roofline(args.matrix_path, achieved, k1_ms, samples,
                                 pmc_traffic(B, N) if checker is None else None),
        }
        if args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(onf.flat_parameters.cpu().numpy(), starts, goals,
                                               min(args.cpu_sample, world * B), args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
