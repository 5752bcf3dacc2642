"""Dev probe: host enqueue time vs device time of BatchPlanner.step(n) at small B (where the callers' loops are launch-bound)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-motion-planner_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import nfopp
torch.random.manual_seed(3)
onf = nfopp.ONF(0, 1, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
bounds = (-0.1, 3.1, -0.1, 3.1)
hyper = nfopp.TrajectoryHyper(collision_weight=3, direction_delta_weight=7, collision_beta=2, bounds=bounds)
Bs = [int(x) for x in os.environ.get("PROBE_B", "1,16").split(",")]
for B in Bs:
    bp = nfopp.BatchPlanner(onf, B, 256, hyper, device="cuda", seed=1)
    rng = np.random.default_rng(0)
    st = np.concatenate([rng.uniform(0.2, 0.8, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(np.float32)
    go = np.concatenate([rng.uniform(2.2, 2.8, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(np.float32)
    bp.init(st, go, bounds)
    bp.step(n=20); torch.cuda.synchronize()
    for n in (1, 10, 30, 50, 100, 300, 10, 300):
        reps = max(1, 300 // n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
        for _ in range(reps): bp.step(n=n)
        e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        k = reps * n
        print("B=%-3d step(n=%3d) x%3d: host enqueue %.1f us/step, wall %.1f us/step, device (events) %.1f us/step"
              % (B, n, reps, (t1 - t0) / k * 1e6, (t2 - t0) / k * 1e6, e0.elapsed_time(e1) * 1e3 / k), flush=True)
