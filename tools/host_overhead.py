"""Dev tool: host enqueue time per planner step vs GPU time (is the step launch-bound?)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "pytorch-motion-planner_amd")
import nfopp

def main():
    torch.manual_seed(0)
    onf = nfopp.ONF(0.0, 10.0, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
    B, N = 4096, 256
    rng = np.random.default_rng(0)
    bp = nfopp.BatchPlanner(onf, B, N, nfopp.TrajectoryHyper(bounds=(0, 100, 0, 100)), device="cuda", seed=1)
    s = np.concatenate([rng.uniform(0, 100, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(np.float32)
    g = np.concatenate([rng.uniform(0, 100, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(np.float32)
    bp.init(s, g, (0, 100, 0, 100))
    for rep in range(3):
        for _ in range(20):
            bp.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            bp.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("rep %d: host enqueue %.1f us/step, total %.1f us/step" % (rep, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))

main()
