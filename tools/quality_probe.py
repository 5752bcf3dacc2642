import sys, os, json, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/pytorch-motion-planner_amd")
import bench, nfopp
torch.cuda.set_device(0)
for wl in ("cfg3", "cfg4"):
    env = bench.GridMap() if wl == "cfg4" else bench.DiscMap()
    onf, fit = bench.make_onf("cuda", env, 300, 4096)
    rng = np.random.default_rng(4321)
    B, N = 4096, 256
    starts, goals = env.free_poses(rng, B), env.free_poses(rng, B)
    truth = env.device_checker("cuda")
    pl = nfopp.BatchPlanner(onf, B, N, bench.bench_hyper(), device="cuda", seed=100)
    pl.init(starts, goals, bench.BOUNDS)
    for k in range(1001):
        pl.step()
        if k % 200 == 0:
            col, ln = pl.evaluate(checker=truth, sub=4, early_stop=(k >= 200))
            act = pl.engine.active
            print(wl, "step", k, "collision-free %.3f" % (1 - col.float().mean().item()), "found %.3f" % torch.isfinite(pl.best_length).float().mean().item(),
                  "active %.3f" % (act.float().mean().item() if act is not None else 1.0), "fit", round(fit, 3), flush=True)
