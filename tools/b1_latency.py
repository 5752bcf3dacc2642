"""Dev measurement: wall time per `.step()` of the B=1 drop-in planner (BASELINE configs[1]: 1 x 256, ONF learning on)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-motion-planner_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import nfopp
from test_gpu_planner_api import _params
z = np.load(os.path.join(ROOT, "tests/golden/g9_full_steps.npz"))
for freeze in (False, True):
    torch.random.manual_seed(100); np.random.seed(400)
    cc = nfopp.CircleDirectedCollisionChecker(0.3, (0, 3, 0, 3)); cc.update_obstacle_points(z["obstacles"]); cc.update_boundaries(tuple(z["bounds"]))
    pl = nfopp.PlannerFactory.make_constrained_onf_planner(cc, _params(256))
    pl.init(z["start"], z["goal"], tuple(z["bounds"]))
    if freeze:
        pl._optimize_collision_model_freq = 10 ** 9; pl._step_count = 1
    for _ in range(30): pl.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 300
    for _ in range(K): pl.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("B=1 N=256 %s: %.3f ms/step  (%.0f steps/s, %.3g waypoint-evals/s)" % ("frozen ONF" if freeze else "ONF learning on", dt * 1e3, 1 / dt, 256 / dt))
