"""Dev measurement: where the wall time of the B = 1 drop-in `.step()` WITH ONF learning goes (BASELINE configs[1] shape)."""
import os, sys, time, cProfile, pstats, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-motion-planner_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import nfopp
from test_gpu_planner_api import _params
z = np.load(os.path.join(ROOT, "tests/golden/g9_full_steps.npz"))
torch.random.manual_seed(100); np.random.seed(400)
cc = nfopp.CircleDirectedCollisionChecker(0.3, (0, 3, 0, 3)); cc.update_obstacle_points(z["obstacles"]); cc.update_boundaries(tuple(z["bounds"]))
pl = nfopp.PlannerFactory.make_constrained_onf_planner(cc, _params(256))
pl.init(z["start"], z["goal"], tuple(z["bounds"]))
for _ in range(50): pl.step()
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K): pl.step()
torch.cuda.synchronize()
print("step: %.1f us" % ((time.perf_counter() - t0) / K * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(K): pl.step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
