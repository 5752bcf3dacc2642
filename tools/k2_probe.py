"""Dev measurement: stencil kernel (K2) time vs the preconditioner band width (cfg3 shape)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-motion-planner_amd")):
    sys.path.insert(0, p)
import nfopp
torch.random.manual_seed(1)
onf = nfopp.ONF(0, 10, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
B, N = 4096, 256
rng = np.random.default_rng(0)
starts = rng.uniform(5, 95, (B, 3)).astype(np.float32); goals = rng.uniform(5, 95, (B, 3)).astype(np.float32)
for w in (0.5,):
    hyper = nfopp.TrajectoryHyper(100, 5, 100, 0.1, 1e-3, 1, 10, 100, 5e-2, (0.9, 0.9), 1e-8, (0, 100, 0, 100))
    pl = nfopp.BatchPlanner(onf, B, N, hyper, velocity_hessian_weight=w)
    pl.init(starts, goals, (0, 100, 0, 100))
    eng = pl.engine
    eng.collision_eval()
    for _ in range(5): eng.update(False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): eng.update(False)
    e1.record(); torch.cuda.synchronize()
    tr = eng.traj.cpu().numpy()
    print("vh_weight %g: half width %d, K2 %.1f us  finite %s  max|traj| %.3g  max|m| %.3g" % (w, eng.half_width, e0.elapsed_time(e1) / 50 * 1e3, np.isfinite(tr).all(), np.nanmax(np.abs(tr)), float(eng.adam_m.abs().max())))
