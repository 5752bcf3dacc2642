"""Prints the measured HIP-vs-reference error levels on the bench-mr-settings fixtures (run on the GPU box); the gates in
tests/conftest.py / tests/test_gpu_benchmr.py are set from these and from the reference's own 1-ulp conditioning."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pytorch-motion-planner_amd")):
    sys.path.insert(0, p)
import gpu_common as gc  # noqa: E402
import nfopp  # noqa: E402
from nfopp import _lib  # noqa: E402

if os.environ.get("NFOPP_DEV_LIB"):      # margins of a development build
    _lib.LIB_PATH = os.environ["NFOPP_DEV_LIB"]
from conftest import load_golden  # noqa: E402
from oracle import nfopp_oracle as orc  # noqa: E402

F32 = np.float32


def stats(name, a, b):
    e = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).reshape(-1)
    print("   %-6s max %.3e  p99 %.3e  p90 %.3e  median %.3e" % (name, e.max(), np.percentile(e, 99), np.percentile(e, 90), np.median(e)))


z = load_golden("g14_benchmr_batch.npz")
onf, cfg = gc.make_onf(z["cfg"], z["params"])
hp = orc.Hyper.from_npz(z)
B, N = z["traj0"].shape[:2]
s = dict(traj=z["traj0"].copy(), start=z["starts"], goal=z["goals"], lam=np.zeros((B, N + 1), F32),
         cm=np.zeros((B, N), F32), adam_m=np.zeros((B, N, 3), F32), adam_v=np.zeros((B, N, 3), F32), adam_step=0)
eng = gc.engine_from_state(onf, s, hp)
step_count = 1
for k in range(int(z["steps"])):
    eng.optimize_trajectory(z["t"][:, k], want_terms=False)
    if step_count % 10 == 0:
        eng.reparametrize()
    step_count += 1
    if "k%d_traj" % (k + 1) in z.files:
        print("g14 after step", k + 1)
        tr = eng.traj.cpu().numpy()
        stats("xy", tr[..., :2], z["k%d_traj" % (k + 1)][..., :2])
        stats("theta", tr[..., 2], z["k%d_traj" % (k + 1)][..., 2])
        stats("lam", eng.lam.cpu().numpy(), z["k%d_lam" % (k + 1)])
        stats("cm", eng.cm.cpu().numpy(), z["k%d_cm" % (k + 1)])

import test_gpu_benchmr as tb  # noqa: E402
z = load_golden("g15_full_steps_n256.npz")
torch.random.manual_seed(100)
np.random.seed(400)
cc = nfopp.CircleDirectedCollisionChecker(0.3, (0, 3, 0, 3))
cc.update_obstacle_points(z["obstacles"])
cc.update_boundaries(tuple(z["bounds"]))
planner = nfopp.PlannerFactory.make_constrained_onf_planner(cc, tb._corridor_params(256))
planner.init(z["start"], z["goal"], tuple(z["bounds"]))
for k in range(int(z["steps"])):
    planner.step()
    print("g15 step", k)
    stats("checked", planner.checked_positions.as_vec(), z["k%d_checked" % k])
    stats("traj", planner._trajectory.detach().cpu().numpy(), z["k%d_traj" % k])
    if "k%d_params" % k in z.files:
        stats("params", planner._collision_model.flat_parameters.cpu().numpy(), z["k%d_params" % k])
    stats("lam", planner._constraint_multipliers.cpu().numpy(), z["k%d_lam" % k])
    stats("cm", planner._collision_multipliers.cpu().numpy(), z["k%d_cm" % k])

from conftest import BENCHMR_FIXTURES  # noqa: E402
for name, ks in BENCHMR_FIXTURES:
    z = load_golden(name)
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = orc.Hyper.from_npz(z)
    s3 = gc.state_of(z, "g3_")
    eng = gc.engine_from_state(onf, s3, hp)
    step_count, done = s3["step_count"], 0
    for K in ks:
        while done < K:
            eng.optimize_trajectory(z["g6_t"][done][None], want_terms=False)
            if step_count % 10 == 0:
                eng.reparametrize()
            step_count += 1
            done += 1
        print(name, "frozen-field rollout, after", K, "steps")
        pre = "g6_k%d_" % K
        tr = eng.traj.cpu().numpy()[0]
        stats("xy", tr[:, :2], z[pre + "traj"][:, :2])
        stats("theta", tr[:, 2], z[pre + "traj"][:, 2])
        stats("lam", eng.lam.cpu().numpy()[0], z[pre + "lam"])
        stats("cm", eng.cm.cpu().numpy()[0], z[pre + "cm"])
