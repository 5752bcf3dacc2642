#!/usr/bin/env python3
"""How far does the REFERENCE move when it is restarted one fp32 ulp away?  (build container only)

The bench-mr settings (scripts/run_bench_mr.py:37-63: w_col 100, beta 10, w_dir 100, lr 5e-2) are stiff, and from a
straight-line start Adam turns zero-up-to-rounding gradient entries into +-lr moves.  The gates of the fixtures made with
those settings (tests/conftest.py: BENCHMR_ROLLOUT_TOL, BENCHMR_BATCH_TOL) and the `parity.short` note of bench.py rest on
this measurement; this script makes it reproducible.

For every fixture state it rebuilds the reference planner exactly as tests/golden/make_golden.py did (same field, same
injected `t` stream), moves EVERY trajectory coordinate by one ulp with a random sign (TRIALS sign patterns), runs the
same steps and records the distance to the unperturbed reference run stored in the fixture:
  * traj_benchmr_n256 / n512: from the post-G3 state, K in (1, 10, 50) / (1, 10) full `.step()`s
  * g14_benchmr_batch: B = 4 problems from the straight-line start, snapshots after 1 / 3 / 12 steps
Output: tests/golden/g17_conditioning.npz (max over trials per quantity xy / theta / lambda / cm; for g14 also the
percentile the gate uses).  Imports the reference like make_golden.py does (two harness shims, reference untouched).

Usage:  MPLBACKEND=Agg python tools/ref_conditioning.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLDEN)
import make_golden as mg  # noqa: E402  (installs the shims, imports the reference)

F32 = np.float32
TRIALS = 6


def load_params(model, flat):
    sd, o = model.state_dict(), 0
    for k, v in sd.items():
        n = v.numel()
        sd[k] = torch.tensor(np.asarray(flat[o:o + n], F32)).reshape(v.shape)
        o += n
    assert o == len(flat)
    model.load_state_dict(sd)


def one_ulp(a, rng):
    a = np.asarray(a, F32)
    up = rng.integers(0, 2, a.shape).astype(bool)
    return np.where(up, np.nextafter(a, F32(np.inf)), np.nextafter(a, F32(-np.inf))).astype(F32)


def restore(planner, z, prefix, perturb_rng=None):
    """fixture state -> reference planner (trajectory, multipliers, Adam moments, counters)"""
    tr = z[prefix + "traj"]
    if perturb_rng is not None:
        tr = one_ulp(tr, perturb_rng)
    with torch.no_grad():
        planner._trajectory.copy_(torch.tensor(tr))
        planner._constraint_multipliers.copy_(torch.tensor(z[prefix + "lam"]))
        planner._collision_multipliers.copy_(torch.tensor(z[prefix + "cm"]))
    step = float(z[prefix + "adam_step"])
    opt = planner._trajectory_optimizer
    if step > 0:
        opt.state[planner._trajectory] = dict(step=torch.tensor(step), exp_avg=torch.tensor(z[prefix + "adam_m"]),
                                              exp_avg_sq=torch.tensor(z[prefix + "adam_v"]))
    else:
        opt.state.pop(planner._trajectory, None)
    planner._step_count = int(z[prefix + "step_count"])


def spread(planner, ref_traj, ref_lam, ref_cm):
    tr = planner._trajectory.detach().numpy()
    return np.asarray([np.abs(tr[:, :2] - ref_traj[:, :2]).max(), np.abs(tr[:, 2] - ref_traj[:, 2]).max(),
                       np.abs(planner._constraint_multipliers.detach().numpy() - ref_lam).max(),
                       np.abs(planner._collision_multipliers.detach().numpy() - ref_cm).max()], np.float64)


def rollout_fixture(name, n, ks, out):
    z = np.load(os.path.join(GOLDEN, name))
    start, goal = z["g3_start"], z["g3_goal"]
    res = {K: np.zeros(4) for K in ks}
    for trial in range(TRIALS + 1):                     # trial 0: unperturbed, must reproduce the fixture
        planner, _ = mg.make_benchmr_planner(n, start, goal, init_iters=0)
        load_params(planner._collision_model, z["params"])
        mg.freeze(planner)
        restore(planner, z, "g3_", np.random.default_rng(1000 + trial) if trial else None)
        mg.freeze(planner)
        done = 0
        for K in ks:
            while done < K:
                mg.draw_t(n, 7100 + done)
                planner.step()
                done += 1
            pre = "g6_k%d_" % K
            s = spread(planner, z[pre + "traj"], z[pre + "lam"], z[pre + "cm"])
            if trial == 0:
                assert s.max() == 0.0, (name, K, s)    # the harness replays the fixture bit for bit
            else:
                res[K] = np.maximum(res[K], s)
    tag = name.replace("traj_benchmr_", "").replace(".npz", "")
    for K in ks:
        out["rollout_%s_k%d" % (tag, K)] = res[K]
        print("%s K=%-3d 1-ulp restart spread (xy, theta, lambda, cm) = %s" % (name, K, " ".join("%.2e" % v for v in res[K])))


def batch_fixture(out):
    z = np.load(os.path.join(GOLDEN, "g14_benchmr_batch.npz"))
    snaps = [int(k) for k in z["snapshots"]]
    n, K = z["traj0"].shape[1], int(z["steps"])
    qs = {1: 99, 3: 99, 12: 90}
    mx = {k: np.zeros(4) for k in snaps}
    pc = {k: np.zeros(4) for k in snaps}
    for trial in range(TRIALS + 1):
        got = {k: dict(traj=[], lam=[], cm=[]) for k in snaps}
        for b in range(4):
            p, _ = mg.make_benchmr_planner(n, z["starts"][b], z["goals"][b], init_iters=0)
            load_params(p._collision_model, z["params"])
            mg.freeze(p)
            if trial:
                with torch.no_grad():
                    p._trajectory.copy_(torch.tensor(one_ulp(z["traj0"][b], np.random.default_rng(2000 + 10 * trial + b))))
            for k in range(K):
                mg.draw_t(n, 9500 + 100 * b + k)
                p.step()
                if k + 1 in got:
                    got[k + 1]["traj"].append(p._trajectory.detach().numpy().copy())
                    got[k + 1]["lam"].append(p._constraint_multipliers.detach().numpy().copy())
                    got[k + 1]["cm"].append(p._collision_multipliers.detach().numpy().copy())
        for k in snaps:
            tr, lam, cm = (np.stack(got[k][x]) for x in ("traj", "lam", "cm"))
            d = [np.abs(tr[..., :2] - z["k%d_traj" % k][..., :2]), np.abs(tr[..., 2] - z["k%d_traj" % k][..., 2]),
                 np.abs(lam - z["k%d_lam" % k]), np.abs(cm - z["k%d_cm" % k])]
            if trial == 0:
                assert max(x.max() for x in d) == 0.0, k
            else:
                mx[k] = np.maximum(mx[k], [x.max() for x in d])
                pc[k] = np.maximum(pc[k], [np.percentile(x, qs[k]) for x in d])
    for k in snaps:
        out["batch_k%d_max" % k] = mx[k]
        out["batch_k%d_pct" % k] = pc[k]
        out["batch_k%d_q" % k] = np.asarray(qs[k])
        print("g14 step %-2d max %s | p%d %s" % (k, " ".join("%.2e" % v for v in mx[k]), qs[k], " ".join("%.2e" % v for v in pc[k])))


if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {"trials": np.asarray(TRIALS)}
    rollout_fixture("traj_benchmr_n256.npz", 256, (1, 10, 50), out)
    rollout_fixture("traj_benchmr_n512.npz", 512, (1, 10), out)
    batch_fixture(out)
    np.savez_compressed(os.path.join(GOLDEN, "g17_conditioning.npz"), **out)
    print("wrote tests/golden/g17_conditioning.npz")
