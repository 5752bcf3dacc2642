"""Helpers shared by the GPU parity tests (HIP path <-> oracle / golden fixtures through the C ABI)."""
import numpy as np
import torch

import nfopp
from oracle import nfopp_oracle as orc

F32 = np.float32
DEV = "cuda"


def make_onf(cfg_vec, flat):
    cfg = orc.OnfConfig.from_vector(cfg_vec)
    m = nfopp.ONF(cfg.mean, cfg.sigma, use_cos=cfg.use_cos, use_normal_init=False, bias=cfg.bias,
                  angle_encoding=cfg.angle_encoding).to(DEV)
    m.load_flat(torch.tensor(np.asarray(flat, F32)))
    return m, cfg


def hyper_from(hp):
    """oracle Hyper -> nfopp TrajectoryHyper"""
    return nfopp.TrajectoryHyper(hp.collision_weight, hp.angle_weight, hp.constraint_deltas_weight, hp.multipliers_lr,
                                 hp.collision_multipliers_lr, hp.boundary_weight, hp.collision_beta,
                                 hp.direction_delta_weight, hp.lr, (hp.beta1, hp.beta2), hp.eps, hp.bounds)


def engine_from_state(onf, s, hp, vh_weight=0.5):
    """s: dict with traj [B,N,D], start, goal, lam, cm, adam_m, adam_v (numpy), adam_step"""
    B, N, D = s["traj"].shape
    eng = nfopp.TrajectoryEngine(onf, B, N, D, hyper_from(hp) if not isinstance(hp, nfopp.TrajectoryHyper) else hp,
                                 vh_weight, DEV)
    eng.traj.copy_(torch.tensor(s["traj"]))
    eng.set_endpoints(s["start"], s["goal"])
    if D == 3:
        eng.lam.copy_(torch.tensor(s["lam"]))
        eng.cm.copy_(torch.tensor(s["cm"]))
    eng.adam_m.copy_(torch.tensor(s["adam_m"]))
    eng.adam_v.copy_(torch.tensor(s["adam_v"]))
    eng.adam_step = int(s["adam_step"])
    return eng


def state_of(z, prefix, reps=1):
    keys = ("traj", "start", "goal", "lam", "cm", "adam_m", "adam_v")
    s = {k: np.repeat(z[prefix + k][None].astype(F32), reps, axis=0) for k in keys}
    s["adam_step"] = int(z[prefix + "adam_step"])
    s["step_count"] = int(z[prefix + "step_count"])
    return s


def scaled_err(a, b):
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / (np.max(np.abs(b)) + 1e-12))


def philox_uniform_np(seed, ctr_lo, ctr_hi):
    """Philox4x32-10 word 0 -> 24-bit uniform: the oracle's restatement of csrc/common.h philox_uniform."""
    return orc.philox_uniform(seed, np.asarray(ctr_lo, np.uint64), ctr_hi)
