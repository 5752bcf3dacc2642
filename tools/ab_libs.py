"""Dev tool: A/B two builds of libnfopp_hip.so IN ONE PROCESS, interleaved rounds (guide rule 24): the fused ONF kernel on
the cfg3 shape.  Usage: python tools/ab_libs.py build/<variant>/libnfopp_hip.so   (compared with the product build)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-motion-planner_amd"))
import nfopp  # noqa: E402
from nfopp import _lib  # noqa: E402


def bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib._SIGNATURES.items():
        if hasattr(lib, name):          # (a variant built before an ABI addition lacks the new symbols)
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


libs = {"product": bind(_lib.LIB_PATH), "variant": bind(os.path.abspath(sys.argv[1]))}
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "k1"     # k1: fused ONF kernel, k2: stencil / optimiser kernel
torch.manual_seed(0)
onf = nfopp.ONF(0.0, 10.0, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
B, N = 4096, 256
traj = torch.rand(B, N, 3, device="cuda") * torch.tensor([100.0, 100.0, 6.0], device="cuda")
t = torch.zeros(B, N - 1, device="cuda")
out = {k: torch.zeros(B, N - 1, 4, device="cuda") for k in libs}
cfg = onf.config_c()


def run(lib, o):
    rc = lib.nfopp_traj_collision_eval(cfg, _lib.ptr(onf.flat_parameters), _lib.ptr(traj), B, N, 3, _lib.ptr(t), 1, 7, 0, 0,
                                       _lib.ptr(o), None, None, _lib.stream_ptr())
    assert rc == 0, lib.nfopp_last_error()


if KERNEL == "k2":
    from nfopp.engine import TrajectoryEngine, TrajectoryHyper
    hyper = TrajectoryHyper(100, 5, 100, 0.1, 1e-3, 1, 10, 100, 5e-2, (0.9, 0.9), 1e-8, (0, 100, 0, 100))
    engs = {}
    for k in libs:
        e = TrajectoryEngine(onf, B, N, 3, hyper, 0.5, "cuda")
        e.traj.copy_(traj)
        e.set_endpoints(traj[:, 0].cpu().numpy(), traj[:, -1].cpu().numpy())
        run(libs[k], e.onf_out)
        e.t.copy_(t)
        engs[k] = e

    def run(lib, o, _libs=libs, _engs=engs):   # noqa: F811
        e = _engs["product" if lib is _libs["product"] else "variant"]
        e.adam_step += 1
        hp = e.hyper.to_c(e.adam_step)
        rc = lib.nfopp_traj_update(hp, e.B, e.N, e.D, _lib.ptr(e.traj), _lib.ptr(e.start), _lib.ptr(e.goal), _lib.ptr(e.lam),
                                   _lib.ptr(e.cm), _lib.ptr(e.adam_m), _lib.ptr(e.adam_v), _lib.ptr(e.t), _lib.ptr(e.onf_out),
                                   _lib.ptr(e.hinv_band), e.half_width, e.interior[0], e.interior[1], None, None, _lib.stream_ptr())
        assert rc == 0, lib.nfopp_last_error()
    out = {k: engs[k].traj for k in libs}

if KERNEL == "k3":      # arc-length reparametrisation
    from nfopp.engine import TrajectoryEngine, TrajectoryHyper
    hyper = TrajectoryHyper(100, 5, 100, 0.1, 1e-3, 1, 10, 100, 5e-2, (0.9, 0.9), 1e-8, (0, 100, 0, 100))
    engs = {}
    walk = torch.cumsum(torch.rand(B, N, 3, device="cuda") * torch.tensor([0.5, 0.5, 0.05], device="cuda"), dim=1)
    for k in libs:
        e = TrajectoryEngine(onf, B, N, 3, hyper, 0.5, "cuda")
        e.traj.copy_(walk)
        e.set_endpoints(walk[:, 0].cpu().numpy(), walk[:, -1].cpu().numpy())
        engs[k] = e

    def run(lib, o, _libs=libs, _engs=engs):   # noqa: F811
        e = _engs["product" if lib is _libs["product"] else "variant"]
        rc = lib.nfopp_reparametrize(e.B, e.N, e.D, _lib.ptr(e.traj), _lib.ptr(e.start), _lib.ptr(e.goal), _lib.ptr(e.lam),
                                     _lib.ptr(e.cm), _lib.ptr(e.u), None, _lib.stream_ptr())
        assert rc == 0, lib.nfopp_last_error()
    out = {k: engs[k].traj for k in libs}

times = {k: [] for k in libs}
for rnd in range(8):
    for k, lib in libs.items():
        for _ in range(3):
            run(lib, out[k])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run(lib, out[k])
        b.record()
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 20)
for k, v in times.items():
    print("%-8s median %.4f ms  min %.4f  (rounds: %s)" % (k, float(np.median(v)), min(v), " ".join("%.3f" % x for x in v)))
print("variant / product (median): %.4f" % (np.median(times["variant"]) / np.median(times["product"])))
print("outputs identical:", bool(torch.equal(out["product"], out["variant"])))
if KERNEL == "k2":
    for name in ("lam", "cm", "adam_m", "adam_v"):
        print("  %s identical: %s" % (name, bool(torch.equal(getattr(engs["product"], name), getattr(engs["variant"], name)))))
if KERNEL == "k1" and not torch.equal(out["product"], out["variant"]):
    d = (out["product"] - out["variant"]).abs().reshape(-1, 4)
    scale = out["product"].abs().reshape(-1, 4).amax(0)
    print("max |difference| per column (logit, gx, gy, gth):", [float("%.3g" % x) for x in d.amax(0).tolist()],
          " column scale:", [float("%.3g" % x) for x in scale.tolist()],
          " rows that differ:", int((d.amax(1) > 0).sum()), "of", d.shape[0])
