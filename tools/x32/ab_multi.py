"""GPU dev tool: A/B several builds of libnfopp_hip.so IN ONE PROCESS, interleaved rounds (guide rule 24), on the fused
ONF kernel at the cfg3 shape (4096 x 256, device Philox draws).
Usage: python tools/x32/ab_multi.py name=path[:matrix_path] ...     (the product build is always included as `product`;
`name=product:2` times the product library on matrix path 2)"""
import ctypes, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-motion-planner_amd"))
import nfopp
from nfopp import _lib


def bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib._SIGNATURES.items():
        if hasattr(lib, name):                  # (a variant built before an ABI addition lacks the new symbols)
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


variants = {"product": (bind(_lib.LIB_PATH), 1)}
for arg in sys.argv[1:]:
    name, spec = arg.split("=")
    path, _, mp = spec.partition(":")
    lib = variants["product"][0] if path == "product" else bind(os.path.abspath(path))
    variants[name] = (lib, int(mp) if mp else 1)
torch.manual_seed(0)
onf = nfopp.ONF(0.0, 10.0, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
B, N = int(os.environ.get("AB_B", "4096")), int(os.environ.get("AB_N", "256"))
traj = torch.rand(B, N, 3, device="cuda") * torch.tensor([100.0, 100.0, 6.0], device="cuda")
t = torch.zeros(B, N - 1, device="cuda")
out = {k: torch.zeros(B, N - 1, 4, device="cuda") for k in variants}
cfg = onf.config_c()
if os.environ.get("AB_WBITS"):    # weights truncated to this many significant bits (8: the hi level alone, 16: hi + mid; the
    bits = int(os.environ["AB_WBITS"])   # lower levels of the A operands are then all zero) -- which products cost the power
    with torch.no_grad():
        w = onf.flat_parameters.view(torch.int32)
        w &= ~((1 << (24 - bits)) - 1)
if os.environ.get("AB_ZERO"):      # data-dependence of the time (power / clocks): all-zero weights, same instruction stream
    with torch.no_grad():
        onf.flat_parameters.zero_()


POINTS = os.environ.get("AB_MODE") == "points"      # explicit poses (nfopp_onf_eval_points) instead of trajectory sampling
pts = traj[:, :-1].reshape(-1, 3).contiguous()


def run(k):
    lib, mp = variants[k]
    lib.nfopp_set_matrix_path(mp)
    if os.environ.get("AB_MODE") == "logits":        # forward only (nfopp_onf_eval_logits)
        rc = lib.nfopp_onf_eval_logits(cfg, _lib.ptr(onf.flat_parameters), _lib.ptr(pts), pts.shape[0], _lib.ptr(out[k]), _lib.stream_ptr())
        assert rc == 0, lib.nfopp_last_error()
        return
    if POINTS:
        rc = lib.nfopp_onf_eval_points(cfg, _lib.ptr(onf.flat_parameters), _lib.ptr(pts), pts.shape[0], _lib.ptr(out[k]), _lib.stream_ptr())
        assert rc == 0, lib.nfopp_last_error()
        return
    rc = lib.nfopp_traj_collision_eval(cfg, _lib.ptr(onf.flat_parameters), _lib.ptr(traj), B, N, 3, _lib.ptr(t), 1, 7, 0, 0,
                                       _lib.ptr(out[k]), None, None, _lib.stream_ptr())
    assert rc == 0, lib.nfopp_last_error()


times = {k: [] for k in variants}
for rnd in range(8):
    for k in variants:
        for _ in range(3):
            run(k)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run(k)
        b.record()
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 20)
base = float(np.median(times["product"]))
for k, v in times.items():
    same = bool(torch.equal(out[k], out["product"]))
    print("%-12s median %.4f ms  min %.4f  x%.3f of product  outputs %s" % (k, float(np.median(v)), min(v), float(np.median(v)) / base,
                                                                         "identical" if same else "DIFFER"), flush=True)
