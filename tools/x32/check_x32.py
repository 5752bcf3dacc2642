"""GPU: the 32x32x16 ONF kernel (matrix path 1) against the fp32-MFMA path (0) and the 16x16x32 split kernel (2) on the
golden networks, then a timing of the three split selections at 1 M points.  Development tool (run through gpurun)."""
import sys, time
import numpy as np
import torch

sys.path.insert(0, "tests"); sys.path.insert(0, "pytorch-motion-planner_amd"); sys.path.insert(0, ".")
import gpu_common as gc
from nfopp import _lib

F32 = np.float32
lib = _lib.load()


def ev(onf, x, path):
    _lib.check(lib.nfopp_set_matrix_path(path))
    xt = torch.tensor(np.ascontiguousarray(x, F32), device="cuda")
    out = onf.forward_with_grad(xt)
    lg = onf(xt)
    torch.cuda.synchronize()
    return out.cpu().numpy(), lg.cpu().numpy().reshape(-1)


z = np.load("tests/golden/g1_onf.npz")
worst = 0.0
for tag in "abc":
    onf, cfg = gc.make_onf(z[tag + "_cfg"], z[tag + "_params"])
    x = z[tag + "_x"]
    d = x.shape[1]
    for n in (len(x), 1, 17, 255, 4099, 70000):
        rng = np.random.default_rng(n)
        xs = x if n == len(x) else x[rng.integers(0, len(x), n)]
        o0, l0 = ev(onf, xs, 0)
        o3, l3 = ev(onf, xs, 1)
        e = [gc.scaled_err(o3[:, 0], o0[:, 0]), gc.scaled_err(o3[:, 1:1 + d], o0[:, 1:1 + d]), gc.scaled_err(l3, l0)]
        worst = max(worst, *e)
        print(tag, n, "logit %.2e grad %.2e fwd-only %.2e" % tuple(e), flush=True)
        if e[1] > 1e-4 and n == 255:
            for c in range(1, 1 + d):
                bad = np.abs(o3[:, c] - o0[:, c]) > 1e-4 * np.abs(o0[:, c]).max()
                print("   col", c, "bad rows", int(bad.sum()), "first", np.flatnonzero(bad)[:8], o3[:4, c], o0[:4, c])
    o3, _ = ev(onf, x, 1)
    print(tag, "vs golden: logit %.2e grad %.2e" % (gc.scaled_err(o3[:, 0], z[tag + "_logit"]),
                                                  gc.scaled_err(o3[:, 1:1 + d], z[tag + "_grad"])), flush=True)
print("worst", worst)

onf, cfg = gc.make_onf(z["a_cfg"], z["a_params"])
x = z["a_x"]
xs = torch.tensor(x[np.random.default_rng(0).integers(0, len(x), 1044480)].astype(F32), device="cuda")
for path in (2, 1, 2, 1):
    _lib.check(lib.nfopp_set_matrix_path(path))
    for _ in range(5):
        onf.forward_with_grad(xs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        onf.forward_with_grad(xs)
    torch.cuda.synchronize()
    print("path", path, "%.3f ms per 1044480 points" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
