"""Dev tool: is the ONF fit gradient bitwise reproducible, and if not, in which parameter block do two runs differ?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-motion-planner_amd"))
import nfopp  # noqa: E402
from nfopp import _lib  # noqa: E402

if os.environ.get("NFOPP_DEV_LIB"):
    _lib.LIB_PATH = os.environ["NFOPP_DEV_LIB"]
torch.manual_seed(0)
onf = nfopp.ONF(0.0, 10.0, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
if os.environ.get("GOLDEN_FIELD"):      # the fitted sigma = 10 field of the bench-mr fixtures instead of a random one
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import gpu_common as gc
    z = np.load(os.path.join(ROOT, "tests", "golden", "traj_benchmr_n512.npz"))
    onf, _ = gc.make_onf(z["cfg"], z["params"])
g = onf.geometry() if hasattr(onf, "geometry") else None
fit = nfopp.OnfFitter(onf, 2e-2, (0.9, 0.9), distributed=False)
RUNS = int(os.environ.get("RUNS", "10"))
for P in (65536, 2543616, 2543616):
    x = torch.rand(P, 3, device="cuda") * torch.tensor([100.0, 100.0, 6.28], device="cuda")
    y = (torch.rand(P, device="cuda") < 0.4).float()
    runs = []
    for _ in range(RUNS):
        fit._hip_grad(x, y, 1.0 / P)
        torch.cuda.synchronize()
        runs.append(fit.grad.clone().cpu().numpy())
    bad_runs = sum(int(not np.array_equal(runs[0], r)) for r in runs[1:])
    d01 = np.zeros(runs[0].shape, bool)
    for r in runs[1:]:
        d01 |= runs[0] != r
    idx = np.nonzero(d01)[0]
    print("P=%d: %d of %d repeats differ from the first; elements ever differing: %d of %d" % (P, bad_runs, RUNS - 1, len(idx), runs[0].size),
          "" if len(idx) == 0 else "first %s last %s max |d| %.3e (scale %.3e)" % (
              idx[:8].tolist(), idx[-4:].tolist(), float(max(np.abs(runs[0] - r).max() for r in runs[1:])), float(np.abs(runs[0][:-2]).max())))
if g is None:
    names = [(n, p.numel()) for n, p in onf.named_parameters()]
    print("parameter blocks (flat order may differ):", names)
