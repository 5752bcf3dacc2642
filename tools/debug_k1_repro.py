"""Dev tool: is the fused ONF kernel bitwise reproducible?  Repeats one launch on identical inputs (trajectory mode with a
fixed t, explicit-pose mode, forward-only mode) and counts the runs that differ from the first."""
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
RUNS = int(os.environ.get("RUNS", "200"))
torch.manual_seed(0)
onf = nfopp.ONF(0.0, 10.0, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
lib = _lib.load()
for path in (1, 0):
    _lib.check(lib.nfopp_set_matrix_path(path))
    for P in (4099, 1044480):
        x = torch.rand(P, 3, device="cuda") * torch.tensor([100.0, 100.0, 6.28], device="cuda")
        for name, fn in (("grad", onf.forward_with_grad), ("forward", onf.forward)):
            first = fn(x).clone()
            bad = 0
            worst = 0.0
            for _ in range(RUNS):
                o = fn(x)
                if not torch.equal(o, first):
                    bad += 1
                    worst = max(worst, float((o - first).abs().max()))
            print("path %s  P=%-8d %-8s: %d of %d repeats differ from the first (max |d| %.3e)" % (
                "split" if path else "fp32", P, name, bad, RUNS, worst))
