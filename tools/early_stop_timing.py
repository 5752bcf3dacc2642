"""GPU tool: time of the fused ONF kernel against the live fraction of an `active` mask (early stop,
scripts/run_bench_mr.py:121-126).  The bit-identity half of this property is tests/test_gpu_benchmr.py::
test_retired_trajectories_leave_the_onf_kernel; timing is kept out of the parity suite."""
import sys
import numpy as np
import torch

sys.path.insert(0, "tests"); sys.path.insert(0, "pytorch-motion-planner_amd"); sys.path.insert(0, ".")
import gpu_common as gc
from oracle import nfopp_oracle as orc

F32 = np.float32
z = np.load("tests/golden/traj_benchmr_n256.npz")
onf, cfg = gc.make_onf(z["cfg"], z["params"])
hp = orc.Hyper.from_npz(z)
B = 4096
s = gc.state_of(z, "s0_", reps=B)


def kernel_ms(e, reps=20):
    for _ in range(3):
        e.collision_eval()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        e.collision_eval()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rng = np.random.default_rng(8)
for frac in (1.0, 0.75, 0.5, 0.25, 0.05):
    eng = gc.engine_from_state(onf, s, hp)
    eng.seed = 77
    if frac < 1.0:
        eng.active = torch.tensor((rng.uniform(size=B) < frac).astype(np.uint8), device="cuda")
    print("live fraction %.2f: %.3f ms per collision_eval" % (frac, min(kernel_ms(eng) for _ in range(3))), flush=True)
