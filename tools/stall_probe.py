"""Dev probe: are the occasional ~70 ms stalls seen in the B = 1 latency probes a property of the platform or of this library?
(a) a stream of trivial torch kernels, (b) BatchPlanner.step(n=10) at B = 1, each for ~3 s, timed in blocks with HIP events and
the host clock; prints when the slow blocks happened."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-motion-planner_amd"))
import nfopp


def run(name, body, seconds=3.0):
    for _ in range(20): body()
    torch.cuda.synchronize()
    rows, t_all = [], time.perf_counter()
    while time.perf_counter() - t_all < seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); body(); e1.record(); torch.cuda.synchronize()
        rows.append((e0.elapsed_time(e1), (time.perf_counter() - t0) * 1e3, time.perf_counter() - t_all))
    ev = sorted(r[0] for r in rows)
    slow = [(round(a, 1), round(b, 1), round(c, 2)) for a, b, c in rows if a > 10 * ev[len(ev) // 2] and a > 5]
    print("%s: %d blocks, event ms median %.3f p99 %.3f max %.3f; slow blocks (event ms, wall ms, at s): %s"
          % (name, len(rows), ev[len(ev) // 2], ev[int(0.99 * len(ev))], ev[-1], slow[:12]), flush=True)


x = torch.zeros(1024, device="cuda")
def trivial():
    for _ in range(100): x.add_(1.0)
run("100 trivial torch kernels per block", trivial)
big = torch.zeros(64 << 20, device="cuda")
def heavy():
    for _ in range(20): big.add_(1.0)
run("20 x 256 MB torch kernels per block", heavy)
torch.random.manual_seed(3)
onf = nfopp.ONF(0, 1, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
bounds = (-0.1, 3.1, -0.1, 3.1)
hyper = nfopp.TrajectoryHyper(collision_weight=3, direction_delta_weight=7, collision_beta=2, bounds=bounds)
for B in (1, 4096):
    bp = nfopp.BatchPlanner(onf, B, 256, hyper, device="cuda", seed=1)
    rng = np.random.default_rng(0)
    st = np.concatenate([rng.uniform(0.2, 0.8, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(np.float32)
    go = np.concatenate([rng.uniform(2.2, 2.8, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(np.float32)
    bp.init(st, go, bounds)
    run("BatchPlanner B=%d step(n=10) per block" % B, lambda: bp.step(n=10))
# the same B = 1 loop with Python's cyclic garbage collector out of the way: is the one slow block a full (generation-2) collection?
import gc
bp = nfopp.BatchPlanner(onf, 1, 256, hyper, device="cuda", seed=1)
bp.init(st[:1], go[:1], bounds)
gc.collect(); gc.freeze(); gc.disable()
run("BatchPlanner B=1 step(n=10) per block, gc frozen + disabled", lambda: bp.step(n=10))
gc.enable()
