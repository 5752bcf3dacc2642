"""GPU: the "nccl" (= RCCL) branch of the data-parallel ONF fit, executed for real on the one card this pool hands out.
RCCL refuses two ranks on one device, so the process group has ONE rank: `init_process_group("nccl", device_id=...)`, the
all-reduce of the flat [n_params | loss | count] device buffer inside `OnfFitter.step`, the count all-reduce of
`global_count()` and bench.py's `ranks_seen` all run through librccl; with one rank the sums are identities, so the result
must equal the non-distributed fit bit for bit.  (What N > 1 adds -- the cross-rank sum -- is covered by the gloo world-size-2
test on CPU, tests/test_distributed_cpu.py.)  Runs in a child process: process-group state stays out of the test session."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent("""
    import os, sys, socket
    import numpy as np, torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(%(root)r, "pytorch-motion-planner_amd")); sys.path.insert(0, %(root)r)
    import nfopp, bench
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    assert bench.ranks_seen(2, "nccl", dev) == 1          # the all-reduce of ones bench.py prints (world > 1 branch, one rank)
    rng = np.random.default_rng(5)
    P = 6000
    x = torch.tensor(np.concatenate([rng.uniform(0, 100, (P, 2)), rng.uniform(0, 6.28, (P, 1))], 1).astype(np.float32), device=dev)
    y = torch.tensor((rng.uniform(size=P) < 0.3).astype(np.float32), device=dev)
    out = []
    for distributed in (True, False):
        torch.random.manual_seed(11)
        onf = nfopp.ONF(0, 10, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to(dev)
        fitter = nfopp.OnfFitter(onf, lr=2e-2, betas=(0.9, 0.9), distributed=distributed)
        assert fitter._in_group() == distributed
        for k in range(3):
            loss = fitter.step(x, y, global_count=None if k == 0 else P)   # k = 0: the count all-reduce too
        torch.cuda.synchronize()
        out.append((onf.flat_parameters.cpu().numpy().copy(), fitter.grad.cpu().numpy().copy(), float(loss)))
    dist.barrier()
    dist.destroy_process_group()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    assert np.isfinite(out[0][0]).all() and out[0][1][-1] == P
    print("RCCL_SINGLE_RANK_OK")
""")


@pytest.mark.timeout(600)
def test_onf_fit_through_a_one_rank_rccl_group_equals_the_local_fit():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=580)
    assert res.returncode == 0 and "RCCL_SINGLE_RANK_OK" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
