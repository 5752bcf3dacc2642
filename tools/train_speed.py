"""Dev measurement: ONF fitting-step gradient throughput (samples/s) of the MFMA path at BASELINE config-5 scale."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-motion-planner_amd")):
    sys.path.insert(0, p)
import nfopp
from nfopp import _lib
if os.environ.get("NFOPP_DEV_LIB"):      # A/B a development build of the library
    _lib.LIB_PATH = os.environ["NFOPP_DEV_LIB"]
torch.random.manual_seed(1)
onf = nfopp.ONF(0, 10, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
lib = nfopp.load_library()
for P in ((2543616,) if os.environ.get("NFOPP_DEV_LIB") else (262144, 1048576, 2543616)):
    x = torch.rand(P, 3, device="cuda") * torch.tensor([100.0, 100.0, 6.28], device="cuda")
    y = (torch.rand(P, device="cuda") < 0.3).float()
    c = onf.config_c()
    need = lib.nfopp_onf_train_workspace_bytes(c, P)
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device="cuda")
    grad = torch.zeros(onf.n_params + 2, device="cuda")
    def run():
        _lib.check(lib.nfopp_onf_train_grad_ex(c, _lib.ptr(onf.flat_parameters), _lib.ptr(x), _lib.ptr(y), P, 1.0 / P,
                                               _lib.ptr(grad), _lib.ptr(ws), ws.numel() * 4, 2, _lib.stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 10
    for _ in range(K): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("P=%8d: %.3f ms/fit-grad  %.3g samples/s  workspace %.2f GB  loss %.4f" % (P, dt * 1e3, P / dt, need / 1e9, float(grad[-2])))
