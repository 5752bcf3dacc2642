"""GPU dev tool: per-phase clock shares of the 32x32x16 ONF kernel on the cfg3 shape.
Build the variant first:  make -C pytorch-motion-planner_amd/csrc OBJ=$PWD/build/csrc_prof OUT=$PWD/build/prof EXTRA=-DX32_PHASE_PROFILE
The variant library prints the shares to stderr every tenth launch."""
import ctypes, os, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-motion-planner_amd"))
import nfopp
from nfopp import _lib

lib = ctypes.CDLL(os.path.join(ROOT, "build", "prof", "libnfopp_hip.so"))
for name, (res, args) in _lib._SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
torch.manual_seed(0)
onf = nfopp.ONF(0.0, 10.0, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
B, N = 4096, 256
traj = torch.rand(B, N, 3, device="cuda") * torch.tensor([100.0, 100.0, 6.0], device="cuda")
t = torch.zeros(B, N - 1, device="cuda")
out = torch.zeros(B, N - 1, 4, device="cuda")
cfg = onf.config_c()
for _ in range(20):
    rc = lib.nfopp_traj_collision_eval(cfg, _lib.ptr(onf.flat_parameters), _lib.ptr(traj), B, N, 3, _lib.ptr(t), 1, 7, 0, 0,
                                       _lib.ptr(out), None, None, _lib.stream_ptr())
    assert rc == 0, lib.nfopp_last_error()
torch.cuda.synchronize()
