"""torch.ops.nfopp.* -- the PyTorch-ROCm extension form of the boundary (csrc/torch_ops.cpp over the C ABI).

`load()` registers the ops with torch.ops.load_library; they take tensors, check device / dtype / contiguity / shapes
with TORCH_CHECK and launch on the current HIP stream of the tensors' device.  The ctypes binding (nfopp/_lib.py) stays
the package's own path to the same C ABI; both reach the same kernels.  No CPU path: CPU tensors raise."""
import os

import torch

from . import _lib

TORCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libnfopp_torch.so")
OPS = ("onf_fwd_bwd_input", "onf_logits", "traj_step", "traj_steps", "reparametrize", "onf_train_grad", "adam_step", "onf_train_step")
_loaded = False


def load():
    """Idempotent.  Needs no GPU (registration only); the ops themselves do."""
    global _loaded
    if not _loaded:
        if not os.path.exists(TORCH_LIB_PATH):
            raise _lib.NfoppError("libnfopp_torch.so is not built (%s): run `make -C pytorch-motion-planner_amd/csrc`"
                                  % TORCH_LIB_PATH)
        torch.ops.load_library(TORCH_LIB_PATH)
        _loaded = True
    return torch.ops.nfopp


def hyper_list(hyper_c):
    """The 18 floats of an `_lib.TrajHyperC` in declaration order (the `hyper` argument of torch.ops.nfopp.traj_step)."""
    out = []
    for name, ctype in hyper_c._fields_:
        v = getattr(hyper_c, name)
        out.extend(list(v) if hasattr(v, "__len__") else [v])
    return [float(x) for x in out]
