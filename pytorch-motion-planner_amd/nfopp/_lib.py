"""ctypes binding of libnfopp_hip.so (C ABI: include/nfopp_hip.h).

The HIP library is the ONLY compute path of this package.  If it is missing, or no MI355X is visible when a
kernel is requested, calls raise `NfoppError` -- there is no CPU fallback by design.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnfopp_hip.so")
ABI_VERSION = 6
NUM_TERMS = 8
TERM_NAMES = ("total", "distance", "softplus_sum", "lambda_dot_c", "c_squared", "boundary", "cm_tanh", "direction")


class NfoppError(RuntimeError):
    pass


class OnfConfigC(ctypes.Structure):
    _fields_ = [("mean", ctypes.c_float), ("sigma", ctypes.c_float), ("use_cos", ctypes.c_int32),
                ("has_bias", ctypes.c_int32), ("angle_dim", ctypes.c_int32)]


class TrajHyperC(ctypes.Structure):
    _fields_ = [("collision_weight", ctypes.c_float), ("angle_weight", ctypes.c_float),
                ("constraint_deltas_weight", ctypes.c_float), ("multipliers_lr", ctypes.c_float),
                ("collision_multipliers_lr", ctypes.c_float), ("boundary_weight", ctypes.c_float),
                ("collision_beta", ctypes.c_float), ("direction_delta_weight", ctypes.c_float),
                ("bounds", ctypes.c_float * 4),
                ("adam_beta2", ctypes.c_float), ("adam_omb1", ctypes.c_float), ("adam_omb2", ctypes.c_float),
                ("adam_eps", ctypes.c_float), ("adam_step_size", ctypes.c_float), ("adam_bc2_sqrt", ctypes.c_float)]


_P = ctypes.c_void_p


class TrajBuffersC(ctypes.Structure):
    """nfopp_traj_buffers (include/nfopp_hip.h): the device state of one batch, borrowed for nfopp_traj_steps."""
    _fields_ = [("traj_dev", _P), ("start_dev", _P), ("goal_dev", _P), ("lam_dev", _P), ("cm_dev", _P), ("adam_m_dev", _P),
                ("adam_v_dev", _P), ("t_dev", _P), ("onf_out4_dev", _P), ("hinv_band_dev", _P), ("u_dev", _P),
                ("active_dev", _P), ("live_ws_dev", _P), ("batch", ctypes.c_int64), ("n_waypoints", ctypes.c_int32),
                ("dim", ctypes.c_int32), ("half_width", ctypes.c_int32), ("interior_lo", ctypes.c_int32),
                ("interior_hi", ctypes.c_int32)]


class StepScheduleC(ctypes.Structure):
    """nfopp_step_schedule: Adam group + counters + draw stream of an n-step call."""
    _fields_ = [("adam_lr", ctypes.c_double), ("adam_beta1", ctypes.c_double), ("adam_beta2", ctypes.c_double),
                ("adam_steps_done", ctypes.c_int64), ("step_count", ctypes.c_int64), ("traj_index_offset", ctypes.c_int64),
                ("seed", ctypes.c_uint64), ("rng_offset", ctypes.c_uint64), ("reparam_freq", ctypes.c_int32),
                ("t_mode", ctypes.c_int32)]


_SIGNATURES = {
    "nfopp_abi_version": (ctypes.c_int, []),
    "nfopp_last_error": (ctypes.c_char_p, []),
    "nfopp_device_count": (ctypes.c_int, []),
    "nfopp_onf_param_count": (ctypes.c_int64, [ctypes.POINTER(OnfConfigC)]),
    "nfopp_onf_eval_points": (ctypes.c_int, [ctypes.POINTER(OnfConfigC), _P, _P, ctypes.c_int64, _P, _P]),
    "nfopp_onf_eval_logits": (ctypes.c_int, [ctypes.POINTER(OnfConfigC), _P, _P, ctypes.c_int64, _P, _P]),
    "nfopp_traj_collision_eval": (ctypes.c_int, [ctypes.POINTER(OnfConfigC), _P, _P, ctypes.c_int64, ctypes.c_int32,
                                                 ctypes.c_int32, _P, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64,
                                                 ctypes.c_int64, _P, _P, _P, _P]),
    "nfopp_traj_update": (ctypes.c_int, [ctypes.POINTER(TrajHyperC), ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                         _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_int32, _P, _P, _P]),
    "nfopp_traj_steps": (ctypes.c_int, [ctypes.POINTER(OnfConfigC), _P, ctypes.POINTER(TrajHyperC), ctypes.POINTER(TrajBuffersC),
                                        ctypes.POINTER(StepScheduleC), ctypes.c_int32, _P, _P, _P]),
    "nfopp_reparametrize": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nfopp_path_interpolate": (ctypes.c_int, [_P, _P, _P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                              _P, _P, _P]),
    "nfopp_path_select_best": (ctypes.c_int, [_P, _P, _P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                              _P, _P, _P, _P, _P]),
    "nfopp_set_matrix_path": (ctypes.c_int, [ctypes.c_int32]),
    "nfopp_get_matrix_path": (ctypes.c_int, []),
    "nfopp_onf_params_version": (ctypes.c_int, [_P, ctypes.c_uint64]),
    "nfopp_init_trajectories": (ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                               _P, _P]),
    "nfopp_path_postprocess": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                              ctypes.c_int32, _P, _P, _P]),
    "nfopp_onf_train_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(OnfConfigC), ctypes.c_int64]),
    "nfopp_onf_train_grad": (ctypes.c_int, [ctypes.POINTER(OnfConfigC), _P, _P, _P, ctypes.c_int64, ctypes.c_float,
                                            _P, _P, ctypes.c_size_t, _P]),
    "nfopp_onf_train_grad_ex": (ctypes.c_int, [ctypes.POINTER(OnfConfigC), _P, _P, _P, ctypes.c_int64, ctypes.c_float,
                                               _P, _P, ctypes.c_size_t, ctypes.c_int32, _P]),
    "nfopp_check_collision_circle": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int32, _P, ctypes.c_int32, ctypes.c_float,
                                                    ctypes.POINTER(ctypes.c_float), _P, _P]),
    "nfopp_check_collision_circle_cells": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int32, _P, ctypes.c_int32, _P,
                                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                                          ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_float), _P, _P]),
    "nfopp_check_collision_rectangle": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int32,
                                                       ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _P, _P]),
    "nfopp_check_collision_grid": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int32, _P, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_double, ctypes.c_double, ctypes.c_double, _P, _P]),
    "nfopp_sample_candidates": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                               ctypes.POINTER(ctypes.c_float), ctypes.c_uint64, ctypes.c_uint64,
                                               ctypes.c_int64, _P, _P, _P, _P, _P, _P]),
    "nfopp_resample_pool": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, _P, _P, _P, _P,
                                           _P, _P, _P]),
    "nfopp_adam_step": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_float, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load the shared library (idempotent).  Loading needs no GPU; launching kernels does."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NfoppError("libnfopp_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "or `make -C pytorch-motion-planner_amd/csrc`; there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.nfopp_abi_version() != ABI_VERSION:
        raise NfoppError("libnfopp_hip.so ABI %d != binding ABI %d" % (lib.nfopp_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise NfoppError("nfopp call failed (%d): %s" % (status, load().nfopp_last_error().decode(errors="replace")))


def require_gpu():
    lib = load()
    if not torch.cuda.is_available() or lib.nfopp_device_count() < 1:
        raise NfoppError("no HIP device visible: the NFOPP hot path runs on MI355X only (no CPU fallback)")


def require_current_device(tensor_index, current_index):
    """Every launch takes its stream, its CU count and its scratch blobs from the CURRENT device, so a buffer that
    lives on another GPU would be handed to a kernel queued on the wrong card (a memory fault without peer access,
    silent cross-device traffic with it).  One process per GPU is the design; a process that drives several GPUs
    must make the buffer's device current (`torch.cuda.set_device` / `with torch.cuda.device(...)`) around the call."""
    if tensor_index != current_index:
        raise NfoppError("buffer lives on cuda:%d but the current device is cuda:%d: make the planner's device "
                         "current (torch.cuda.set_device) before calling into the HIP library"
                         % (tensor_index, current_index))


# torch.cuda.current_device() / current_stream() cost 2-9 us per call in Python; a planner step makes ~30 of them, which at
# B = 1 was a quarter of the step's wall time (tools/b1_breakdown.py).  The raw accessors below are what they wrap.
_raw_device = getattr(torch._C, "_cuda_getDevice", None)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _current_device():
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA(HIP) tensor on the current device, or NULL for None."""
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise NfoppError("expected a contiguous %s HIP tensor, got %s" % (dtype, _describe(t)))
    dev = _current_device()
    if t.device.index != dev:
        require_current_device(t.device.index, dev)
    return t.data_ptr() or None   # empty tensors have no storage: pass NULL (the C side accepts it for size 0)


def _describe(t):
    if isinstance(t, torch.Tensor):
        return "tensor(device=%s, dtype=%s, contiguous=%s, shape=%s)" % (t.device, t.dtype, t.is_contiguous(), tuple(t.shape))
    return repr(type(t))


def stream_ptr():
    """hipStream_t of the CURRENT stream of the current device, as an integer."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream
