"""Occupancy neural field (ONF) host object for the HIP path.

Mirrors the reference's `ONF` / `AngleEncoder` modules (nfop/onf_model.py:7-50, nfop/angle_encoder.py:6-22):
same constructor arguments, same `state_dict()` keys and shapes, same initialisation calls in the same order
(so a given torch seed yields the same initial field).  All parameters are VIEWS of one flat fp32 buffer laid out
in `state_dict()` order -- the buffer the HIP kernels read directly (include/nfopp_hip.h, nfopp_onf_config).
`forward` evaluates the field with the fused HIP kernel; there is no PyTorch arithmetic on this path.
"""
import itertools
import math

import torch
from torch import nn

from . import _lib

HIDDEN = 100
_CONTENT_VERSIONS = itertools.count(1)   # process-wide: a version names ONE content of ONE buffer, never reused


class _Leaf(nn.Module):
    """Parameter holder whose attribute names reproduce the reference's state_dict keys."""

    def __init__(self, **params):
        super().__init__()
        for k, v in params.items():
            self.register_parameter(k, v)


class ONF(nn.Module):
    def __init__(self, mean, sigma, use_cos=False, use_normal_init=False, bias=True, angle_encoding=False):
        super().__init__()
        self._mean, self._sigma = float(mean), float(sigma)
        self._use_cos, self._bias = bool(use_cos), bool(bias)
        self._angle_dim = 10 if angle_encoding else 0
        n_enc = 200 if use_cos else 100
        feature_dim = n_enc + 2 * self._angle_dim
        self.feature_dim, self.n_enc = feature_dim, n_enc
        self.point_dim = 3 if angle_encoding else 2

        # same creation order and init calls as the reference => same RNG consumption
        pieces = []
        if angle_encoding:
            biases = torch.zeros(2 * self._angle_dim)
            freq = torch.linspace(1, self._angle_dim, self._angle_dim)
            nn.init.uniform_(biases, -math.pi, math.pi)
            pieces += [("_angle_encoder._biases", biases), ("_angle_encoder._frequencies", torch.cat([freq, freq]))]
        l1, l2 = nn.Linear(feature_dim, HIDDEN), nn.Linear(HIDDEN, HIDDEN)
        l3 = nn.Linear(HIDDEN + feature_dim, 1)
        enc = nn.Linear(2, n_enc, bias=bias)
        if use_normal_init:
            nn.init.normal_(enc.weight)
        pieces += [("mlp.0.weight", l1.weight.data), ("mlp.0.bias", l1.bias.data), ("mlp.2.weight", l2.weight.data),
                   ("mlp.2.bias", l2.bias.data), ("mlp2.0.weight", l3.weight.data), ("mlp2.0.bias", l3.bias.data),
                   ("encoding_layer.weight", enc.weight.data)]
        if bias:
            pieces.append(("encoding_layer.bias", enc.bias.data))
        self._layout = [(k, tuple(v.shape)) for k, v in pieces]
        flat = torch.cat([v.reshape(-1).to(torch.float32) for _, v in pieces])
        self._bind(flat)

    # ---- flat buffer <-> named views ----------------------------------------------------------------------------
    def _bind(self, flat):
        self._withdraw_version()
        object.__setattr__(self, "_flat", flat.contiguous())
        object.__setattr__(self, "_vouched", None)   # (data_ptr, torch version counter) the registered version stands for
        object.__setattr__(self, "_registered", None)   # (device, data_ptr) the library holds a version for, if any
        if not hasattr(self, "_frozen"):
            object.__setattr__(self, "_frozen", False)
        views, o = {}, 0
        for name, shape in self._layout:
            n = 1
            for s in shape:
                n *= s
            views[name] = nn.Parameter(self._flat[o:o + n].view(shape), requires_grad=False)
            o += n
        assert o == self._flat.numel()
        for key in ("_angle_encoder", "mlp", "mlp2", "encoding_layer"):
            if key in self._modules:
                del self._modules[key]
        if self._angle_dim:
            self._angle_encoder = _Leaf(_biases=views["_angle_encoder._biases"],
                                        _frequencies=views["_angle_encoder._frequencies"])
        self.mlp = nn.Module()
        self.mlp.add_module("0", _Leaf(weight=views["mlp.0.weight"], bias=views["mlp.0.bias"]))
        self.mlp.add_module("2", _Leaf(weight=views["mlp.2.weight"], bias=views["mlp.2.bias"]))
        self.mlp2 = nn.Module()
        self.mlp2.add_module("0", _Leaf(weight=views["mlp2.0.weight"], bias=views["mlp2.0.bias"]))
        enc = {"weight": views["encoding_layer.weight"]}
        if self._bias:
            enc["bias"] = views["encoding_layer.bias"]
        self.encoding_layer = _Leaf(**enc)

    def _apply(self, fn, *args, **kwargs):
        # `.to(device)` / `.cuda()` must keep every parameter a view of ONE flat buffer
        self._bind(fn(self._flat.detach()))
        return self

    def load_flat(self, flat):
        """Overwrite all parameters from a flat fp32 vector in state_dict order."""
        flat = torch.as_tensor(flat, dtype=torch.float32).reshape(-1)
        if flat.numel() != self._flat.numel():
            raise ValueError("expected %d parameters, got %d" % (self._flat.numel(), flat.numel()))
        self._flat.copy_(flat.to(self._flat.device))

    @property
    def flat_parameters(self):
        return self._flat

    @property
    def n_params(self):
        return self._flat.numel()

    def config_c(self):
        """Configuration block of the C ABI.  Every launch that reads the parameters fetches it first, so this is also where
        the buffer's content version is registered with -- or withdrawn from -- the library (nfopp_onf_params_version).
        DEFAULT: no version, the split kernels rebuild their pre-split weight image in front of every launch (5 us), so a
        write to the parameters by ANY means (in-place ops, `.data`, `dist.broadcast`, a foreign kernel) shows in the very
        next evaluation.  Image reuse is OPT-IN: see `freeze()`."""
        self._vouch()
        c = self.__dict__.get("_cfg_block")
        if c is None:
            c = _lib.OnfConfigC(self._mean, self._sigma, int(self._use_cos), int(self._bias), self._angle_dim)
            object.__setattr__(self, "_cfg_block", c)
        return c

    # ---- weight-image reuse (ABI 5 content versions): opt-in ------------------------------------------------------
    def freeze(self):
        """The caller takes responsibility for telling this object about parameter writes: from now on launches reuse the
        stream's pre-split weight image (no prep launch) until `mark_modified()` / `unfreeze()` is called.  In-place
        torch ops on the flat buffer or on a parameter view are still noticed by themselves (torch's version counter); writes
        the counter does NOT see -- `torch.distributed.broadcast` / `all_reduce` into the buffer, `.data` assignments,
        raw-pointer kernels -- need `mark_modified()`.  The planners that own the field's update loop (`BatchPlanner`, the
        drop-in planners inside `step(n)`) freeze the field themselves and call `mark_modified()` after their own Adam
        steps.  Returns self."""
        object.__setattr__(self, "_frozen", True)
        return self

    def unfreeze(self):
        """Back to the default: the weight image is rebuilt in front of every launch."""
        object.__setattr__(self, "_frozen", False)
        self._withdraw_version()
        object.__setattr__(self, "_vouched", None)
        return self

    @property
    def is_frozen(self):
        return self._frozen

    def frozen(self):
        """Context manager: `with onf.frozen(): ...` = freeze() for the block, the previous state afterwards."""
        onf = self

        class _Scope(object):
            def __enter__(self_inner):
                self_inner.was = onf._frozen
                onf.freeze()
                return onf

            def __exit__(self_inner, *exc):
                if not self_inner.was:
                    onf.unfreeze()
                return False
        return _Scope()

    def _torch_version(self):
        """torch's in-place version counter of the flat buffer, None where torch keeps none (inference tensors: an ONF built,
        moved or re-bound under `torch.inference_mode()`); then only `mark_modified()` announces a write."""
        try:
            return self._flat._version
        except RuntimeError:
            return None

    def _vouch(self):
        f = self._flat
        if not f.is_cuda:
            return
        if not self._frozen:
            self._withdraw_version()          # a version left over from a frozen phase
            object.__setattr__(self, "_vouched", None)
            return
        key = (f.data_ptr(), self._torch_version())
        if self._vouched != key:
            lib = _lib.load()
            with torch.cuda.device(f.device):
                _lib.check(lib.nfopp_onf_params_version(f.data_ptr(), next(_CONTENT_VERSIONS)))
            object.__setattr__(self, "_vouched", key)
            object.__setattr__(self, "_registered", (f.device, f.data_ptr()))

    def mark_modified(self):
        """Call after writing the parameters of a FROZEN field in a way torch's version counter does not see (a raw-pointer
        kernel such as nfopp_adam_step, `.data` writes, collectives into the buffer): the next launch registers a new content
        version and rebuilds the image.  No-op for an unfrozen field (it rebuilds anyway)."""
        object.__setattr__(self, "_vouched", None)

    def _withdraw_version(self):
        reg = getattr(self, "_registered", None)
        if reg is not None:
            try:
                with torch.cuda.device(reg[0]):
                    _lib.load().nfopp_onf_params_version(reg[1], 0)
            except Exception:
                pass
            object.__setattr__(self, "_registered", None)

    def __del__(self):
        self._withdraw_version()

    # ---- evaluation (HIP only) ------------------------------------------------------------------------------------
    def _eval(self, x, with_grad):
        _lib.require_gpu()
        if not self._flat.is_cuda:
            raise _lib.NfoppError("ONF parameters live on %s; move the model to the HIP device first" % self._flat.device)
        x = torch.as_tensor(x, dtype=torch.float32, device=self._flat.device).contiguous()
        if x.dim() != 2 or x.shape[1] != self.point_dim:
            raise ValueError("expected points of shape [P, %d], got %s" % (self.point_dim, tuple(x.shape)))
        out = torch.empty(x.shape[0], 4, dtype=torch.float32, device=x.device)
        if x.shape[0]:
            lib = _lib.load()
            fn = lib.nfopp_onf_eval_points if with_grad else lib.nfopp_onf_eval_logits
            _lib.check(fn(self.config_c(), _lib.ptr(self._flat), _lib.ptr(x), x.shape[0], _lib.ptr(out), _lib.stream_ptr()))
        return out

    def forward_with_grad(self, x):
        """x [P, point_dim] on the HIP device -> out4 [P, 4] = logit, dlogit/dx, dlogit/dy, dlogit/dtheta."""
        return self._eval(x, True)

    def forward(self, x):
        """Logits [P, 1] like the reference module (nfop/onf_model.py:33-50); forward-only kernel."""
        return self._eval(x, False)[:, :1]
