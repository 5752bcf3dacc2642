"""Planner construction with the reference's factory entry points.

`PlannerFactory.make_onf_planner / make_constrained_onf_planner` and `UniversalFactory` follow
nfop/planner_factory.py:11-77 and nfop/utils/universal_factory.py:9-44: classes are resolved by `name`, keyword
arguments the constructor does not accept are silently dropped, an unknown name raises KeyError.  The only
difference a caller sees is `device`: this path is HIP-only, so the default device is "cuda" and "cpu" is rejected.
"""
from inspect import signature

import torch

from .host_utils import AstarTrajectoryInitializer, AttributeDict, TrajectoryInitializer
from .onf_model import ONF
from .planner import ConstrainedNERFOptPlanner, NERFOptPlanner

PARAMETER_ATTRIBUTE_NAMES = ("parameters", "params", "param", "parameter")
CLASS_NAME_ATTRIBUTES = ("name", "type")

DEFAULT_PARAMETERS = AttributeDict(
    device="cuda",
    trajectory_length=100,
    collision_model=AttributeDict(mean=0, sigma=10, use_cos=True, bias=True, use_normal_init=True, name="ONF"),
    collision_optimizer=AttributeDict(lr=1e-2, betas=(0.9, 0.9)),
    trajectory_optimizer=AttributeDict(lr=1e-2, betas=(0.9, 0.9)),
    planner=AttributeDict(name="ConstrainedNERFOptPlanner", trajectory_random_offset=0.02, collision_weight=1,
                          velocity_hessian_weight=0.5, random_field_points=10, init_collision_iteration=0,
                          constraint_deltas_weight=0.2, multipliers_lr=0.001, init_collision_points=100,
                          reparametrize_trajectory_freq=10, optimize_collision_model_freq=1, angle_weight=0.5,
                          boundary_weight=1, collision_multipliers_lr=1e-3),
)


class UniversalFactory(object):
    def __init__(self, classes):
        self._classes = {c.__name__: c for c in classes}

    def make_from_parameters(self, parameters, **kwargs):
        if not isinstance(parameters, dict):
            return parameters
        ctor = None
        for attribute in CLASS_NAME_ATTRIBUTES:
            if attribute in parameters:
                try:
                    ctor = self._classes[parameters[attribute]]
                except KeyError:
                    raise KeyError("Unknown class %s" % parameters[attribute])
        if ctor is None:
            return parameters
        for key, value in parameters.items():
            kwargs[key] = self.make_from_parameters(value)
        accepted = signature(ctor).parameters.keys()
        for name in PARAMETER_ATTRIBUTE_NAMES:
            if name in accepted:
                kwargs[name] = parameters
        return ctor(**{k: v for k, v in kwargs.items() if k in accepted})


def _device(parameters):
    device = torch.device(parameters.get("device", "cuda"))
    if device.type != "cuda":
        raise RuntimeError("nfopp runs the planner step on MI355X only: set parameters.device='cuda' (got %r); "
                           "there is no CPU fallback" % (parameters.get("device"),))
    return device


class PlannerFactory(object):
    @staticmethod
    def make_onf_planner(collision_checker, device="cuda"):
        device = _device({"device": device})
        collision_model = ONF(1.5, 1).to(device)
        collision_optimizer = torch.optim.Adam(collision_model.parameters(), 1e-3, betas=(0.9, 0.9))
        trajectory = torch.zeros(100, 2, requires_grad=True, device=device)
        trajectory_optimizer = torch.optim.Adam([trajectory], 1e-2, betas=(0.9, 0.999))
        return NERFOptPlanner(trajectory, collision_model, collision_checker, collision_optimizer, trajectory_optimizer,
                              trajectory_random_offset=0.02, collision_weight=0.01, velocity_hessian_weight=3,
                              random_field_points=10, init_collision_iteration=400)

    @staticmethod
    def make_constrained_onf_planner(collision_checker, parameters=None):
        if parameters is None:
            parameters = DEFAULT_PARAMETERS
        factory = UniversalFactory([ONF, ConstrainedNERFOptPlanner, TrajectoryInitializer, AstarTrajectoryInitializer])
        device = _device(parameters)
        collision_model = factory.make_from_parameters(parameters.collision_model).to(device)
        collision_optimizer = torch.optim.Adam(collision_model.parameters(), **parameters.collision_optimizer)
        trajectory = torch.zeros(parameters.trajectory_length, 3, requires_grad=True, device=device)
        trajectory_optimizer = torch.optim.Adam([trajectory], **parameters.trajectory_optimizer)
        trajectory_initializer = factory.make_from_parameters(parameters.trajectory_initializer,
                                                              collision_checker=collision_checker)
        return factory.make_from_parameters(parameters.planner, trajectory=trajectory, collision_model=collision_model,
                                            collision_checker=collision_checker,
                                            collision_optimizer=collision_optimizer,
                                            trajectory_optimizer=trajectory_optimizer,
                                            trajectory_initializer=trajectory_initializer)
