"""Planner construction behind the reference's two factory entry points.

What a driver of the reference relies on (nfop/planner_factory.py:11-77, nfop/utils/universal_factory.py:9-44) and what is
kept here: `PlannerFactory.make_onf_planner(checker)` and `PlannerFactory.make_constrained_onf_planner(checker, parameters)`
return a ready planner; nested parameter dictionaries whose `name` (or `type`) names a registered class are turned into
objects, keyword arguments that class does not take are dropped without complaint, an unregistered name raises
`KeyError("Unknown class ...")`; the default parameter block has the reference's values (and, like the reference's, no
`trajectory_initializer` entry).  The one visible difference is `device`: this path is HIP-only, so the default is "cuda"
and "cpu" is refused.
"""
import inspect

import torch

from .host_utils import AstarTrajectoryInitializer, AttributeDict, TrajectoryInitializer
from .onf_model import ONF
from .planner import ConstrainedNERFOptPlanner, NERFOptPlanner


def _as_attribute_dicts(tree):
    """Plain nested dicts -> AttributeDicts (attribute access, as the reference's drivers index their parameters)."""
    if isinstance(tree, dict):
        return AttributeDict({key: _as_attribute_dicts(value) for key, value in tree.items()})
    return tree


# values of the reference's default block (nfop/planner_factory.py:11-46), device aside
_ADAM_DEFAULT = {"lr": 1e-2, "betas": (0.9, 0.9)}
DEFAULT_PARAMETERS = _as_attribute_dicts({
    "device": "cuda",
    "trajectory_length": 100,
    "collision_model": {"name": "ONF", "mean": 0, "sigma": 10, "use_cos": True, "bias": True, "use_normal_init": True},
    "collision_optimizer": dict(_ADAM_DEFAULT),
    "trajectory_optimizer": dict(_ADAM_DEFAULT),
    "planner": {
        "name": "ConstrainedNERFOptPlanner",
        "collision_weight": 1, "velocity_hessian_weight": 0.5, "angle_weight": 0.5, "boundary_weight": 1,
        "constraint_deltas_weight": 0.2, "multipliers_lr": 0.001, "collision_multipliers_lr": 1e-3,
        "trajectory_random_offset": 0.02, "random_field_points": 10,
        "init_collision_iteration": 0, "init_collision_points": 100,
        "reparametrize_trajectory_freq": 10, "optimize_collision_model_freq": 1,
    },
})

# the fixed recipe of the 2-D planner (nfop/planner_factory.py:50-59)
_ONF_2D = {"mean": 1.5, "sigma": 1, "field_lr": 1e-3, "field_betas": (0.9, 0.9), "waypoints": 100,
           "trajectory_lr": 1e-2, "trajectory_betas": (0.9, 0.999),
           "planner": {"trajectory_random_offset": 0.02, "collision_weight": 0.01, "velocity_hessian_weight": 3,
                       "random_field_points": 10, "init_collision_iteration": 400}}


class UniversalFactory(object):
    """Builds objects from parameter dictionaries: `{"name": <registered class>, <constructor arguments>...}`."""

    NAME_KEYS = ("name", "type")
    WHOLE_DICT_ARGUMENTS = ("parameters", "params", "param", "parameter")   # constructors that want the dict itself

    def __init__(self, classes):
        self._registry = dict((cls.__name__, cls) for cls in classes)

    def _lookup(self, parameters):
        found = None
        for key in self.NAME_KEYS:
            if key in parameters:
                class_name = parameters[key]
                if class_name not in self._registry:
                    raise KeyError("Unknown class %s" % class_name)
                found = self._registry[class_name]
        return found

    def make_from_parameters(self, parameters, **kwargs):
        if not isinstance(parameters, dict):
            return parameters                      # leaves: numbers, tuples, ready-made objects
        cls = self._lookup(parameters)
        if cls is None:
            return parameters                      # a dictionary that names no class stays a dictionary
        arguments = dict(kwargs)
        arguments.update((key, self.make_from_parameters(value)) for key, value in parameters.items())
        takes = inspect.signature(cls).parameters
        arguments.update((key, parameters) for key in self.WHOLE_DICT_ARGUMENTS if key in takes)
        return cls(**dict((key, value) for key, value in arguments.items() if key in takes))


def _hip_device(requested):
    device = torch.device(requested)
    if device.type != "cuda":
        raise RuntimeError("nfopp runs the planner step on MI355X only: set parameters.device='cuda' (got %r); "
                           "there is no CPU fallback" % (requested,))
    return device


def _adam(tensors, lr, betas):
    return torch.optim.Adam(tensors, lr, betas=betas)


class PlannerFactory(object):
    @staticmethod
    def make_onf_planner(collision_checker, device="cuda"):
        recipe, device = _ONF_2D, _hip_device(device)
        field = ONF(recipe["mean"], recipe["sigma"]).to(device)
        waypoints = torch.zeros(recipe["waypoints"], 2, requires_grad=True, device=device)
        return NERFOptPlanner(waypoints, field, collision_checker,
                              _adam(field.parameters(), recipe["field_lr"], recipe["field_betas"]),
                              _adam([waypoints], recipe["trajectory_lr"], recipe["trajectory_betas"]), **recipe["planner"])

    @staticmethod
    def make_constrained_onf_planner(collision_checker, parameters=None):
        p = DEFAULT_PARAMETERS if parameters is None else parameters
        device = _hip_device(p.get("device", "cuda"))
        build = UniversalFactory([ONF, ConstrainedNERFOptPlanner, TrajectoryInitializer,
                                  AstarTrajectoryInitializer]).make_from_parameters
        parts = {"collision_checker": collision_checker}
        parts["collision_model"] = build(p.collision_model).to(device)
        parts["trajectory"] = torch.zeros(p.trajectory_length, 3, requires_grad=True, device=device)
        parts["collision_optimizer"] = torch.optim.Adam(parts["collision_model"].parameters(), **p.collision_optimizer)
        parts["trajectory_optimizer"] = torch.optim.Adam([parts["trajectory"]], **p.trajectory_optimizer)
        # DEFAULT_PARAMETERS has no initialiser entry: AttributeError here, exactly as with the reference's defaults
        parts["trajectory_initializer"] = build(p.trajectory_initializer, collision_checker=collision_checker)
        return build(p.planner, **parts)
