"""nfopp -- MI355X-native inner loop of the Neural Field Optimal Path Planner.

Host-side mirror of the reference's planner interface over libnfopp_hip.so (C ABI: include/nfopp_hip.h).
"""
from ._lib import LIB_PATH, NfoppError, load as load_library
from .batch import BatchPlanner, OnfFitter, shard_range, straight_line_init
from .engine import TrajectoryEngine, TrajectoryHyper, band_of, inverse_hessian
from .factory import DEFAULT_PARAMETERS, PlannerFactory, UniversalFactory
from .host_utils import (AstarTrajectoryInitializer, AttributeDict, CircleCollisionChecker,
                         CircleDirectedCollisionChecker, CollisionChecker, Position2, RectangleCollisionChecker,
                         TrajectoryInitializer)
from .learning import BatchSampler, DeviceCircleChecker, DeviceGridChecker, DeviceRectangleChecker
from .onf_model import ONF
from .path_tools import PathPostprocessor, init_trajectories
from .planner import ConstrainedNERFOptPlanner, ContinuousPlanner, NERFOptPlanner

__all__ = [
    "BatchPlanner", "BatchSampler", "DeviceCircleChecker", "DeviceGridChecker", "DeviceRectangleChecker", "OnfFitter", "shard_range", "straight_line_init", "LIB_PATH", "NfoppError", "load_library", "TrajectoryEngine", "TrajectoryHyper", "band_of", "inverse_hessian",
    "DEFAULT_PARAMETERS", "PlannerFactory", "UniversalFactory", "AstarTrajectoryInitializer", "AttributeDict",
    "CircleCollisionChecker", "CircleDirectedCollisionChecker", "CollisionChecker", "Position2",
    "RectangleCollisionChecker", "TrajectoryInitializer", "ONF", "ConstrainedNERFOptPlanner", "ContinuousPlanner",
    "NERFOptPlanner", "PathPostprocessor", "init_trajectories",
]
