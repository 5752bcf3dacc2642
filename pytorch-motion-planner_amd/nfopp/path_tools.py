"""The steps either side of the planner step, on the device for whole batches:
trajectory initialisation (nfop/trajectory_initializer.py) and path post-processing for the follower
(nfop/ros/path_postprocessor.py)."""
import numpy as np
import torch

from . import _lib
from .host_utils import Position2


def init_trajectories(starts, goals, n_waypoints, init_angles_with_trajectory=False, out=None):
    """Batched `TrajectoryInitializer.initialize_trajectory` (nfop/trajectory_initializer.py:12-45).

    starts, goals: [B, D] fp32 HIP tensors (D = 3, or 2 for point robots) -> [B, N, D] straight-line trajectories with
    torch.linspace's rounding; `init_angles_with_trajectory` pulls the headings towards the travel direction."""
    starts = starts.contiguous().float()
    goals = goals.contiguous().float()
    if not starts.is_cuda:
        raise _lib.NfoppError("init_trajectories needs HIP tensors (there is no CPU path)")
    b, d = starts.shape
    if out is None:
        out = torch.empty(b, n_waypoints, d, dtype=torch.float32, device=starts.device)
    _lib.check(_lib.load().nfopp_init_trajectories(_lib.ptr(starts), _lib.ptr(goals), b, int(n_waypoints), d,
                                                   1 if init_angles_with_trajectory else 0, _lib.ptr(out),
                                                   _lib.stream_ptr()))
    return out


class PathPostprocessor(object):
    """Drop-in for nfop/ros/path_postprocessor.py `PathPostprocessor` (same ctor, same `process`), plus
    `process_batch` for many paths at once.  Output is float64 like the reference's (scipy's spline)."""

    def __init__(self, minimal_distance=0.001, distance_step=0.05, device="cuda"):
        self._distance_step = distance_step
        self._minimal_distance = minimal_distance
        self._device = device

    def process_batch(self, paths):
        """paths: [B, n, 3] fp32 (tensor or array) -> (poses [B, max_count, 3] float64 HIP tensor, counts [B] int32);
        poses[b, :counts[b]] is path b's result.  A path that collapses to < 3 poses raises like the reference."""
        paths = torch.as_tensor(np.asarray(paths, np.float32) if not torch.is_tensor(paths) else paths,
                                dtype=torch.float32, device=self._device).contiguous()
        b, n, d = paths.shape
        if d != 3:
            raise ValueError("paths must be [B, n, 3] (x, y, heading)")
        lib = _lib.load()
        counts = torch.empty(b, dtype=torch.int32, device=paths.device)
        _lib.check(lib.nfopp_path_postprocess(_lib.ptr(paths), b, n, float(self._minimal_distance),
                                              float(self._distance_step), 0, None, _lib.ptr(counts, torch.int32), _lib.stream_ptr()))
        if b and int(counts.min()) < 0:
            raise ValueError("a path collapses to fewer than 3 poses: no quadratic spline through it")
        cap = int(counts.max()) if b else 0
        out = torch.zeros(b, max(cap, 1), 3, dtype=torch.float64, device=paths.device)
        if cap:
            _lib.check(lib.nfopp_path_postprocess(_lib.ptr(paths), b, n, float(self._minimal_distance),
                                                  float(self._distance_step), cap, _lib.ptr(out, torch.float64), _lib.ptr(counts, torch.int32),
                                                  _lib.stream_ptr()))
        return out[:, :cap], counts

    def process(self, trajectory):
        """`trajectory`: Position2 batch (as handed over by the ROS adapter, goal_planner_adapter.py:56-60)."""
        if len(trajectory) < 3:
            return trajectory
        vec = np.asarray(trajectory.as_vec(), np.float32)
        out, counts = self.process_batch(vec[None])
        return Position2.from_vec(out[0, :int(counts[0])].cpu().numpy())
