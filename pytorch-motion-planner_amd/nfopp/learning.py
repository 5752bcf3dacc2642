"""Continuous ONF learning for a batch of trajectories, entirely on the device: ground-truth checkers, training-pose
generation, retained-pool resampling (csrc/sampling.hip) and the data-parallel fitting step (batch.OnfFitter).

Reference semantics (one trajectory, host numpy): nfop/nerf_opt_planner.py:76-141.  Per step and per trajectory the
sample set is  N-1 "course" poses + the retained pool (<= 100 poses kept by weighted resampling) + `random_field_points`
uniform poses; labels come from the ground-truth checker; all trajectories of all ranks fit ONE shared field.
Differences to the reference, by construction of the batch: the pool holds min(100, N-1) poses and is full from the
first step (the reference grows it for N <= 100), and draws come from a counter-based Philox stream instead of
numpy's global generator (distributional parity, SURVEY.md "Hard parts").
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _f4(values):
    return (ctypes.c_float * 4)(*[float(v) for v in values])


class DeviceCircleChecker(object):
    """Disc robot against a point cloud + bounds (nfop/collision_checker/circle_collision_checker.py).  With more than a
    few obstacle points they are sorted into a uniform cell index (cell >= robot radius) once, on the host, and every
    pose tests the points of its 3 x 3 cells only: same predicate, same labels."""

    INDEX_FROM = 32   # obstacle points from which the cell index pays

    def __init__(self, obstacle_points, robot_radius, boundaries=None, device="cuda"):
        pts = np.ascontiguousarray(obstacle_points, dtype=np.float32).reshape(-1, 2)
        self.radius, self.boundaries = float(robot_radius), boundaries
        self.cells = None
        if len(pts) >= self.INDEX_FROM and self.radius > 0:
            lo, hi = pts.min(0), pts.max(0)
            # a little more than the radius: fp32 rounding of the cell arithmetic must not move a point two cells away
            size = np.float32(max(self.radius * 1.001, float((hi - lo).max()) / 64.0))
            nx, ny = (int(np.floor((hi[k] - lo[k]) / size)) + 1 for k in (0, 1))
            # the kernel's own cell arithmetic (fp32 subtract, divide, floor), so points and poses agree on the cells
            cx = np.clip(np.floor((pts[:, 0] - lo[0]) / size).astype(np.int64), 0, nx - 1)
            cy = np.clip(np.floor((pts[:, 1] - lo[1]) / size).astype(np.int64), 0, ny - 1)
            cell = cy * nx + cx
            order = np.argsort(cell, kind="stable")
            start = np.searchsorted(cell[order], np.arange(nx * ny + 1)).astype(np.int32)
            pts = pts[order]
            self.cells = (torch.tensor(start, device=device), nx, ny, float(lo[0]), float(lo[1]), float(size))
        self.obstacles = torch.tensor(pts, device=device)

    def labels(self, poses, out=None):
        n, d = poses.shape
        out = torch.empty(n, dtype=torch.float32, device=poses.device) if out is None else out
        b = _f4(self.boundaries) if self.boundaries is not None else None
        lib = _lib.load()
        if self.cells is None:
            _lib.check(lib.nfopp_check_collision_circle(_lib.ptr(poses), n, d, _lib.ptr(self.obstacles),
                                                        self.obstacles.shape[0], self.radius, b, _lib.ptr(out),
                                                        _lib.stream_ptr()))
        else:
            start, nx, ny, x0, y0, size = self.cells
            _lib.check(lib.nfopp_check_collision_circle_cells(_lib.ptr(poses), n, d, _lib.ptr(self.obstacles),
                                                              self.obstacles.shape[0], _lib.ptr(start, torch.int32), nx, ny,
                                                              x0, y0, size, self.radius, b, _lib.ptr(out), _lib.stream_ptr()))
        return out


class DeviceRectangleChecker(object):
    """Box robot (x0, x1, y0, y1 in its own frame) against a point cloud (rectangle_collision_checker.py)."""

    def __init__(self, obstacle_points, box, boundaries=None, device="cuda"):
        self.obstacles = torch.tensor(np.ascontiguousarray(obstacle_points, dtype=np.float32), device=device).reshape(-1, 2)
        self.box, self.boundaries = tuple(box), boundaries

    def labels(self, poses, out=None):
        n = poses.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=poses.device) if out is None else out
        b = _f4(self.boundaries) if self.boundaries is not None else None
        _lib.check(_lib.load().nfopp_check_collision_rectangle(_lib.ptr(poses), n, _lib.ptr(self.obstacles),
                                                               self.obstacles.shape[0], _f4(self.box), b, _lib.ptr(out),
                                                               _lib.stream_ptr()))
        return out


class DeviceGridChecker(object):
    """uint8 occupancy image (MapCollisionChecker of notebooks/onf_planner_image_map.ipynb cell 2)."""

    def __init__(self, grid, origin_x, origin_y, cell_size, device="cuda"):
        self.grid = torch.tensor(np.ascontiguousarray(grid, dtype=np.uint8), device=device)
        self.origin_x, self.origin_y, self.cell_size = float(origin_x), float(origin_y), float(cell_size)

    def labels(self, poses, out=None):
        n, d = poses.shape
        out = torch.empty(n, dtype=torch.float32, device=poses.device) if out is None else out
        _lib.check(_lib.load().nfopp_check_collision_grid(_lib.ptr(poses), n, d, _lib.ptr(self.grid, torch.uint8),
                                                          self.grid.shape[0], self.grid.shape[1], self.origin_x,
                                                          self.origin_y, self.cell_size, _lib.ptr(out), _lib.stream_ptr()))
        return out


class BatchSampler(object):
    """Per-trajectory training-pose generation with a retained pool, for B trajectories on one GPU."""

    def __init__(self, onf, batch, n_waypoints, course_sigma=1.5, fine_sigma=0.02, angle_sigma=0.0, n_field=10,
                 pool_cap=100, device="cuda", seed=0, traj_index_offset=0):
        self.onf, self.B, self.N, self.D = onf, int(batch), int(n_waypoints), onf.point_dim
        self.cap = min(int(pool_cap), self.N - 1)
        self.n_field = int(n_field)
        self.sigmas = (float(course_sigma), float(fine_sigma), float(angle_sigma))
        self.seed, self.offset, self.traj_index_offset = int(seed), 0, int(traj_index_offset)
        f32 = dict(dtype=torch.float32, device=device)
        B, N, D = self.B, self.N, self.D
        self.C = self.cap + N - 1
        self.S = (N - 1) + self.cap + self.n_field
        self.pool = torch.zeros(B, self.cap, D, **f32)
        self.pool_age = torch.zeros(B, self.cap, **f32)
        self.pool_full = False
        self.cand = torch.zeros(B, self.C, D, **f32)
        self.cand_age = torch.zeros(B, self.C, **f32)
        self.cand_out = torch.zeros(B, self.C, 4, **f32)
        self.samples = torch.zeros(B, self.S, D, **f32)
        self.labels = torch.zeros(B * self.S, **f32)

    def draw(self, prev_traj, bounds):
        """Fills `self.samples` [B, S, D] from the previous trajectories [B, N, D]; returns it flattened [B*S, D]."""
        lib = _lib.load()
        pool_n = self.cap if self.pool_full else 0
        n_cand = pool_n + self.N - 1
        course, fine, angle = self.sigmas
        _lib.check(lib.nfopp_sample_candidates(_lib.ptr(prev_traj), self.B, self.N, self.D, self.cap, pool_n, self.n_field,
                                               course, fine, angle, _f4(bounds), self.seed, self.offset,
                                               self.traj_index_offset, _lib.ptr(self.pool), _lib.ptr(self.pool_age),
                                               _lib.ptr(self.cand), _lib.ptr(self.cand_age), _lib.ptr(self.samples),
                                               _lib.stream_ptr()))
        if self.cap:
            cfg = self.onf.config_c()   # weights of the pool candidates: sigmoid(ONF) * exp(-0.03 age)  (nerf:124-126)
            _lib.check(lib.nfopp_onf_eval_logits(cfg, _lib.ptr(self.onf.flat_parameters), _lib.ptr(self.cand),
                                                 self.B * self.C, _lib.ptr(self.cand_out), _lib.stream_ptr()))
            _lib.check(lib.nfopp_resample_pool(self.B, n_cand, self.C, self.cap, self.D, self.S, self.N - 1, self.seed, self.offset,
                                               self.traj_index_offset, _lib.ptr(self.cand), _lib.ptr(self.cand_age),
                                               _lib.ptr(self.cand_out), _lib.ptr(self.pool), _lib.ptr(self.pool_age),
                                               _lib.ptr(self.samples), _lib.stream_ptr()))
            self.pool_full = True
        self.offset += 1
        return self.samples.view(self.B * self.S, self.D)
