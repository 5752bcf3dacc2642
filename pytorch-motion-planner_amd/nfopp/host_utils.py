"""Host-side glue the hot path's callers expect: SE(2) pose batch, numpy ground-truth checkers, the straight-line
trajectory initialiser and an attribute dict.  None of this is on the accelerated path; it exists so that the planner
classes are usable with the reference's driver code."""
import numpy as np
import torch


class AttributeDict(dict):
    """dict with attribute access (stand-in for pytorch_lightning's AttributeDict, which this image lacks)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value


def wrap_angle_np(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def wrap_angle_t(a):
    """nfop/torch_math.py:5-7."""
    return (a + np.pi) % (2 * np.pi) - np.pi


class Position2(object):
    """Batch of SE(2) poses with the accessors collision checkers use (nfop/utils/position2.py:10-110)."""

    def __init__(self, x, y, angle):
        self._x, self._y, self._angle = x, y, angle

    x = property(lambda self: self._x)
    y = property(lambda self: self._y)
    rotation = property(lambda self: self._angle)

    @property
    def translation(self):
        return np.array([self._x, self._y]).T

    @classmethod
    def from_vec(cls, vec):
        vec = np.asarray(vec)
        if vec.ndim == 1:
            return cls(vec[0], vec[1], vec[2])
        return cls(vec[:, 0], vec[:, 1], vec[:, 2])

    def as_vec(self):
        return np.array([self._x, self._y, self._angle]).T

    def inv(self):
        c, s = np.cos(self._angle), np.sin(self._angle)
        return Position2(-self._x * c - self._y * s, self._x * s - self._y * c, -self._angle)

    def __len__(self):
        return 1 if np.ndim(self._x) == 0 else np.shape(self._x)[0]


class CollisionChecker(object):
    """Bounds-only checker and the protocol (`check_collision`, `update_obstacle_points`, `update_boundaries`,
    `get_boundaries`) of nfop/collision_checker/collision_checker.py:4-28."""

    def __init__(self, collision_boundaries=None):
        self._obstacle_points = np.zeros((0, 2))
        self._boundaries = collision_boundaries

    def check_collision(self, test_positions):
        return self._check_boundaries_collision(test_positions)

    def _check_boundaries_collision(self, xy):
        if self._boundaries is None:
            return False
        b = self._boundaries
        return (xy[:, 0] > b[1]) | (xy[:, 0] < b[0]) | (xy[:, 1] > b[3]) | (xy[:, 1] < b[2])

    def update_obstacle_points(self, points):
        self._obstacle_points = points

    def update_boundaries(self, boundaries):
        self._boundaries = boundaries

    def get_boundaries(self):
        return self._boundaries


class CircleCollisionChecker(CollisionChecker):
    """Disc robot against a point cloud (nfop/collision_checker/circle_collision_checker.py:6-14)."""

    def __init__(self, robot_radius, boundaries=None):
        super().__init__(boundaries)
        self._robot_radius = robot_radius

    def check_collision(self, test_positions):
        d = np.linalg.norm(test_positions[None] - self._obstacle_points[:, None], axis=2)
        return np.any(d < self._robot_radius, axis=0) | self._check_boundaries_collision(test_positions)


class CircleDirectedCollisionChecker(CircleCollisionChecker):
    """Same test on the translation of SE(2) poses (circle_directed_collision_checker.py:4-6)."""

    def check_collision(self, test_positions):
        return super().check_collision(test_positions.translation)


class RectangleCollisionChecker(CollisionChecker):
    """Box robot: obstacle points moved into the robot frame (rectangle_collision_checker.py:6-26)."""

    def __init__(self, box, collision_boundaries=None):
        super().__init__(collision_boundaries)
        self._box = box

    def check_collision(self, test_positions):
        x, y, th = test_positions.x, test_positions.y, test_positions.rotation
        c, s = np.cos(th)[:, None], np.sin(th)[:, None]
        dx = self._obstacle_points[:, 0][None] - np.asarray(x)[:, None]
        dy = self._obstacle_points[:, 1][None] - np.asarray(y)[:, None]
        rx, ry = c * dx + s * dy, -s * dx + c * dy
        b = self._box
        inside = (rx > b[0]) & (rx < b[1]) & (ry > b[2]) & (ry < b[3])
        return np.any(inside, axis=1) | self._check_boundaries_collision(test_positions.translation)


class TrajectoryInitializer(object):
    """Straight line in xy, heading interpolated along the wrapped shortest rotation
    (nfop/trajectory_initializer.py:7-43).  One-time host work per `init()`."""

    def __init__(self, collision_checker=None, init_angles_with_trajectory=False):
        self._collision_checker = collision_checker
        self._init_angles_with_trajectory = init_angles_with_trajectory

    def initialize_trajectory(self, trajectory, start_point, goal_point):
        n = trajectory.shape[0] + 2
        with torch.no_grad():
            for k in range(2):
                trajectory[:, k] = torch.linspace(start_point[0, k], goal_point[0, k], n)[1:-1]
            goal_angle = wrap_angle_t(goal_point[0, 2] - start_point[0, 2]) + start_point[0, 2]
            trajectory[:, 2] = torch.linspace(start_point[0, 2], goal_angle, n)[1:-1]
            if self._init_angles_with_trajectory:
                full = torch.cat([start_point, trajectory, goal_point], dim=0)
                angles = torch.atan2(full[2:, 1] - full[:-2, 1], full[2:, 0] - full[:-2, 0])
                m = trajectory.shape[0]
                w = torch.cat([torch.linspace(0., 1, m // 2), torch.linspace(1., 0, (m + 1) // 2)], dim=0)
                trajectory[:, 2] = trajectory[:, 2] + wrap_angle_t(angles - trajectory[:, 2]) * w


class AstarTrajectoryInitializer(object):
    """A*/JPS seeding (nfop/astar/) is one-time host work outside the accelerated path."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("AstarTrajectoryInitializer is outside the hot path rebuilt here; "
                                  "use TrajectoryInitializer or pass your own initializer object")
