"""MoveToRegion (mirror of magical/benchmarks/move_to_region.py: Demo, TestColour and TestDynamics branches)."""
import numpy as np

from .. import entities as en
from ..base_env import BaseEnv

DEFAULT_ROBOT_POSE = ((0.058, 0.53), -2.13)
DEFAULT_GOAL_COLOUR = en.ShapeColour.BLUE
DEFAULT_GOAL_XYHW = (-0.62, -0.17, 0.76, 0.75)


class MoveToRegionEnv(BaseEnv):
    def __init__(self, rand_poses_minor=False, rand_poses_full=False, rand_goal_colour=False, **kwargs):
        if rand_poses_minor or rand_poses_full:
            raise NotImplementedError('built: Demo, TestColour, TestDynamics (goal size / poses need per-env geometry: SURVEY.md §8f)')
        self.rand_goal_colour = rand_goal_colour
        super().__init__(**kwargs)

    def sample_variation(self, rng, k):   # move_to_region.py:47-51
        if not self.rand_goal_colour:
            return None
        return {'colours': {self.__goal_ref: rng.choice(np.asarray(en.SHAPE_COLOURS, dtype='object'))}}

    def on_reset(self):   # move_to_region.py:30-63
        goal = en.GoalRegion(*DEFAULT_GOAL_XYHW, DEFAULT_GOAL_COLOUR)
        self.add_entities([goal])
        self.__goal_ref = goal
        robot = self._make_robot(*DEFAULT_ROBOT_POSE)
        self.add_entities([robot])

    def score_on_end_of_traj(self, poses):   # move_to_region.py:85-94
        # goal_shape.point_query(robot_pos)[0] <= 0  <=>  not strictly outside any face of the box
        x, y = poses[:, self._robot.body, 0], poses[:, self._robot.body, 1]
        l, b, r, t = self.__goal_ref.bb
        outside = (x - r > 0.0) | (y - t > 0.0) | (l - x > 0.0) | (b - y > 0.0)
        return np.where(outside, 0.0, 1.0)
