"""MoveToCorner (mirror of magical/benchmarks/move_to_corner.py: Demo, TestColour and TestDynamics branches)."""
import math

import numpy as np

from .. import entities as en
from ..base_env import BaseEnv
from ._scoring import row_norm


class MoveToCornerEnv(BaseEnv):
    def __init__(self, rand_shape_colour=False, rand_shape_type=False, rand_poses=False, debug_reward=False, **kwargs):
        if rand_shape_type or debug_reward:
            raise NotImplementedError('built: Demo, TestColour, TestJitter, TestDynamics (shape types need per-env geometry: SURVEY.md §8f)')
        self.rand_shape_colour, self.rand_poses = rand_shape_colour, rand_poses
        super().__init__(**kwargs)

    def sample_variation(self, rng, k):   # move_to_corner.py:42-63, in the reference's order: colour, then poses
        if not (self.rand_shape_colour or self.rand_poses):
            return None
        var = {}
        if self.rand_shape_colour:
            var['colours'] = {self.__shape_ref: rng.choice(np.asarray(en.SHAPE_COLOURS, dtype='object'))}
        if self.rand_poses:
            var['randomise_poses'] = ((self._robot, self.__shape_ref), dict(
                rand_pos=True, rand_rot=True, rel_pos_linf_limits=self.JITTER_POS_BOUND, rel_rot_limits=self.JITTER_ROT_BOUND))
        return var

    def on_reset(self):   # move_to_corner.py:31-54
        robot = self._make_robot(np.asarray((0.4, -0.0)), 0.55 * math.pi)
        self.add_entities([robot])
        shape = self._make_shape(shape_type=en.ShapeType.SQUARE, colour_name='red',
                                 init_pos=np.asarray((0.1, -0.65)), init_angle=0.13 * math.pi)
        self.add_entities([shape])
        self.__shape_ref = shape

    def score_on_end_of_traj(self, poses):   # move_to_corner.py:66-75
        shape_pos = poses[:, self.__shape_ref.body, :2]
        dist = row_norm(np.asarray([-1.0, 1.0]) - shape_pos)   # target is top left
        succeed_dist = np.sqrt(2) / 2
        furthest_dist = np.sqrt(2)
        drange = furthest_dist - succeed_dist
        return np.minimum(1.0, np.maximum(0.0, furthest_dist - dist) / drange)
