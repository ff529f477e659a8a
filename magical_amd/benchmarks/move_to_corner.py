"""MoveToCorner (mirror of magical/benchmarks/move_to_corner.py, every rand_* branch)."""
import math
import warnings

import numpy as np

from .. import entities as en
from ..base_env import BaseEnv
from ._scoring import row_norm


class MoveToCornerEnv(BaseEnv):
    def __init__(self, rand_shape_colour=False, rand_shape_type=False, rand_poses=False, debug_reward=False, **kwargs):
        self.rand_shape_colour, self.rand_shape_type, self.rand_poses, self.debug_reward = rand_shape_colour, rand_shape_type, rand_poses, debug_reward
        self.variable_worlds = bool(rand_shape_type)      # the block's shape type is drawn per episode
        if debug_reward:     # move_to_corner.py:25-29
            warnings.warn('DEBUG REWARD ENABLED IN MOVE-TO-CORNER ENV! This reward is ONLY intended for training RL algorithms '
                          "during debugging, so don't forget to disable it when benchmarking IL")
        super().__init__(**kwargs)

    def step(self, actions):   # move_to_corner.py:77-98: a dense, heavily shaped reward instead of 0, computed on the device
        obs, rew, done, info = super().step(actions)
        if self.debug_reward:
            rew = self.debug_shaped_reward()
        return obs, rew, done, info

    def debug_shaped_reward(self):
        """float64[N] on the device, from the poses after this step (for envs that were just auto-reset: of the new
        episode's initial state, the closest batched reading of a reward handed out together with `done`)."""
        import torch
        rows = self._pose_rows
        sx, sy = self.state_p[rows[self.__shape_ref.body, 0]], self.state_p[rows[self.__shape_ref.body, 1]]
        rx, ry = self.state_p[rows[self._robot.body, 0]], self.state_p[rows[self._robot.body, 1]]
        sx, sy, rx, ry = (v.to(torch.float64) for v in (sx, sy, rx, ry))
        shape_to_corner_dist = torch.sqrt((sx - 0.0) ** 2 + (sy - 1.0) ** 2)          # (sic) target (0, 1), :89-91
        robot_to_shape_dist = torch.sqrt((rx - sx) ** 2 + (ry - sy) ** 2)
        shaping = -shape_to_corner_dist / 5 - torch.clamp(robot_to_shape_dist, min=0.2) / 20
        # + score_on_end_of_traj() of the same poses (:66-75)
        dist = torch.sqrt((-1.0 - sx) ** 2 + (1.0 - sy) ** 2)
        furthest, succeed = 2.0 ** 0.5, 2.0 ** 0.5 / 2
        score = torch.clamp(torch.clamp(furthest - dist, min=0.0) / (furthest - succeed), max=1.0)
        return shaping + score

    def sample_variation(self, rng, k):   # move_to_corner.py:42-63, in the reference's order: colour, then poses
        if not (self.rand_shape_colour or self.rand_shape_type or self.rand_poses):
            return None
        var = {}
        if self.rand_shape_colour:
            var['colours'] = {self.__shape_ref: en.draw_choice(rng, en.SHAPE_COLOUR_NAMES)}
        if self.rand_shape_type:
            var['shape_types'] = {self.__shape_ref: en.draw_choice(rng, en.SHAPE_TYPE_NAMES)}
        if self.rand_poses:
            var['randomise_poses'] = ((self._robot, self.__shape_ref), dict(
                rand_pos=True, rand_rot=True, rel_pos_linf_limits=self.JITTER_POS_BOUND, rel_rot_limits=self.JITTER_ROT_BOUND))
        return var

    def sample_variation_batch(self, brng, env_idx):   # the same draws, all envs at once (batch_rng.py)
        if not (self.rand_shape_colour or self.rand_shape_type or self.rand_poses):
            return None
        var, m, shape = {}, brng.m, self.__shape_ref
        if self.rand_shape_colour:
            var['colours'] = np.tile(self._default_colours, (m, 1))
            var['colours'][:, shape.ent_id] = en.colour_id_of_draw()[brng.randint(len(en.SHAPE_COLOUR_NAMES))[:, 0]]
        if self.rand_shape_type:
            var['shape_types'] = np.tile(self._default_shape_types, (m, 1))
            var['shape_types'][:, shape.ent_id] = en.type_id_of_draw()[brng.randint(len(en.SHAPE_TYPE_NAMES))[:, 0]]
        if self.rand_poses:
            var['randomise_poses'] = ((self._robot, shape), dict(
                rand_pos=True, rand_rot=True, rel_pos_linf_limits=self.JITTER_POS_BOUND, rel_rot_limits=self.JITTER_ROT_BOUND))
        return var

    def on_reset(self):   # move_to_corner.py:31-54
        robot = self._make_robot(np.asarray((0.4, -0.0)), 0.55 * math.pi)
        self.add_entities([robot])
        shape = self._make_shape(shape_type=en.ShapeType.SQUARE, colour_name='red',
                                 init_pos=np.asarray((0.1, -0.65)), init_angle=0.13 * math.pi)
        self.add_entities([shape])
        self.__shape_ref = shape

    def device_score_spec(self):   # the same score on the device (mgx_engine_score_points)
        from .. import _native as nat
        return dict(task=nat.SCORE_CORNER, ents=[self.__shape_ref.ent_id], params=(np.sqrt(2), np.sqrt(2) - np.sqrt(2) / 2))

    def score_on_end_of_traj(self, poses):   # move_to_corner.py:66-75
        shape_pos = poses[:, self.__shape_ref.body, :2]
        dist = row_norm(np.asarray([-1.0, 1.0]) - shape_pos)   # target is top left
        succeed_dist = np.sqrt(2) / 2
        furthest_dist = np.sqrt(2)
        drange = furthest_dist - succeed_dist
        return np.minimum(1.0, np.maximum(0.0, furthest_dist - dist) / drange)
