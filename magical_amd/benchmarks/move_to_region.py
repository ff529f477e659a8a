"""MoveToRegion (mirror of magical/benchmarks/move_to_region.py: all six variants)."""
import numpy as np

from .. import entities as en
from .. import geom
from ..base_env import BaseEnv

DEFAULT_ROBOT_POSE = ((0.058, 0.53), -2.13)
DEFAULT_GOAL_COLOUR = en.ShapeColour.BLUE
DEFAULT_GOAL_XYHW = (-0.62, -0.17, 0.76, 0.75)


class MoveToRegionEnv(BaseEnv):
    score_needs_poses = False      # the score is a function of the goal regions' overlap sets (k_score on the device)

    def __init__(self, rand_poses_minor=False, rand_poses_full=False, rand_goal_colour=False, **kwargs):
        assert not (rand_poses_minor and rand_poses_full), "cannot specify both 'rand_poses_minor' and 'rand_poses_full'"
        self.rand_poses_minor, self.rand_poses_full, self.rand_goal_colour = rand_poses_minor, rand_poses_full, rand_goal_colour
        super().__init__(**kwargs)

    def sample_variation(self, rng, k):   # move_to_region.py:31-78, in the reference's order: goal size, colour, then poses
        if not (self.rand_poses_minor or self.rand_poses_full or self.rand_goal_colour):
            return None
        var = {}
        if self.rand_poses_minor or self.rand_poses_full:
            hw_bound = self.JITTER_TARGET_BOUND if self.rand_poses_minor else None
            var['goal_hw'] = {self.__goal_ref: geom.randomise_hw(self.RAND_GOAL_MIN_SIZE, self.RAND_GOAL_MAX_SIZE, rng,
                                                                 current_hw=DEFAULT_GOAL_XYHW[2:], linf_bound=hw_bound)}
        if self.rand_goal_colour:
            var['colours'] = {self.__goal_ref: en.draw_choice(rng, en.SHAPE_COLOUR_NAMES)}
        if self.rand_poses_minor or self.rand_poses_full:
            # the goal region is never rotated; under minor jitter only the robot's rotation is bounded
            pos_limits, rot_limits = (self.JITTER_POS_BOUND, [None, self.JITTER_ROT_BOUND]) if self.rand_poses_minor else (None, None)
            var['randomise_poses'] = ((self.__goal_ref, self._robot), dict(
                rand_pos=True, rand_rot=(False, True), rel_pos_linf_limits=pos_limits, rel_rot_limits=rot_limits))
        return var

    def sample_variation_batch(self, brng, env_idx):   # the same draws, all envs at once (batch_rng.py)
        if not (self.rand_poses_minor or self.rand_poses_full or self.rand_goal_colour):
            return None
        from ..batch_rng import uniform_hw
        var, m, goal = {}, brng.m, self.__goal_ref
        if self.rand_poses_minor or self.rand_poses_full:
            hw_bound = self.JITTER_TARGET_BOUND if self.rand_poses_minor else None
            var['goal_hw'] = {goal.ent_id: uniform_hw(brng.random_sample(2), self.RAND_GOAL_MIN_SIZE, self.RAND_GOAL_MAX_SIZE,
                                                     current_hw=DEFAULT_GOAL_XYHW[2:], linf_bound=hw_bound)}
        if self.rand_goal_colour:
            var['colours'] = np.tile(self._default_colours, (m, 1))
            var['colours'][:, goal.ent_id] = en.colour_id_of_draw()[brng.randint(len(en.SHAPE_COLOUR_NAMES))[:, 0]]
        if self.rand_poses_minor or self.rand_poses_full:
            pos_limits, rot_limits = (self.JITTER_POS_BOUND, [None, self.JITTER_ROT_BOUND]) if self.rand_poses_minor else (None, None)
            var['randomise_poses'] = ((goal, self._robot), dict(
                rand_pos=True, rand_rot=(False, True), rel_pos_linf_limits=pos_limits, rel_rot_limits=rot_limits))
        return var

    def on_reset(self):   # move_to_region.py:30-63
        goal = en.GoalRegion(*DEFAULT_GOAL_XYHW, DEFAULT_GOAL_COLOUR)
        self.add_entities([goal])
        self.__goal_ref = goal
        robot = self._make_robot(*DEFAULT_ROBOT_POSE)
        self.add_entities([robot])

    def score_on_end_of_traj(self, poses):   # move_to_region.py:85-94
        # goal_shape.point_query(robot_pos)[0] <= 0  <=>  not strictly outside any face of the box
        if poses is None:      # the engine's own episode ends: bit 0 of the device's overlap flags is exactly this test
            g = self._goal_ent_idx.index(self.__goal_ref.ent_id)
            return np.where(self._overlap[g, self._robot.ent_id] & 1, 1.0, 0.0)
        x, y = poses[:, self._robot.body, 0], poses[:, self._robot.body, 1]
        l, b, r, t = self.goal_bb(self.__goal_ref)
        outside = (x - r > 0.0) | (y - t > 0.0) | (l - x > 0.0) | (b - y > 0.0)
        return np.where(outside, 0.0, 1.0)
