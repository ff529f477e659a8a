"""MatchRegions (mirror of magical/benchmarks/match_regions.py: Demo, TestColour and TestDynamics branches)."""
import math

import numpy as np

from .. import entities as en
from ..base_env import BaseEnv
from ._scoring import overlapping_ents


class MatchRegionsEnv(BaseEnv):
    def __init__(self, rand_target_colour=False, rand_shape_type=False, rand_shape_count=False,
                 rand_layout_minor=False, rand_layout_full=False, **kwargs):
        if rand_shape_type or rand_shape_count or rand_layout_minor or rand_layout_full:
            raise NotImplementedError('built: Demo, TestColour, TestDynamics (shape types / counts / layouts need per-env geometry: SURVEY.md §8f)')
        self.rand_target_colour = rand_target_colour
        super().__init__(**kwargs)

    def sample_variation(self, rng, k):   # match_regions.py:51-58: the sensor and the targets take the drawn colour, the
        if not self.rand_target_colour:   # distractor groups the remaining ones in SHAPE_COLOURS order
            return None
        target_colour = rng.choice(en.SHAPE_COLOUR_NAMES)
        distractor_colours = [c for c in en.SHAPE_COLOUR_NAMES if c != target_colour]
        colours = {self.__sensor_ref: target_colour}
        colours.update({s: target_colour for s in self.__target_shapes})
        colours.update({s: distractor_colours[g] for s, g in zip(self.__distractor_shapes, self.__distractor_group)})
        return {'colours': colours}

    def on_reset(self):   # match_regions.py:44-162
        robot = self._make_robot(np.asarray((-0.5, 0.1)), -math.pi * 1.2)
        target_colour = en.ShapeColour.GREEN
        distractor_colours = [c for c in en.SHAPE_COLOURS if c != target_colour]
        sensor = en.GoalRegion(0.1, 0.7, 0.7, 0.6, target_colour)    # x, y, h, w
        self.add_entities([sensor])
        self.__sensor_ref = sensor
        target_types = [en.ShapeType.STAR, en.ShapeType.SQUARE]
        distractor_types = [[], [en.ShapeType.PENTAGON], [en.ShapeType.CIRCLE, en.ShapeType.PENTAGON]]
        target_poses = [(0.8, -0.7, 2.37), (-0.68, 0.72, 1.28)]
        distractor_poses = [[], [(-0.05, -0.2, -1.09)], [(-0.75, -0.55, 2.78), (0.3, -0.82, -1.15)]]
        self.__target_shapes = [
            self._make_shape(shape_type=st, colour_name=target_colour, init_pos=(x, y), init_angle=a)
            for st, (x, y, a) in zip(target_types, target_poses)]
        self.__distractor_shapes, self.__distractor_group = [], []
        for group, (colour, types, poses) in enumerate(zip(distractor_colours, distractor_types, distractor_poses)):
            for st, (x, y, a) in zip(types, poses):
                self.__distractor_group.append(group)
                self.__distractor_shapes.append(self._make_shape(shape_type=st, colour_name=colour, init_pos=(x, y), init_angle=a))
        self.add_entities(self.__target_shapes + self.__distractor_shapes)
        self.add_entities([robot])    # last, so it is drawn on top

    def score_on_end_of_traj(self, poses):   # match_regions.py:193-213
        ents = self.__target_shapes + self.__distractor_shapes
        ov = overlapping_ents(self, self.__sensor_ref, ents, poses)
        nt = len(self.__target_shapes)
        n_overlap_targets = ov[:, :nt].sum(axis=1)
        n_overlap_distractors = ov[:, nt:].sum(axis=1)
        n_overlap = ov.sum(axis=1)
        target_frac_done = n_overlap_targets / nt
        contamination_rate = np.where(n_overlap == 0, 0.0, n_overlap_distractors / np.maximum(n_overlap, 1))
        return target_frac_done * (1 - contamination_rate)
