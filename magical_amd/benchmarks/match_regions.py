"""MatchRegions (mirror of magical/benchmarks/match_regions.py, every rand_* branch)."""
import math

import numpy as np

from .. import entities as en
from .. import geom
from ..base_env import BaseEnv
from ._scoring import overlapping_ents


class MatchRegionsEnv(BaseEnv):
    score_needs_poses = False      # the score is a function of the goal regions' overlap sets (k_score on the device)

    def __init__(self, rand_target_colour=False, rand_shape_type=False, rand_shape_count=False,
                 rand_layout_minor=False, rand_layout_full=False, **kwargs):
        assert not (rand_layout_minor and rand_layout_full)
        if rand_shape_count:     # match_regions.py:33-40
            assert rand_layout_full, 'if shape count is randomised then layout must also be fully randomised'
            assert rand_shape_type, 'if shape count is randomised then shape type must also be randomised'
        self.rand_target_colour, self.rand_layout_minor, self.rand_layout_full = rand_target_colour, rand_layout_minor, rand_layout_full
        self.rand_shape_type, self.rand_shape_count = rand_shape_type, rand_shape_count
        self.variable_worlds = bool(rand_shape_type or rand_shape_count)
        super().__init__(**kwargs)

    def sample_variation(self, rng, k):   # match_regions.py:51-72 (colour, then the region's size), :166-188 (poses)
        if not (self.rand_target_colour or self.rand_layout_minor or self.rand_layout_full or self.rand_shape_type):
            return None
        var = {}
        if self.rand_target_colour:
            # the sensor and the targets take the drawn colour, the distractor groups the remaining ones in SHAPE_COLOURS order
            target_colour = en.draw_choice(rng, en.SHAPE_COLOUR_NAMES)
            distractor_colours = [c for c in en.SHAPE_COLOUR_NAMES if c != target_colour]
            colours = {self.__sensor_ref: target_colour}
            colours.update({s: target_colour for s in self.__target_shapes})
            colours.update({s: distractor_colours[g] for s, g in zip(self.__distractor_shapes, self.__distractor_group)})
            var['colours'] = colours
        if self.rand_layout_minor or self.rand_layout_full:
            hw_bound = self.JITTER_TARGET_BOUND if self.rand_layout_minor else None
            var['goal_hw'] = {self.__sensor_ref: geom.randomise_hw(self.RAND_GOAL_MIN_SIZE, self.RAND_GOAL_MAX_SIZE, rng,
                                                                   current_hw=(0.7, 0.6), linf_bound=hw_bound)}
        # match_regions.py:101-117: how many blocks (targets, then per distractor colour), then their types in the same order
        targets, groups = self.__target_shapes, self.__distractor_by_group
        target_count, distractor_counts = len(targets), [len(g) for g in groups]
        if self.rand_shape_count:
            target_count = rng.randint(1, 2 + 1)
            distractor_counts = [rng.randint(0, 2 + 1) for _ in groups]
            var['enabled'] = {s: i < target_count for i, s in enumerate(targets)}
            for g, n in zip(groups, distractor_counts):
                var['enabled'].update({s: i < n for i, s in enumerate(g)})
        if self.rand_shape_type:
            types_np = en.SHAPE_TYPE_NAMES
            var['shape_types'] = {s: en.draw_choice(rng, types_np) for s in targets[:target_count]}
            for g, n in zip(groups, distractor_counts):
                var['shape_types'].update({s: en.draw_choice(rng, types_np) for s in g[:n]})
        if self.rand_layout_minor or self.rand_layout_full:
            all_ents = (self.__sensor_ref, self._robot, *self.__target_shapes, *self.__distractor_shapes)
            pos_limits, rot_limits = (self.JITTER_POS_BOUND, self.JITTER_ROT_BOUND) if self.rand_layout_minor else (None, None)
            var['randomise_poses'] = (all_ents, dict(rand_pos=True, rand_rot=[False] + [True] * (len(all_ents) - 1),
                                                     rel_pos_linf_limits=pos_limits, rel_rot_limits=rot_limits))
        return var

    def sample_variation_batch(self, brng, env_idx):   # the same draws, all envs at once (batch_rng.py)
        if not (self.rand_target_colour or self.rand_layout_minor or self.rand_layout_full or self.rand_shape_type):
            return None
        from ..batch_rng import uniform_hw
        var, m = {}, brng.m
        sensor, targets, groups = self.__sensor_ref, self.__target_shapes, self.__distractor_by_group
        cid, tid = en.colour_id_of_draw(), en.type_id_of_draw()
        if self.rand_target_colour:
            t = brng.randint(len(en.SHAPE_COLOUR_NAMES))[:, 0]                     # draw index of the target colour
            rows = np.tile(self._default_colours, (m, 1))
            rows[:, [sensor.ent_id] + [s.ent_id for s in targets]] = cid[t][:, None]
            # distractor group g takes the g-th of the remaining colours, in SHAPE_COLOURS order
            for s, g in zip(self.__distractor_shapes, self.__distractor_group):
                rows[:, s.ent_id] = cid[np.where(g < t, g, g + 1)]
            var['colours'] = rows
        if self.rand_layout_minor or self.rand_layout_full:
            hw_bound = self.JITTER_TARGET_BOUND if self.rand_layout_minor else None
            var['goal_hw'] = {sensor.ent_id: uniform_hw(brng.random_sample(2), self.RAND_GOAL_MIN_SIZE, self.RAND_GOAL_MAX_SIZE,
                                                       current_hw=(0.7, 0.6), linf_bound=hw_bound)}
        target_count = np.full(m, len(targets), dtype=np.int32)
        distractor_counts = [np.full(m, len(g), dtype=np.int32) for g in groups]
        if self.rand_shape_count:
            target_count = 1 + brng.randint(2)[:, 0]
            dc = brng.randint(3, count=len(groups))
            distractor_counts = [dc[:, g] for g in range(len(groups))]
            enabled = np.ones((m, len(self._entities)), dtype=bool)
            for i, s in enumerate(targets):
                enabled[:, s.ent_id] = i < target_count
            for g, n in zip(groups, distractor_counts):
                for i, s in enumerate(g):
                    enabled[:, s.ent_id] = i < n
            var['enabled'] = enabled
        if self.rand_shape_type:
            types = np.tile(self._default_shape_types, (m, 1))
            for ents, n in [(targets, target_count)] + list(zip(groups, distractor_counts)):
                if len(ents) == 0:
                    continue
                d = brng.randint(len(en.SHAPE_TYPE_NAMES), counts=n)
                for i, s in enumerate(ents):
                    if i < d.shape[1]:
                        types[:, s.ent_id] = np.where(i < n, tid[d[:, i]], types[:, s.ent_id])
            var['shape_types'] = types
        if self.rand_layout_minor or self.rand_layout_full:
            all_ents = (sensor, self._robot, *targets, *self.__distractor_shapes)
            pos_limits, rot_limits = (self.JITTER_POS_BOUND, self.JITTER_ROT_BOUND) if self.rand_layout_minor else (None, None)
            var['randomise_poses'] = (all_ents, dict(rand_pos=True, rand_rot=[False] + [True] * (len(all_ents) - 1),
                                                     rel_pos_linf_limits=pos_limits, rel_rot_limits=rot_limits))
        return var

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
        if self.rand_shape_count:
            # every block an episode can have (match_regions.py:101-105: up to 2 targets and 2 distractors per colour);
            # which of them exist, their types and their poses are drawn per episode
            distractor_types = [[en.ShapeType.SQUARE] * 2 for _ in distractor_colours]
            distractor_poses = [[(0, 0, 0)] * 2 for _ in distractor_colours]
        self.__target_shapes = [
            self._make_shape(shape_type=st, colour_name=target_colour, init_pos=(x, y), init_angle=a)
            for st, (x, y, a) in zip(target_types, target_poses)]
        self.__distractor_shapes, self.__distractor_group, self.__distractor_by_group = [], [], []
        for group, (colour, types, poses) in enumerate(zip(distractor_colours, distractor_types, distractor_poses)):
            self.__distractor_by_group.append([])
            for st, (x, y, a) in zip(types, poses):
                self.__distractor_group.append(group)
                self.__distractor_shapes.append(self._make_shape(shape_type=st, colour_name=colour, init_pos=(x, y), init_angle=a))
                self.__distractor_by_group[-1].append(self.__distractor_shapes[-1])
        self.add_entities(self.__target_shapes + self.__distractor_shapes)
        self.add_entities([robot])    # last, so it is drawn on top

    def score_on_end_of_traj(self, poses):   # match_regions.py:193-213
        ents = self.__target_shapes + self.__distractor_shapes
        ov = overlapping_ents(self, self.__sensor_ref, ents, poses)
        nt = len(self.__target_shapes)
        n_overlap_targets = ov[:, :nt].sum(axis=1)
        n_overlap_distractors = ov[:, nt:].sum(axis=1)
        n_overlap = ov.sum(axis=1)
        # len(self.__target_shapes) of the env's own episode
        n_targets = self.entity_enabled[self._scoring_envs][:, [s.ent_id for s in self.__target_shapes]].sum(axis=1) if self.variable_worlds else nt
        target_frac_done = n_overlap_targets / n_targets
        contamination_rate = np.where(n_overlap == 0, 0.0, n_overlap_distractors / np.maximum(n_overlap, 1))
        return target_frac_done * (1 - contamination_rate)
