"""FindDupe (mirror of magical/benchmarks/find_dupe.py, every rand_* branch)."""
import numpy as np

from .. import entities as en
from .. import geom
from ..base_env import BaseEnv
from ._scoring import overlapping_ents

DEFAULT_QUERY_COLOUR = en.ShapeColour.YELLOW
DEFAULT_QUERY_SHAPE = en.ShapeType.PENTAGON
DEFAULT_OUT_BLOCK_SHAPES = [en.ShapeType.PENTAGON, en.ShapeType.CIRCLE, en.ShapeType.CIRCLE, en.ShapeType.SQUARE,
                            en.ShapeType.STAR, DEFAULT_QUERY_SHAPE]
DEFAULT_OUT_BLOCK_COLOURS = [en.ShapeColour.GREEN, en.ShapeColour.RED, en.ShapeColour.RED, en.ShapeColour.YELLOW,
                             en.ShapeColour.BLUE, DEFAULT_QUERY_COLOUR]
DEFAULT_OUT_BLOCK_POSES = [((-0.066751, 0.7552), -2.9266), ((-0.05195, 0.31468), 1.5418), ((0.57528, -0.46865), -2.2141),
                           ((0.40594, -0.74977), 0.24582), ((0.45254, 0.3681), -1.0834), ((0.76849, -0.10652), 0.10028)]
DEFAULT_ROBOT_POSE = ((-0.57, 0.25), 3.83)
DEFAULT_TARGET_REGION_XYHW = (-0.72, -0.22, 0.67, 0.72)
DEFAULT_QUERY_BLOCK_POSE = ((-0.33, -0.49), -0.51)


class FindDupeEnv(BaseEnv):
    score_needs_poses = False      # the score is a function of the goal regions' overlap sets (k_score on the device)

    def __init__(self, rand_colours=False, rand_shapes=False, rand_count=False, rand_layout_minor=False,
                 rand_layout_full=False, **kwargs):
        assert not (rand_layout_minor and rand_layout_full)
        if rand_count:       # find_dupe.py:67-70
            assert rand_layout_full and rand_shapes and rand_colours, 'if count is randomised then layout, shapes and colours must be too'
        self.rand_colours, self.rand_layout_minor, self.rand_layout_full = rand_colours, rand_layout_minor, rand_layout_full
        self.rand_shapes, self.rand_count = rand_shapes, rand_count
        self.variable_worlds = bool(rand_shapes or rand_count)
        self._is_target_env = None
        self.TASK_STATE_ATTRS = ('_is_target_env',)
        super().__init__(**kwargs)

    def sample_variation(self, rng, k):   # find_dupe.py:84-100 (count, colours, shapes), :101-112 (region size), :157-196 (poses)
        if not (self.rand_colours or self.rand_shapes or self.rand_layout_minor or self.rand_layout_full):
            return None
        var = {}
        outside = self.__outside_blocks
        n_out_blocks = len(outside)
        if self.rand_count:
            # 1..5 random blocks plus one that always matches the query: the first n of the six outside blocks
            n_out_blocks = rng.randint(1, 5 + 1) + 1
            var['enabled'] = {b: i < n_out_blocks for i, b in enumerate(outside)}
        n_distractors = n_out_blocks - 1
        query_colour, query_shape = DEFAULT_QUERY_COLOUR.value, DEFAULT_QUERY_SHAPE.value
        out_block_colours = [c.value for c in DEFAULT_OUT_BLOCK_COLOURS]
        out_block_shapes = [t.value for t in DEFAULT_OUT_BLOCK_SHAPES]
        if self.rand_colours:
            names = en.SHAPE_COLOUR_NAMES
            query_colour = en.draw_choice(rng, names)
            out_block_colours = en.draw_choice(rng, names, size=n_distractors)
            out_block_colours.append(query_colour)               # the last outside block always matches the query
            colours = {self.__sensor_ref: query_colour, self.__all_blocks[0]: query_colour}
            colours.update(zip(outside, out_block_colours))
            var['colours'] = colours
        if self.rand_shapes:
            query_shape = en.draw_choice(rng, en.SHAPE_TYPE_NAMES)
            out_block_shapes = en.draw_choice(rng, en.SHAPE_TYPE_NAMES, size=n_distractors)
            out_block_shapes.append(query_shape)
            var['shape_types'] = {self.__all_blocks[0]: query_shape}
            var['shape_types'].update(zip(outside, out_block_shapes))
        if self.rand_colours or self.rand_shapes:
            if self._is_target_env is None:
                self._is_target_env = np.tile(self.__is_target, (self.n_envs, 1))
            # __all_blocks = [query block, *outside blocks]; a block is a target iff it has the query's colour and shape
            row = [True] + [c == query_colour and t == query_shape for c, t in zip(out_block_colours, out_block_shapes)]
            self._is_target_env[k] = row + [False] * (len(self.__all_blocks) - len(row))
        if self.rand_layout_minor or self.rand_layout_full:
            minor = self.rand_layout_minor
            var['goal_hw'] = {self.__sensor_ref: geom.randomise_hw(self.RAND_GOAL_MIN_SIZE, self.RAND_GOAL_MAX_SIZE, rng,
                                                                   current_hw=DEFAULT_TARGET_REGION_XYHW[2:],
                                                                   linf_bound=self.JITTER_TARGET_BOUND if minor else None)}
            var['randomise_poses'] = self._pose_stages(minor)
        return var

    def _pose_stages(self, minor):   # find_dupe.py:157-196
        sensor, query = self.__sensor_ref, self.__all_blocks[0]
        all_ents = (sensor, self._robot, *self.__outside_blocks)
        pos_limits, rot_limit = (self.JITTER_POS_BOUND, self.JITTER_ROT_BOUND) if minor else (None, None)

        def place_query(poses, ent_hw, place):
            # the query block goes onto the (moved) region and is then jittered so that it stays mostly inside it
            poses[:, query.ent_id, :2] = poses[:, sensor.ent_id, :2]
            lim = np.maximum(0.0, np.minimum(ent_hw[:, sensor.ent_id, 0], ent_hw[:, sensor.ent_id, 1]) / 2 - self.SHAPE_RAD / 2)
            if minor:
                lim = np.minimum(self.JITTER_POS_BOUND, lim)
            place([query], rand_pos=True, rand_rot=True, rel_pos_linf_limits=lim[:, None],
                  rel_rot_limits=np.full((len(lim), 1), np.nan if rot_limit is None else rot_limit), ignore=[sensor])
        return [(all_ents, dict(rand_pos=True, rand_rot=[False] + [True] * (len(all_ents) - 1), rel_pos_linf_limits=pos_limits,
                                rel_rot_limits=rot_limit, ignore=[query])),
                place_query]

    def sample_variation_batch(self, brng, env_idx):   # the same draws, all envs at once (batch_rng.py)
        if not (self.rand_colours or self.rand_shapes or self.rand_layout_minor or self.rand_layout_full):
            return None
        from ..batch_rng import uniform_hw
        var, m = {}, brng.m
        outside, query, sensor = self.__outside_blocks, self.__all_blocks[0], self.__sensor_ref
        cid, tid = en.colour_id_of_draw(), en.type_id_of_draw()
        n_out = np.full(m, len(outside), dtype=np.int32)
        if self.rand_count:
            n_out = brng.randint(5)[:, 0] + 1 + 1                   # rng.randint(1, 5 + 1) + 1
            var['enabled'] = np.ones((m, len(self._entities)), dtype=bool)
            for i, b in enumerate(outside):
                var['enabled'][:, b.ent_id] = i < n_out
        n_distractors = n_out - 1
        # native ids of the query's and of every outside block's colour / shape: the Demo's, or this episode's draws
        q_col = np.full(m, en.COLOUR_ID[DEFAULT_QUERY_COLOUR], dtype=np.int64)
        q_typ = np.full(m, en.SHAPE_TYPE_ID[DEFAULT_QUERY_SHAPE], dtype=np.int32)
        o_col = np.tile(np.array([en.COLOUR_ID[c] for c in DEFAULT_OUT_BLOCK_COLOURS], dtype=np.int64), (m, 1))
        o_typ = np.tile(np.array([en.SHAPE_TYPE_ID[t] for t in DEFAULT_OUT_BLOCK_SHAPES], dtype=np.int32), (m, 1))
        last = np.arange(m), n_out - 1                                # the last outside block of the episode always matches the query
        if self.rand_colours:
            q_col = cid[brng.randint(len(en.SHAPE_COLOUR_NAMES))[:, 0]]
            d = brng.randint(len(en.SHAPE_COLOUR_NAMES), counts=n_distractors)
            for i in range(min(d.shape[1], len(outside))):
                o_col[:, i] = np.where(i < n_distractors, cid[d[:, i]], o_col[:, i])
            o_col[last] = q_col
            rows = np.tile(self._default_colours, (m, 1))
            rows[:, sensor.ent_id] = q_col
            rows[:, query.ent_id] = q_col
            for i, b in enumerate(outside):
                rows[:, b.ent_id] = np.where(i < n_out, o_col[:, i], rows[:, b.ent_id])
            var['colours'] = rows
        if self.rand_shapes:
            q_typ = tid[brng.randint(len(en.SHAPE_TYPE_NAMES))[:, 0]]
            d = brng.randint(len(en.SHAPE_TYPE_NAMES), counts=n_distractors)
            for i in range(min(d.shape[1], len(outside))):
                o_typ[:, i] = np.where(i < n_distractors, tid[d[:, i]], o_typ[:, i])
            o_typ[last] = q_typ
            rows = np.tile(self._default_shape_types, (m, 1))
            rows[:, query.ent_id] = q_typ
            for i, b in enumerate(outside):
                rows[:, b.ent_id] = np.where(i < n_out, o_typ[:, i], rows[:, b.ent_id])
            var['shape_types'] = rows
        if self.rand_colours or self.rand_shapes:
            if self._is_target_env is None:
                self._is_target_env = np.tile(self.__is_target, (self.n_envs, 1))
            # __all_blocks = [query block, *outside blocks]; a block is a target iff it has the query's colour and shape
            row = np.zeros((m, len(self.__all_blocks)), dtype=bool)
            row[:, 0] = True
            for i in range(len(outside)):
                row[:, 1 + i] = (i < n_out) & (o_col[:, i] == q_col) & (o_typ[:, i] == q_typ)
            self._is_target_env[env_idx] = row
        if self.rand_layout_minor or self.rand_layout_full:
            minor = self.rand_layout_minor
            var['goal_hw'] = {sensor.ent_id: uniform_hw(brng.random_sample(2), self.RAND_GOAL_MIN_SIZE, self.RAND_GOAL_MAX_SIZE,
                                                       current_hw=DEFAULT_TARGET_REGION_XYHW[2:], linf_bound=self.JITTER_TARGET_BOUND if minor else None)}
            var['randomise_poses'] = self._pose_stages(minor)
        return var

    def on_reset(self):   # find_dupe.py:72-155
        robot = self._make_robot(*DEFAULT_ROBOT_POSE)
        sensor = en.GoalRegion(*DEFAULT_TARGET_REGION_XYHW, DEFAULT_QUERY_COLOUR)
        self.add_entities([sensor])
        self.__sensor_ref = sensor
        outside_blocks, targets = [], []
        # (with rand_count the six outside blocks are the most an episode can have: find_dupe.py:84-87)
        for bshape, bcol, (bpos, bangle) in zip(DEFAULT_OUT_BLOCK_SHAPES, DEFAULT_OUT_BLOCK_COLOURS, DEFAULT_OUT_BLOCK_POSES):
            blk = self._make_shape(shape_type=bshape, colour_name=bcol, init_pos=bpos, init_angle=bangle)
            outside_blocks.append(blk)
            if bcol == DEFAULT_QUERY_COLOUR and bshape == DEFAULT_QUERY_SHAPE:
                targets.append(blk)
        self.add_entities(outside_blocks)
        query_block = self._make_shape(shape_type=DEFAULT_QUERY_SHAPE, colour_name=DEFAULT_QUERY_COLOUR,
                                       init_pos=DEFAULT_QUERY_BLOCK_POSE[0], init_angle=DEFAULT_QUERY_BLOCK_POSE[1])
        targets.append(query_block)
        self.add_entities([query_block])
        self.add_entities([robot])
        self.__all_blocks = [query_block, *outside_blocks]
        self.__outside_blocks = outside_blocks
        self.__is_target = np.array([b in targets for b in self.__all_blocks])

    def score_on_end_of_traj(self, poses):   # find_dupe.py:203-216
        ov = overlapping_ents(self, self.__sensor_ref, self.__all_blocks, poses)
        is_target = self.__is_target[None, :] if self._is_target_env is None else self._is_target_env[self._scoring_envs]
        n_overlap_targets = (ov & is_target).sum(axis=1)
        n_overlap_distractors = (ov & ~is_target).sum(axis=1)
        n_overlap = ov.sum(axis=1)
        have_two_shapes = (n_overlap_targets >= 2).astype(np.float64)
        contamination_rate = np.where(n_overlap == 0, 0.0, n_overlap_distractors / np.maximum(n_overlap, 1))
        return have_two_shapes * (1 - contamination_rate)
