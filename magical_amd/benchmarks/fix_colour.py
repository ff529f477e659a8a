"""FixColour (mirror of magical/benchmarks/fix_colour.py, every rand_* branch)."""
import numpy as np

from .. import entities as en
from .. import geom
from ..base_env import BaseEnv
from ._scoring import overlapping_ents

DEFAULT_ROBOT_POSE = ((0.368, 0.586), 0.718)
DEFAULT_BLOCK_COLOURS = [en.ShapeColour.GREEN, en.ShapeColour.GREEN, en.ShapeColour.BLUE]
DEFAULT_BLOCK_SHAPES = [en.ShapeType.PENTAGON, en.ShapeType.SQUARE, en.ShapeType.PENTAGON]
DEFAULT_BLOCK_POSES = [((0.289, 0.030), 0.307), ((0.133, -0.561), 1.699), ((-0.336, 0.000), -1.529)]
DEFAULT_REGION_XYHWS = [(-0.032, 0.348, 0.427, 0.468), (0.019, -0.391, 0.460, 0.458), (-0.681, 0.196, 0.498, 0.418)]
DEFAULT_REGION_COLOURS = [en.ShapeColour.GREEN, en.ShapeColour.GREEN, en.ShapeColour.RED]
MIN_GOAL_SIZE, MAX_GOAL_SIZE = 0.4, 0.5      # fix_colour.py:13-14
MIN_REGIONS, MAX_REGIONS = 2, 3             # fix_colour.py:9-10


class FixColourEnv(BaseEnv):
    score_needs_poses = False      # the score is a function of the goal regions' overlap sets (k_score on the device)

    def __init__(self, rand_colours=False, rand_shapes=False, rand_count=False, rand_layout_minor=False,
                 rand_layout_full=False, **kwargs):
        assert not (rand_layout_minor and rand_layout_full)
        if rand_count:       # fix_colour.py:62-65
            assert rand_layout_full and rand_shapes and rand_colours, 'if count is randomised then layout, shapes and colours must be too'
        self.rand_colours, self.rand_layout_minor, self.rand_layout_full = rand_colours, rand_layout_minor, rand_layout_full
        self.rand_shapes, self.rand_count = rand_shapes, rand_count
        self.variable_worlds = bool(rand_shapes or rand_count)
        self._keep_env = None
        self.TASK_STATE_ATTRS = ('_keep_env',)
        super().__init__(**kwargs)

    def sample_variation(self, rng, k):   # fix_colour.py:84-94 (colours), :102-113 (region sizes), :143-187 (poses)
        if not (self.rand_colours or self.rand_shapes or self.rand_layout_minor or self.rand_layout_full):
            return None
        var = {}
        n_regions = len(self._sensors)
        if self.rand_count:                       # fix_colour.py:78-82: the first n of the (up to MAX_REGIONS = 3) region + block pairs
            n_regions = rng.randint(MIN_REGIONS, MAX_REGIONS + 1)
            var['enabled'] = {e: i < n_regions for ents in (self._sensors, self._blocks) for i, e in enumerate(ents)}
        if self.rand_colours:
            names = en.SHAPE_COLOUR_NAMES
            region_colours = en.draw_choice(rng, names, size=n_regions)
            block_colours = list(region_colours)
            odd_idx = rng.randint(len(block_colours))            # one block gets a colour that is not its region's
            new_col_idx = rng.randint(len(names) - 1)
            if names[new_col_idx] == block_colours[odd_idx]:
                new_col_idx += 1
            block_colours[odd_idx] = names[new_col_idx]
            if self._keep_env is None:
                self._keep_env = np.tile(np.asarray(self._keep, dtype=bool), (self.n_envs, 1))
            keep = [b == t for b, t in zip(block_colours, region_colours)]
            self._keep_env[k] = keep + [False] * (len(self._sensors) - n_regions)
            colours = dict(zip(self._sensors, region_colours))
            colours.update(zip(self._blocks, block_colours))
            var['colours'] = colours
        if self.rand_shapes:                      # fix_colour.py:97-99
            var['shape_types'] = dict(zip(self._blocks, en.draw_choice(rng, en.SHAPE_TYPE_NAMES, size=n_regions)))
        if self.rand_layout_minor or self.rand_layout_full:
            minor = self.rand_layout_minor
            hw_bound = self.JITTER_TARGET_BOUND if minor else None
            var['goal_hw'] = {s: geom.randomise_hw(MIN_GOAL_SIZE, MAX_GOAL_SIZE, rng, current_hw=xyhw[2:], linf_bound=hw_bound)
                              for s, xyhw in zip(self._sensors[:n_regions], DEFAULT_REGION_XYHWS)}
            var['randomise_poses'] = self._pose_stages(minor)
        return var

    def _pose_stages(self, minor):   # fix_colour.py:143-187
        sensors, blocks, robot = self._sensors, self._blocks, self._robot
        pos_limits, rot_limit = (self.JITTER_POS_BOUND, self.JITTER_ROT_BOUND) if minor else (None, None)

        def place_blocks(poses, ent_hw, place):
            # every block goes onto its (moved) region, then each is jittered inside its own region
            for block, sensor in zip(blocks, sensors):
                poses[:, block.ent_id, :2] = poses[:, sensor.ent_id, :2]
            for block, sensor in zip(blocks, sensors):
                lim = np.maximum(0.0, np.minimum(ent_hw[:, sensor.ent_id, 0], ent_hw[:, sensor.ent_id, 1]) / 2 - self.SHAPE_RAD)
                if minor:
                    lim = np.minimum(self.JITTER_POS_BOUND, lim)
                place([block], rand_pos=True, rand_rot=True, rel_pos_linf_limits=lim[:, None],
                      rel_rot_limits=np.full((len(lim), 1), np.nan if rot_limit is None else rot_limit), ignore=[sensor])
        return [((*sensors, robot), dict(rand_pos=True, rand_rot=[False] * len(sensors) + [True], rel_pos_linf_limits=pos_limits,
                                         rel_rot_limits=rot_limit, ignore=blocks)),
                place_blocks]

    def sample_variation_batch(self, brng, env_idx):   # the same draws, all envs at once (batch_rng.py)
        if not (self.rand_colours or self.rand_shapes or self.rand_layout_minor or self.rand_layout_full):
            return None
        from ..batch_rng import uniform_hw
        var, m = {}, brng.m
        sensors, blocks = self._sensors, self._blocks
        cid, tid, n_names = en.colour_id_of_draw(), en.type_id_of_draw(), len(en.SHAPE_COLOUR_NAMES)
        n_regions = np.full(m, len(sensors), dtype=np.int32)
        if self.rand_count:
            n_regions = MIN_REGIONS + brng.randint(MAX_REGIONS + 1 - MIN_REGIONS)[:, 0]
            var['enabled'] = np.ones((m, len(self._entities)), dtype=bool)
            for ents in (sensors, blocks):
                for i, e in enumerate(ents):
                    var['enabled'][:, e.ent_id] = i < n_regions
        if self.rand_colours:
            region = brng.randint(n_names, counts=n_regions)                   # draw indices, [m, <= 3]
            block = region.copy()
            odd = np.zeros(m, dtype=np.int64)                                   # rng.randint(n_regions): the bound differs between envs
            for n in np.unique(n_regions):
                rows = np.nonzero(n_regions == n)[0]
                odd[rows] = brng.randint(int(n), rows=rows)[:, 0]
            new = brng.randint(n_names - 1)[:, 0].astype(np.int64)
            r = np.arange(m)
            new = np.where(new == block[r, odd], new + 1, new)
            block[r, odd] = new
            if self._keep_env is None:
                self._keep_env = np.tile(np.asarray(self._keep, dtype=bool), (self.n_envs, 1))
            keep = np.zeros((m, len(sensors)), dtype=bool)
            rows_c = np.tile(self._default_colours, (m, 1))
            for i in range(len(sensors)):
                if i < region.shape[1]:
                    on = i < n_regions
                    keep[:, i] = on & (block[:, i] == region[:, i])
                    rows_c[:, sensors[i].ent_id] = np.where(on, cid[region[:, i]], rows_c[:, sensors[i].ent_id])
                    rows_c[:, blocks[i].ent_id] = np.where(on, cid[block[:, i]], rows_c[:, blocks[i].ent_id])
            self._keep_env[env_idx] = keep
            var['colours'] = rows_c
        if self.rand_shapes:
            d = brng.randint(len(en.SHAPE_TYPE_NAMES), counts=n_regions)
            rows_t = np.tile(self._default_shape_types, (m, 1))
            for i, b in enumerate(blocks):
                if i < d.shape[1]:
                    rows_t[:, b.ent_id] = np.where(i < n_regions, tid[d[:, i]], rows_t[:, b.ent_id])
            var['shape_types'] = rows_t
        if self.rand_layout_minor or self.rand_layout_full:
            minor = self.rand_layout_minor
            hw_bound = self.JITTER_TARGET_BOUND if minor else None
            u = brng.random_sample(counts=2 * n_regions)                        # (h, w) of the episode's regions, in order
            var['goal_hw'] = {}
            for i, (sensor, xyhw) in enumerate(zip(sensors, DEFAULT_REGION_XYHWS)):
                if 2 * i + 1 < u.shape[1]:
                    h, w = uniform_hw(u[:, 2 * i:2 * i + 2], MIN_GOAL_SIZE, MAX_GOAL_SIZE, current_hw=xyhw[2:], linf_bound=hw_bound)
                    on = i < n_regions
                    var['goal_hw'][sensor.ent_id] = (np.where(on, h, xyhw[2]), np.where(on, w, xyhw[3]))
            var['randomise_poses'] = self._pose_stages(minor)
        return var

    def on_reset(self):   # fix_colour.py:69-141
        robot = self._make_robot(*DEFAULT_ROBOT_POSE)
        self._sensors = [en.GoalRegion(*xyhw, colour) for colour, xyhw in zip(DEFAULT_REGION_COLOURS, DEFAULT_REGION_XYHWS)]
        self.add_entities(self._sensors)
        self._blocks, self._keep = [], []
        for bshape, bcol, tcol, (bpos, bangle) in zip(DEFAULT_BLOCK_SHAPES, DEFAULT_BLOCK_COLOURS, DEFAULT_REGION_COLOURS,
                                                      DEFAULT_BLOCK_POSES):
            self._blocks.append(self._make_shape(shape_type=bshape, colour_name=bcol, init_pos=bpos, init_angle=bangle))
            self._keep.append(bcol == tcol)   # region k must end up holding exactly its block (True) or nothing (False)
        self.add_entities(self._blocks)
        self.add_entities([robot])

    def score_on_end_of_traj(self, poses):   # fix_colour.py:193-202: list(overlap_ents) == expected, per region
        M = len(self._scoring_envs) if poses is None else poses.shape[0]      # poses None: the overlap sets come from the device
        complete = np.ones(M, dtype=bool)
        keep = np.tile(np.asarray(self._keep, dtype=bool), (M, 1)) if self._keep_env is None else self._keep_env[self._scoring_envs]
        for k, sensor in enumerate(self._sensors):
            ov = overlapping_ents(self, sensor, self._blocks, poses)
            expected = np.zeros((M, len(self._blocks)), dtype=bool)
            expected[:, k] = keep[:, k]
            ok = (ov == expected).all(axis=1)
            if self.variable_worlds:              # a region the episode does not have asks for nothing
                ok |= ~self.entity_enabled[self._scoring_envs, sensor.ent_id]
            complete &= ok
        return np.where(complete, 1.0, 0.0)
