"""MakeLine (mirror of magical/benchmarks/make_line.py, every rand_* branch)."""
import numpy as np

from .. import entities as en
from ..base_env import BaseEnv
from ._scoring import row_norm

INLIER_RAD_MULT = 1.5
MAX_SEP_RADS = 3.5
MIN_BLOCKS, MAX_BLOCKS = 3, 4      # make_line.py:12-13
DEFAULT_ROBOT_POSE = ((0.702, -0.255), 0.347)
DEFAULT_BLOCK_COLOURS = [en.ShapeColour.BLUE, en.ShapeColour.YELLOW, en.ShapeColour.RED, en.ShapeColour.GREEN]
DEFAULT_BLOCK_SHAPES = [en.ShapeType.STAR, en.ShapeType.CIRCLE, en.ShapeType.STAR, en.ShapeType.PENTAGON]
DEFAULT_BLOCK_POSES = [((0.790, -0.820), -0.721), ((-0.177, 0.383), -1.733),
                       ((-0.051, -0.128), 2.696), ((-0.292, -0.745), -0.159)]


def longest_line(points, inlier_dist, max_separation):
    """make_line.py:31-71 for one env: exhaustive pair lines, inliers within inlier_dist, longest run of
    inliers whose neighbours along the line are at most max_separation apart."""
    npts = len(points)
    best = min(1, npts)
    for i in range(npts - 1):
        for j in range(i + 1, npts):
            offs = points - points[i][None]
            pj_off = offs[j]
            pj_unit = pj_off / np.linalg.norm(pj_off)
            proj_lens = np.squeeze(offs @ pj_unit[:, None], axis=1)
            dists = np.linalg.norm(offs - proj_lens[:, None] * pj_unit, axis=1)
            inliers = np.nonzero(dists <= inlier_dist)[0]
            if len(inliers) <= best:
                continue
            lens = np.sort(proj_lens[inliers])
            close = np.abs(np.diff(lens)) <= max_separation
            run = longest = 0
            for flag in close:
                run = run + 1 if flag else 0
                longest = max(longest, run)
            if longest + 1 > best:
                best = longest + 1
    return best


def longest_line_batch(points, inlier_dist, max_separation):
    """longest_line() for M envs at once: points float64[M, n, 2] -> int[M].  Every arithmetic step is the same numpy
    primitive on the same operands as in the per-env code (1-D norm through row_norm = BLAS ddot, the projection through
    np.matmul's small-matrix kernel, the point-line distances elementwise), so the results are bit-identical
    (tests/test_host_api.py checks that against longest_line on random and degenerate inputs)."""
    M, npts = points.shape[0], points.shape[1]
    best = np.full(M, min(1, npts), dtype=np.int64)
    for i in range(npts - 1):
        for j in range(i + 1, npts):
            offs = points - points[:, i, None, :]                               # [M, n, 2]
            pj_off = offs[:, j]                                                 # [M, 2]
            with np.errstate(divide='ignore', invalid='ignore'):
                pj_unit = pj_off / row_norm(pj_off)[:, None]
                proj_lens = np.squeeze(offs @ pj_unit[:, :, None], axis=2)      # [M, n]
                dists = np.linalg.norm(offs - proj_lens[:, :, None] * pj_unit[:, None, :], axis=2)
                inlier = dists <= inlier_dist                                   # NaN (coincident points) -> no inliers
            # sorted projections of the inliers, the others pushed to the end
            lens = np.sort(np.where(inlier, proj_lens, np.inf), axis=1)
            n_in = inlier.sum(axis=1)
            with np.errstate(invalid='ignore'):
                close = np.abs(np.diff(lens, axis=1)) <= max_separation         # inf - inf = NaN -> False
            close &= np.arange(1, npts)[None, :] < n_in[:, None]                # only gaps between two inliers
            run = np.zeros(M, dtype=np.int64)
            longest = np.zeros(M, dtype=np.int64)
            for k in range(npts - 1):
                run = np.where(close[:, k], run + 1, 0)
                longest = np.maximum(longest, run)
            cand = np.where(n_in > best, longest + 1, 0)
            best = np.maximum(best, cand)
    return best


class MakeLineEnv(BaseEnv):
    def __init__(self, rand_colours=False, rand_shapes=False, rand_count=False, rand_layout_minor=False,
                 rand_layout_full=False, **kwargs):
        assert not (rand_layout_minor and rand_layout_full)
        if rand_count:       # make_line.py:86-89
            assert rand_layout_full and rand_shapes and rand_colours, 'if count is randomised then layout, shapes and colours must be too'
        self.rand_colours, self.rand_layout_minor, self.rand_layout_full = rand_colours, rand_layout_minor, rand_layout_full
        self.rand_shapes, self.rand_count = rand_shapes, rand_count
        self.variable_worlds = bool(rand_shapes or rand_count)
        super().__init__(**kwargs)
        self.inlier_dist = self.SHAPE_RAD * INLIER_RAD_MULT
        self.max_sep = self.SHAPE_RAD * MAX_SEP_RADS

    def on_reset(self):   # make_line.py:91-122
        robot = self._make_robot(*DEFAULT_ROBOT_POSE)
        self._blocks = [self._make_shape(shape_type=s, colour_name=c, init_pos=p, init_angle=a)
                        for s, c, (p, a) in zip(DEFAULT_BLOCK_SHAPES, DEFAULT_BLOCK_COLOURS, DEFAULT_BLOCK_POSES)]
        self.add_entities(self._blocks)
        self.add_entities([robot])

    def sample_variation(self, rng, k):   # make_line.py:105-107 (colours), :124-139 (poses: robot first, then the blocks)
        if not (self.rand_colours or self.rand_shapes or self.rand_layout_minor or self.rand_layout_full):
            return None
        var = {}
        n_blocks = len(self._blocks)
        if self.rand_count:                       # make_line.py:100-102: the first n of the (up to MAX_BLOCKS = 4) blocks
            n_blocks = rng.randint(MIN_BLOCKS, MAX_BLOCKS + 1)
            var['enabled'] = {b: i < n_blocks for i, b in enumerate(self._blocks)}
        if self.rand_colours:
            block_colours = en.draw_choice(rng, en.SHAPE_COLOUR_NAMES, size=n_blocks)
            var['colours'] = dict(zip(self._blocks, block_colours))
        if self.rand_shapes:                      # make_line.py:108-110
            var['shape_types'] = dict(zip(self._blocks, en.draw_choice(rng, en.SHAPE_TYPE_NAMES, size=n_blocks)))
        if self.rand_layout_minor or self.rand_layout_full:
            all_ents = (self._robot, *self._blocks)
            pos_limits, rot_limit = (self.JITTER_POS_BOUND, self.JITTER_ROT_BOUND) if self.rand_layout_minor else (None, None)
            var['randomise_poses'] = (all_ents, dict(rand_pos=True, rand_rot=True, rel_pos_linf_limits=pos_limits, rel_rot_limits=rot_limit))
        return var

    def sample_variation_batch(self, brng, env_idx):   # the same draws, all envs at once (batch_rng.py)
        if not (self.rand_colours or self.rand_shapes or self.rand_layout_minor or self.rand_layout_full):
            return None
        var, m, blocks = {}, brng.m, self._blocks
        n_blocks = np.full(m, len(blocks), dtype=np.int32)
        if self.rand_count:
            n_blocks = MIN_BLOCKS + brng.randint(MAX_BLOCKS + 1 - MIN_BLOCKS)[:, 0]
            var['enabled'] = np.ones((m, len(self._entities)), dtype=bool)
            for i, b in enumerate(blocks):
                var['enabled'][:, b.ent_id] = i < n_blocks
        for flag, key, n_choices, table, default in ((self.rand_colours, 'colours', len(en.SHAPE_COLOUR_NAMES), en.colour_id_of_draw(), self._default_colours),
                                                     (self.rand_shapes, 'shape_types', len(en.SHAPE_TYPE_NAMES), en.type_id_of_draw(), self._default_shape_types)):
            if flag:
                d = brng.randint(n_choices, counts=n_blocks)
                rows = np.tile(default, (m, 1))
                for i, b in enumerate(blocks):
                    if i < d.shape[1]:          # (no env of this reset drew that many blocks otherwise)
                        rows[:, b.ent_id] = np.where(i < n_blocks, table[d[:, i]], rows[:, b.ent_id])
                var[key] = rows
        if self.rand_layout_minor or self.rand_layout_full:
            all_ents = (self._robot, *blocks)
            pos_limits, rot_limit = (self.JITTER_POS_BOUND, self.JITTER_ROT_BOUND) if self.rand_layout_minor else (None, None)
            var['randomise_poses'] = (all_ents, dict(rand_pos=True, rand_rot=True, rel_pos_linf_limits=pos_limits, rel_rot_limits=rot_limit))
        return var

    def device_score_spec(self):   # the same score on the device (mgx_engine_score_points)
        from .. import _native as nat
        return dict(task=nat.SCORE_LINE, ents=[b.ent_id for b in self._blocks], params=(self.inlier_dist, self.max_sep))

    def score_on_end_of_traj(self, poses):   # make_line.py:142-152
        bodies = [b.body for b in self._blocks]
        if not self.variable_worlds:
            return self._line_score(poses, bodies)
        # the episode's blocks are the first n of the list: score the envs count by count
        n_env = self.entity_enabled[self._scoring_envs][:, [b.ent_id for b in self._blocks]].sum(axis=1)
        score = np.zeros(poses.shape[0], dtype=np.float64)
        for n in np.unique(n_env):
            sel = np.nonzero(n_env == n)[0]
            score[sel] = self._line_score(poses[sel], bodies[:int(n)])
        return score

    def _line_score(self, poses, bodies):
        max_line_len = len(bodies)
        min_line_len = max(max_line_len - 2, 2)
        points = np.ascontiguousarray(poses[:, bodies, :2], dtype='float64')
        line_len = longest_line_batch(points, self.inlier_dist, self.max_sep)
        return np.maximum(line_len - min_line_len, 0) / (max_line_len - min_line_len)
