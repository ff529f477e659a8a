"""ClusterColour / ClusterShape (mirror of magical/benchmarks/cluster.py, every rand_* branch)."""
import abc
import enum

import numpy as np

from .. import entities as en
from ..base_env import BaseEnv


class BaseClusterEnv(BaseEnv, abc.ABC):
    class ClusterBy(str, enum.Enum):
        COLOUR = 'colour'
        TYPE = 'type'

    def __init__(self, rand_shape_colour=False, rand_shape_type=False, rand_layout_minor=False, rand_layout_full=False,
                 rand_shape_count=False, cluster_by=ClusterBy.COLOUR, **kwargs):
        assert not (rand_layout_minor and rand_layout_full)
        if rand_shape_count:     # cluster.py:56-63
            assert rand_layout_full, 'if shape count is randomised then layout must also be fully randomised'
            assert rand_shape_type, 'if shape count is randomised then shape type must also be randomised'
            assert rand_shape_colour, 'if shape count is randomised then colour must be randomised too'
        self.rand_layout_minor, self.rand_layout_full = rand_layout_minor, rand_layout_full
        self.cluster_by = cluster_by
        self.rand_shape_colour, self.rand_shape_type, self.rand_shape_count = rand_shape_colour, rand_shape_type, rand_shape_count
        self.variable_worlds = bool(rand_shape_type or rand_shape_count)
        self._class_env = None
        self.TASK_STATE_ATTRS = ('_class_env',)
        super().__init__(**kwargs)

    def sample_variation(self, rng, k):   # cluster.py:81-110 (count, colours, types), :148-161 (poses: robot first, then the blocks)
        if not (self.rand_shape_colour or self.rand_shape_type or self.rand_layout_minor or self.rand_layout_full):
            return None
        var = {}
        ents = self.__shape_ents
        n_shapes = len(ents)
        if self.rand_shape_count:                 # the first n of the (up to 10) blocks
            n_shapes = rng.randint(7, 10 + 1)
            var['enabled'] = {b: i < n_shapes for i, b in enumerate(ents)}
        values = None
        if self.rand_shape_colour:
            # at least one block of each colour, the rest drawn, then shuffled
            names = en.SHAPE_COLOUR_NAMES
            colours = list(names)
            colours.extend(en.draw_choice(rng, names, size=n_shapes - len(colours)) if n_shapes > len(colours) else [])
            rng.shuffle(colours)
            var['colours'] = dict(zip(ents, colours))
            if self.cluster_by == self.ClusterBy.COLOUR:
                values = colours
        if self.rand_shape_type:
            # likewise one of each type
            names = en.SHAPE_TYPE_NAMES
            shape_types = list(names)
            shape_types.extend(en.draw_choice(rng, names, size=n_shapes - len(shape_types)) if n_shapes > len(shape_types) else [])
            rng.shuffle(shape_types)
            var['shape_types'] = dict(zip(ents, shape_types))
            if self.cluster_by == self.ClusterBy.TYPE:
                values = shape_types
        if values is not None or self.rand_shape_count:
            # class of a block = rank of its colour / type among the values present (np.unique sorts them; all four are present)
            if self._class_env is None:
                self._class_env = np.tile(np.concatenate([self.__class_of_block, np.zeros(len(ents) - len(self.__class_of_block), dtype=np.int64)]), (self.n_envs, 1))
            if values is not None:
                order = {c: i for i, c in enumerate(sorted(set(str(v) for v in values)))}
                row = [order[str(c)] for c in values]
                self._class_env[k] = row + [0] * (len(ents) - len(row))
        if self.rand_layout_minor or self.rand_layout_full:
            all_ents = [self._robot, *self.__shape_ents]
            pos_limit, rot_limit = (self.JITTER_POS_BOUND, self.JITTER_ROT_BOUND) if self.rand_layout_minor else (None, None)
            var['randomise_poses'] = (all_ents, dict(rand_pos=True, rand_rot=True, rel_pos_linf_limits=pos_limit, rel_rot_limits=rot_limit))
        return var

    def sample_variation_batch(self, brng, env_idx):   # the same draws, all envs at once (batch_rng.py)
        if not (self.rand_shape_colour or self.rand_shape_type or self.rand_layout_minor or self.rand_layout_full):
            return None
        var, m, ents = {}, brng.m, self.__shape_ents
        n_shapes = np.full(m, len(ents), dtype=np.int32)
        if self.rand_shape_count:
            n_shapes = 7 + brng.randint(10 + 1 - 7)[:, 0]
            var['enabled'] = np.ones((m, len(self._entities)), dtype=bool)
            for i, b in enumerate(ents):
                var['enabled'][:, b.ent_id] = i < n_shapes
        draws = None
        for flag, names, key, table, default, by in (
                (self.rand_shape_colour, en.SHAPE_COLOUR_NAMES, 'colours', en.colour_id_of_draw(), self._default_colours, self.ClusterBy.COLOUR),
                (self.rand_shape_type, en.SHAPE_TYPE_NAMES, 'shape_types', en.type_id_of_draw(), self._default_shape_types, self.ClusterBy.TYPE)):
            if not flag:
                continue
            # one block of each value, the rest drawn, then shuffled: value index of every block of the episode
            k = len(names)
            extra = brng.randint(k, counts=np.maximum(n_shapes - k, 0))
            vals = np.zeros((m, len(ents)), dtype=np.int64)
            vals[:, :k] = np.arange(k)
            for i in range(extra.shape[1]):
                if k + i < len(ents):
                    vals[:, k + i] = extra[:, i]
            perm = brng.shuffle(n_shapes)
            r = np.arange(m)[:, None]
            vals = vals[r, perm[:, :len(ents)]] if perm.shape[1] >= len(ents) else np.pad(vals[r, perm], ((0, 0), (0, len(ents) - perm.shape[1])))
            rows = np.tile(default, (m, 1))
            for i, b in enumerate(ents):
                rows[:, b.ent_id] = np.where(i < n_shapes, table[vals[:, i]], rows[:, b.ent_id])
            var[key] = rows
            if self.cluster_by == by:
                # class of a block = rank of its colour / type name among the values present (np.unique sorts them; all are present)
                rank = np.argsort(np.argsort(np.asarray(names)))
                draws = rank[vals]
        if draws is not None or self.rand_shape_count:
            if self._class_env is None:
                self._class_env = np.tile(np.concatenate([self.__class_of_block, np.zeros(len(ents) - len(self.__class_of_block), dtype=np.int64)]), (self.n_envs, 1))
            if draws is not None:
                self._class_env[env_idx] = np.where(np.arange(len(ents))[None, :] < n_shapes[:, None], draws, 0)
        if self.rand_layout_minor or self.rand_layout_full:
            all_ents = [self._robot, *ents]
            pos_limit, rot_limit = (self.JITTER_POS_BOUND, self.JITTER_ROT_BOUND) if self.rand_layout_minor else (None, None)
            var['randomise_poses'] = (all_ents, dict(rand_pos=True, rand_rot=True, rel_pos_linf_limits=pos_limit, rel_rot_limits=rot_limit))
        return var

    def on_reset(self):   # cluster.py:67-164
        robot = self._make_robot(*self.DEFAULT_ROBOT_POSE)
        colours, shape_types, poses = self.DEFAULT_BLOCK_COLOURS, self.DEFAULT_BLOCK_SHAPES, self.DEFAULT_BLOCK_POSES
        shape_ents = [self._make_shape(shape_type=st, colour_name=c, init_pos=(x, y), init_angle=a)
                      for ((x, y), a), c, st in zip(poses, colours, shape_types)]
        if self.rand_shape_count:
            # every block an episode can have (cluster.py:81-85: 7 to 10); which exist, their colours, types and poses are
            # drawn per episode
            shape_ents = [self._make_shape(shape_type=en.ShapeType.SQUARE, colour_name=en.ShapeColour.RED, init_pos=(0, 0), init_angle=0)
                          for _ in range(10)]
        self.add_entities(shape_ents)
        values = colours if self.cluster_by == self.ClusterBy.COLOUR else shape_types
        c_values_list = np.asarray([v.value for v in values], dtype='object')     # (of the Demo layout; per env where drawn)
        self.__characteristic_values = np.unique(c_values_list)          # sorted, like the reference
        self.__shape_ents = shape_ents
        self.__members = [[k for k, v in enumerate(c_values_list) if v == c_value]
                          for c_value in self.__characteristic_values]
        self.__class_of_block = np.array([list(self.__characteristic_values).index(v) for v in c_values_list], dtype=np.int64)
        self.add_entities([robot])

    def device_score_spec(self):   # the same score on the device (mgx_engine_score_points)
        from .. import _native as nat
        return dict(task=nat.SCORE_CLUSTER, ents=[e.ent_id for e in self.__shape_ents], params=(0.0, 0.0),
                    cls_default=[int(c) for c in self.__class_of_block] + [0] * (len(self.__shape_ents) - len(self.__class_of_block)),
                    n_classes=len(self.__characteristic_values), cls_env=self._class_env)

    def score_on_end_of_traj(self, poses):   # cluster.py:166-216
        pos = poses[:, [e.body for e in self.__shape_ents], :2]            # [M, n_blocks, 2]
        M, n_blocks = pos.shape[0], pos.shape[1]
        nvals = len(self.__characteristic_values)
        cls = np.tile(self.__class_of_block, (M, 1)) if self._class_env is None else self._class_env[self._scoring_envs]   # [M, n_blocks]
        # centroid of every class: np.mean over its members = their sum in block order / their number (adding the
        # exact zeros of non-members changes nothing, so one loop serves envs with different memberships)
        sums = np.zeros((M, nvals, 2))
        counts = np.zeros((M, nvals))
        rows = np.arange(M)
        # blocks the env's episode does not have (rand_shape_count) take no part
        present = self.entity_enabled[self._scoring_envs][:, [e.ent_id for e in self.__shape_ents]] if self.variable_worlds else np.ones((M, n_blocks), dtype=bool)
        for k in range(n_blocks):
            on = present[:, k]
            sums[rows[on], cls[on, k]] += pos[on, k]
            counts[rows[on], cls[on, k]] += 1.0
        with np.errstate(divide='ignore', invalid='ignore'):
            centroids = np.where(counts[:, :, None] > 0, sums / counts[:, :, None], 0.0)      # a class without members: (0, 0), cluster.py:173-175
        min_margin = 2.0
        n_correct = np.zeros(M, dtype=np.int64)
        for k in range(n_blocks):
            centroid_sses = np.sum((pos[:, k, None, :] - centroids)**2, axis=2)     # [M, nvals]
            true_sse = centroid_sses[rows, cls[:, k]]
            others = np.where(np.arange(nvals)[None, :] == cls[:, k, None], np.inf, centroid_sses)
            nearest_bad_centroid = np.min(others, axis=1)
            margin = min_margin * true_sse        # squared distance as margin: reference quirk (cluster.py:203-206)
            n_correct += ((np.sqrt(true_sse) < np.sqrt(nearest_bad_centroid) - margin) & present[:, k]).astype(np.int64)
        frac_correct = n_correct.astype(np.float64) / np.maximum(present.sum(axis=1), 1)
        thresh = 0.75
        return np.maximum(frac_correct - thresh, 0) / (1 - thresh)


class ClusterColourEnv(BaseClusterEnv):   # cluster.py:219-256
    DEFAULT_ROBOT_POSE = ((0.71692, -0.34374), 0.83693)
    DEFAULT_BLOCK_COLOURS = [en.ShapeColour.BLUE, en.ShapeColour.BLUE, en.ShapeColour.BLUE, en.ShapeColour.GREEN,
                             en.ShapeColour.GREEN, en.ShapeColour.RED, en.ShapeColour.YELLOW, en.ShapeColour.YELLOW]
    DEFAULT_BLOCK_SHAPES = [en.ShapeType.CIRCLE, en.ShapeType.STAR, en.ShapeType.SQUARE, en.ShapeType.PENTAGON,
                            en.ShapeType.PENTAGON, en.ShapeType.SQUARE, en.ShapeType.STAR, en.ShapeType.PENTAGON]
    DEFAULT_BLOCK_POSES = [((-0.5147, 0.14149), -0.38871), ((-0.1347, -0.71414), 1.0533), ((-0.74247, -0.097592), 1.1571),
                           ((-0.077363, -0.42964), -0.64379), ((0.51978, 0.1853), -1.1762), ((-0.5278, -0.21642), 2.9356),
                           ((-0.54039, 0.48292), 0.072818), ((-0.16761, 0.64303), -2.3255)]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, cluster_by=BaseClusterEnv.ClusterBy.COLOUR, **kwargs)


class ClusterShapeEnv(BaseClusterEnv):   # cluster.py:259-297
    DEFAULT_ROBOT_POSE = ((0.286, -0.202), -1.878)
    DEFAULT_BLOCK_COLOURS = [en.ShapeColour.YELLOW, en.ShapeColour.BLUE, en.ShapeColour.RED, en.ShapeColour.RED,
                             en.ShapeColour.GREEN, en.ShapeColour.YELLOW, en.ShapeColour.BLUE, en.ShapeColour.GREEN]
    DEFAULT_BLOCK_SHAPES = [en.ShapeType.SQUARE, en.ShapeType.PENTAGON, en.ShapeType.PENTAGON, en.ShapeType.PENTAGON,
                            en.ShapeType.CIRCLE, en.ShapeType.STAR, en.ShapeType.STAR, en.ShapeType.CIRCLE]
    DEFAULT_BLOCK_POSES = [((-0.414, 0.297), -1.731), ((0.068, 0.705), 2.184), ((0.821, 0.220), 0.650),
                           ((-0.461, -0.749), -2.673), ((0.867, -0.149), -2.215), ((-0.785, -0.140), -0.405),
                           ((-0.305, -0.226), 1.341), ((0.758, -0.708), -2.140)]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, cluster_by=BaseClusterEnv.ClusterBy.TYPE, **kwargs)
