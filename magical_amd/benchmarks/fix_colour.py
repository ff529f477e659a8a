"""FixColour (mirror of magical/benchmarks/fix_colour.py: Demo, TestColour and TestDynamics branches)."""
import numpy as np

from .. import entities as en
from ..base_env import BaseEnv
from ._scoring import overlapping_ents

DEFAULT_ROBOT_POSE = ((0.368, 0.586), 0.718)
DEFAULT_BLOCK_COLOURS = [en.ShapeColour.GREEN, en.ShapeColour.GREEN, en.ShapeColour.BLUE]
DEFAULT_BLOCK_SHAPES = [en.ShapeType.PENTAGON, en.ShapeType.SQUARE, en.ShapeType.PENTAGON]
DEFAULT_BLOCK_POSES = [((0.289, 0.030), 0.307), ((0.133, -0.561), 1.699), ((-0.336, 0.000), -1.529)]
DEFAULT_REGION_XYHWS = [(-0.032, 0.348, 0.427, 0.468), (0.019, -0.391, 0.460, 0.458), (-0.681, 0.196, 0.498, 0.418)]
DEFAULT_REGION_COLOURS = [en.ShapeColour.GREEN, en.ShapeColour.GREEN, en.ShapeColour.RED]


class FixColourEnv(BaseEnv):
    def __init__(self, rand_colours=False, rand_shapes=False, rand_count=False, rand_layout_minor=False,
                 rand_layout_full=False, **kwargs):
        if rand_shapes or rand_count or rand_layout_minor or rand_layout_full:
            raise NotImplementedError('built: Demo, TestColour, TestDynamics (shape types / counts / layouts need per-env geometry: SURVEY.md §8f)')
        self.rand_colours = rand_colours
        self._keep_env = None
        super().__init__(**kwargs)

    def sample_variation(self, rng, k):   # fix_colour.py:84-94
        if not self.rand_colours:
            return None
        names = en.SHAPE_COLOUR_NAMES
        region_colours = rng.choice(names, size=len(self._blocks)).tolist()
        block_colours = list(region_colours)
        odd_idx = rng.randint(len(block_colours))            # one block gets a colour that is not its region's
        new_col_idx = rng.randint(len(names) - 1)
        if names[new_col_idx] == block_colours[odd_idx]:
            new_col_idx += 1
        block_colours[odd_idx] = names[new_col_idx]
        if self._keep_env is None:
            self._keep_env = np.tile(np.asarray(self._keep, dtype=bool), (self.n_envs, 1))
        self._keep_env[k] = [b == t for b, t in zip(block_colours, region_colours)]
        colours = dict(zip(self._sensors, region_colours))
        colours.update(zip(self._blocks, block_colours))
        return {'colours': colours}

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
        complete = np.ones(poses.shape[0], dtype=bool)
        keep = np.tile(np.asarray(self._keep, dtype=bool), (poses.shape[0], 1)) if self._keep_env is None else self._keep_env[self._scoring_envs]
        for k, sensor in enumerate(self._sensors):
            ov = overlapping_ents(self, sensor, self._blocks, poses)
            expected = np.zeros((poses.shape[0], len(self._blocks)), dtype=bool)
            expected[:, k] = keep[:, k]
            complete &= (ov == expected).all(axis=1)
        return np.where(complete, 1.0, 0.0)
