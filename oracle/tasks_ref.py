"""Task layouts + score_on_end_of_traj restatements.

TEST INFRASTRUCTURE.  One class per reference task file, with every rand_* branch of its on_reset() (the Demo and all
Test* variants).  `slots` lists the episode's entities in the order of the product's entity list for the task (None
where this episode has no such entity), so that tests can compare entity by entity.  Scores follow the reference's
float64 numpy/Python operation order so they can be compared bit-for-bit with
the product's host scoring.
"""
import itertools as it
import math

import numpy as np

from .entities_ref import RAND_SHAPE_TYPES, SHAPE_COLOURS, GoalRegion, Robot, Shape

ROBOT_RAD = 0.2          # base_env.py:62
ROBOT_MASS = 1.0         # base_env.py:63
SHAPE_RAD = ROBOT_RAD * 0.6   # base_env.py:64
JITTER_PCT = 0.05             # base_env.py:73-75
JITTER_POS_BOUND = 1 * JITTER_PCT / 2.0
JITTER_ROT_BOUND = JITTER_PCT * np.pi
RAND_GOAL_MIN_SIZE, RAND_GOAL_MAX_SIZE = 0.5, 0.8          # base_env.py:68-69
JITTER_TARGET_BOUND = JITTER_PCT * (RAND_GOAL_MAX_SIZE - RAND_GOAL_MIN_SIZE) / 2   # base_env.py:76


def randomise_hw(min_side, max_side, rng, current_hw=None, linf_bound=None):
    """geom.py:344-360."""
    minima = np.asarray((min_side, min_side))
    maxima = np.asarray((max_side, max_side))
    if linf_bound is not None:
        current_hw = np.asarray(current_hw)
        minima = np.maximum(minima, current_hw - linf_bound)
        maxima = np.minimum(maxima, current_hw + linf_bound)
    h, w = rng.uniform(minima, maxima)
    return h, w


def _robot(pos, angle):
    return Robot(radius=ROBOT_RAD, init_pos=tuple(pos), init_angle=angle, mass=ROBOT_MASS)


def _shape(shape_type, colour, pos, angle):
    return Shape(shape_type=shape_type, colour_name=colour, shape_size=SHAPE_RAD,
                 init_pos=tuple(pos), init_angle=angle)


class TaskRef:
    name = None
    ep_len = None

    def __init__(self, world, rng=None, replay=None, **flags):
        self.world = world
        self.rng = rng                 # the env's np.random.RandomState (base_env.py:133-140)
        self.flags = flags             # rand_* keyword arguments of the reference constructor that are True
        self.choices = {}              # every random choice of this reset, by name ...
        self.replay = replay           # ... or the choices to rebuild the same world from, without touching the rng
        self.on_reset()

    def draw(self, name, fn):
        if self.replay is not None:
            return self.replay[name]
        self.choices[name] = v = fn()
        return v

    def main_pose(self, ent):
        return tuple(float(v) for v in self.world_pose(ent.bodies[0]))

    def jitter(self, entities, **kw):
        """pm_randomise_all_poses on this (scratch) world; the env then rebuilds the world AT the drawn poses
        (placement_ref.py explains why)."""
        from .placement_ref import randomise_all_poses
        return self.draw('poses', lambda: randomise_all_poses(self.world, entities, [-1, 1, -1, 1], self.rng, **kw))

    def block_pos(self, ent):
        return self.world_pose(ent.shape_body)[:2]

    def world_pose(self, body):
        import ctypes as C
        w = self.world
        n = w.L.ref_nbodies(w.h)
        buf = (C.c_double * (9 * n))()
        w.L.ref_get_bodies(w.h, buf)
        return (buf[9 * body], buf[9 * body + 1], buf[9 * body + 2])


class MoveToCornerRef(TaskRef):
    """benchmarks/move_to_corner.py:31-75."""
    name, ep_len = 'MoveToCorner', 80

    def on_reset(self):
        w = self.world
        robot_pose, shape_pose = (0.4, -0.0, 0.55 * math.pi), (0.1, -0.65, 0.13 * math.pi)
        if self.replay is not None and 'poses' in self.replay:
            robot_pose, shape_pose = self.replay['poses']
        self.robot = w.add(_robot(robot_pose[:2], robot_pose[2]))
        shape_colour = 'red'
        if self.flags.get('rand_shape_colour'):          # move_to_corner.py:42-44
            shape_colour = self.draw('colour', lambda: self.rng.choice(np.asarray(SHAPE_COLOURS, dtype='object')))
        shape_type = 'square'
        if self.flags.get('rand_shape_type'):            # move_to_corner.py:45-47
            shape_type = self.draw('shape_type', lambda: self.rng.choice(np.asarray(RAND_SHAPE_TYPES, dtype='object')))
        self.shape = w.add(_shape(shape_type, shape_colour, shape_pose[:2], shape_pose[2]))
        self.slots = [self.robot, self.shape]
        if self.flags.get('rand_poses') and self.replay is None:    # move_to_corner.py:56-63
            self.jitter([self.robot, self.shape], rel_pos_linf_limits=JITTER_POS_BOUND, rel_rot_limits=JITTER_ROT_BOUND)

    def score_on_end_of_traj(self):
        robot_pos = np.asarray(self.block_pos(self.shape))
        dist = np.linalg.norm(np.asarray([-1.0, 1.0]) - robot_pos)
        succeed_dist = np.sqrt(2) / 2
        furthest_dist = np.sqrt(2)
        drange = (furthest_dist - succeed_dist)
        score = min(1.0, max(0.0, furthest_dist - dist) / drange)
        return score

    def debug_shaped_reward(self):
        """move_to_corner.py:84-98."""
        shape_pos = np.asarray(self.block_pos(self.shape))
        shape_to_corner_dist = np.linalg.norm(shape_pos - np.array((0, 1)))
        robot_pos = np.asarray(self.world_pose(self.robot.robot_body)[:2])
        robot_to_shape_dist = np.linalg.norm(robot_pos - shape_pos)
        shaping = -shape_to_corner_dist / 5 - max(robot_to_shape_dist, 0.2) / 20
        return shaping + self.score_on_end_of_traj()


class MoveToRegionRef(TaskRef):
    """benchmarks/move_to_region.py:9-11,30-94."""
    name, ep_len = 'MoveToRegion', 40

    def on_reset(self):
        w = self.world
        minor, full = self.flags.get('rand_poses_minor'), self.flags.get('rand_poses_full')
        gx, gy, gh, gw = -0.62, -0.17, 0.76, 0.75
        if minor or full:                                 # move_to_region.py:32-45
            gh, gw = self.draw('goal_hw', lambda: randomise_hw(RAND_GOAL_MIN_SIZE, RAND_GOAL_MAX_SIZE, self.rng, current_hw=(gh, gw),
                                                               linf_bound=JITTER_TARGET_BOUND if minor else None))
        goal_colour = 'blue'
        if self.flags.get('rand_goal_colour'):            # move_to_region.py:47-51
            goal_colour = self.draw('colour', lambda: self.rng.choice(np.asarray(SHAPE_COLOURS, dtype='object')))
        robot_pose = (0.058, 0.53, -2.13)
        if self.replay is not None and 'poses' in self.replay:
            (cx, cy, _), robot_pose = self.replay['poses']
            gx, gy = cx - gw / 2, cy + gh / 2             # GoalRegion(x, y, h, w): top-left corner of the box around the body
        self.goal = w.add(GoalRegion(gx, gy, gh, gw, goal_colour))
        self.robot = w.add(_robot(robot_pose[:2], robot_pose[2]))
        self.slots = [self.goal, self.robot]
        if (minor or full) and self.replay is None:       # move_to_region.py:63-78
            lim = dict(rel_pos_linf_limits=JITTER_POS_BOUND, rel_rot_limits=[None, JITTER_ROT_BOUND]) if minor else {}
            self.jitter([self.goal, self.robot], rand_rot=(False, True), **lim)

    def score_on_end_of_traj(self):
        # goal_shape.point_query(robot_pos) -> cpPolyShapePointQuery: dist <= 0
        # iff the point is not strictly outside any edge plane (or lies on the
        # boundary).  Box is axis-aligned, radius 0.
        x, y, _ = self.world_pose(self.robot.robot_body)
        l, b, r, t = self.goal.bb
        outside = (x - r > 0.0) or (y - t > 0.0) or (l - x > 0.0) or (b - y > 0.0)
        return 0.0 if outside else 1.0


class MatchRegionsRef(TaskRef):
    """benchmarks/match_regions.py:44-213."""
    name, ep_len = 'MatchRegions', 120

    def on_reset(self):
        w = self.world
        robot = _robot((-0.5, 0.1), -math.pi * 1.2)
        target_colour = 'green'
        if self.flags.get('rand_target_colour'):          # match_regions.py:51-58
            target_colour = self.draw('colour', lambda: self.rng.choice(SHAPE_COLOURS))
        minor, full = self.flags.get('rand_layout_minor'), self.flags.get('rand_layout_full')
        gx, gy, gh, gw = 0.1, 0.7, 0.7, 0.6
        if minor or full:                                 # match_regions.py:63-72
            gh, gw = self.draw('goal_hw', lambda: randomise_hw(RAND_GOAL_MIN_SIZE, RAND_GOAL_MAX_SIZE, self.rng, current_hw=(gh, gw),
                                                               linf_bound=JITTER_TARGET_BOUND if minor else None))
        target_types = ['star', 'square']
        target_poses = [(0.8, -0.7, 2.37), (-0.68, 0.72, 1.28)]
        distractor_colours = [c for c in SHAPE_COLOURS if c != target_colour]
        distractor_types = [[], ['pentagon'], ['circle', 'pentagon']]
        distractor_poses = [[], [(-0.05, -0.2, -1.09)], [(-0.75, -0.55, 2.78), (0.3, -0.82, -1.15)]]
        target_count, distractor_counts = len(target_types), [len(lst) for lst in distractor_types]
        if self.flags.get('rand_shape_count'):           # match_regions.py:101-105
            target_count, distractor_counts = self.draw('counts', lambda: (
                self.rng.randint(1, 2 + 1), [self.rng.randint(0, 2 + 1) for _ in distractor_colours]))
        if self.flags.get('rand_shape_type'):            # match_regions.py:110-117
            types_np = np.asarray(RAND_SHAPE_TYPES, dtype='object')
            target_types, distractor_types = self.draw('shape_types', lambda: (
                [self.rng.choice(types_np) for _ in range(target_count)],
                [[self.rng.choice(types_np) for _ in range(n)] for n in distractor_counts]))
        if full:                                          # match_regions.py:122-126
            target_poses = [(0, 0, 0)] * target_count
            distractor_poses = [[(0, 0, 0)] * n for n in distractor_counts]
        if self.replay is not None and 'poses' in self.replay:
            (cx, cy, _), (rx, ry, ra), *sp = self.replay['poses']
            gx, gy = cx - gw / 2, cy + gh / 2
            robot = _robot((rx, ry), ra)
            target_poses, rest = sp[:target_count], sp[target_count:]
            distractor_poses = []
            for n in distractor_counts:
                distractor_poses.append(rest[:n]); rest = rest[n:]
        self.sensor = w.add(GoalRegion(gx, gy, gh, gw, target_colour))
        self.target_shapes = [_shape(t, target_colour, (x, y), a) for t, (x, y, a) in zip(target_types, target_poses)]
        self.distractor_shapes = []
        for col, types, poses in zip(distractor_colours, distractor_types, distractor_poses):
            for t, (x, y, a) in zip(types, poses):
                self.distractor_shapes.append(_shape(t, col, (x, y), a))
        for e in self.target_shapes + self.distractor_shapes:
            w.add(e)
        self.robot = w.add(robot)
        # product slots: region, 2 targets, 2 distractors per distractor colour, robot (the Demo list without counts:
        # region, the 2 targets, the 3 distractors, robot)
        if self.flags.get('rand_shape_count'):
            pad = lambda lst, n: list(lst) + [None] * (n - len(lst))
            self.slots, at = [self.sensor, *pad(self.target_shapes, 2)], 0
            for n in distractor_counts:
                self.slots += pad(self.distractor_shapes[at:at + n], 2); at += n
            self.slots.append(self.robot)
        else:
            self.slots = [self.sensor, *self.target_shapes, *self.distractor_shapes, self.robot]
        if (minor or full) and self.replay is None:       # match_regions.py:166-188: the region is never rotated
            all_ents = [self.sensor, self.robot, *self.target_shapes, *self.distractor_shapes]
            lim = dict(rel_pos_linf_limits=JITTER_POS_BOUND, rel_rot_limits=JITTER_ROT_BOUND) if minor else {}
            self.jitter(all_ents, rand_rot=[False] + [True] * (len(all_ents) - 1), **lim)

    def score_on_end_of_traj(self):
        ents = self.target_shapes + self.distractor_shapes
        overlap = self.sensor.get_overlapping_ents(ents)
        n_t = len([e for e in overlap if e in self.target_shapes])
        n_d = len([e for e in overlap if e in self.distractor_shapes])
        target_frac_done = n_t / len(self.target_shapes)
        if len(overlap) == 0:
            contamination_rate = 0
        else:
            contamination_rate = n_d / len(overlap)
        return target_frac_done * (1 - contamination_rate)


def longest_line(points, inlier_dist, max_separation):
    """benchmarks/make_line.py:31-71."""
    points = np.asarray(points)
    npts = len(points)
    best = min(1, npts)
    for i in range(npts - 1):
        for j in range(i + 1, npts):
            pi = points[i]
            offs = points - pi[None]
            pj_off = offs[j]
            pj_unit = pj_off / np.linalg.norm(pj_off)
            proj_lens = np.squeeze(offs @ pj_unit[:, None], axis=1)
            dists = np.linalg.norm(offs - proj_lens[:, None] * pj_unit, axis=1)
            inlier_inds, = np.nonzero(dists <= inlier_dist)
            if len(inlier_inds) <= best:
                continue
            inlier_proj_lens = proj_lens[inlier_inds]
            inlier_proj_lens.sort()
            seps = np.abs(np.diff(inlier_proj_lens))
            all_runs = it.groupby(seps <= max_separation)
            one_run_lens = [len(list(r)) for v, r in all_runs if v]
            max_run = max(one_run_lens, default=0) + 1
            if max_run > best:
                best = max_run
    return best


class MakeLineRef(TaskRef):
    """benchmarks/make_line.py:10-28,91-152."""
    name, ep_len = 'MakeLine', 180

    def on_reset(self):
        w = self.world
        robot = _robot((0.702, -0.255), 0.347)
        colours = ['blue', 'yellow', 'red', 'green']
        shapes = ['star', 'circle', 'star', 'pentagon']
        poses = [((0.790, -0.820), -0.721), ((-0.177, 0.383), -1.733),
                 ((-0.051, -0.128), 2.696), ((-0.292, -0.745), -0.159)]
        n_blocks = len(shapes)
        if self.flags.get('rand_count'):                  # make_line.py:100-102 (MIN_BLOCKS, MAX_BLOCKS = 3, 4, :12-13)
            n_blocks = self.draw('count', lambda: self.rng.randint(3, 4 + 1))
            poses = poses[:1] * n_blocks
        if self.flags.get('rand_colours'):                # make_line.py:105-107
            colours = self.draw('colours', lambda: self.rng.choice(SHAPE_COLOURS, size=n_blocks).tolist())
        if self.flags.get('rand_shapes'):                 # make_line.py:108-110
            shapes = self.draw('shape_types', lambda: self.rng.choice(RAND_SHAPE_TYPES, size=n_blocks).tolist())
        if self.replay is not None and 'poses' in self.replay:
            (rx, ry, ra), *bp = self.replay['poses']
            robot = _robot((rx, ry), ra)
            poses = [((x, y), a) for x, y, a in bp]
        self.blocks = [w.add(_shape(s, c, p, a)) for s, c, (p, a) in zip(shapes, colours, poses)]
        self.robot = w.add(robot)
        self.slots = [*self.blocks, *[None] * (4 - len(self.blocks)), self.robot]
        if (self.flags.get('rand_layout_minor') or self.flags.get('rand_layout_full')) and self.replay is None:   # make_line.py:124-139
            lim = dict(rel_pos_linf_limits=JITTER_POS_BOUND, rel_rot_limits=JITTER_ROT_BOUND) if self.flags.get('rand_layout_minor') else {}
            self.jitter([self.robot, *self.blocks], **lim)

    def score_on_end_of_traj(self):
        points = np.asarray([self.block_pos(b) for b in self.blocks], dtype='float64')
        line_len = longest_line(points, SHAPE_RAD * 1.5, SHAPE_RAD * 3.5)
        max_line_len = len(points)
        min_line_len = max(max_line_len - 2, 2)
        return max(line_len - min_line_len, 0) / (max_line_len - min_line_len)


class FindDupeRef(TaskRef):
    """benchmarks/find_dupe.py:7-38,72-216."""
    name, ep_len = 'FindDupe', 100

    def on_reset(self):
        w = self.world
        robot = _robot((-0.57, 0.25), 3.83)
        out_shapes = ['pentagon', 'circle', 'circle', 'square', 'star', 'pentagon']
        out_colours = ['green', 'red', 'red', 'yellow', 'blue', 'yellow']
        query_colour, query_shape = 'yellow', 'pentagon'
        n_out_blocks = len(out_colours)
        if self.flags.get('rand_count'):                  # find_dupe.py:84-87
            n_out_blocks = self.draw('count', lambda: self.rng.randint(1, 5 + 1) + 1)
        n_distractors = n_out_blocks - 1
        if self.flags.get('rand_colours'):                # find_dupe.py:90-95
            def draw_colours():
                q = self.rng.choice(SHAPE_COLOURS)
                return q, self.rng.choice(SHAPE_COLOURS, size=n_distractors).tolist() + [q]
            query_colour, out_colours = self.draw('colours', draw_colours)
        if self.flags.get('rand_shapes'):                 # find_dupe.py:96-100
            def draw_shapes():
                q = self.rng.choice(RAND_SHAPE_TYPES)
                return q, self.rng.choice(RAND_SHAPE_TYPES, size=n_distractors).tolist() + [q]
            query_shape, out_shapes = self.draw('shape_types', draw_shapes)
        out_poses = [((-0.066751, 0.7552), -2.9266), ((-0.05195, 0.31468), 1.5418),
                     ((0.57528, -0.46865), -2.2141), ((0.40594, -0.74977), 0.24582),
                     ((0.45254, 0.3681), -1.0834), ((0.76849, -0.10652), 0.10028)]
        minor, full = self.flags.get('rand_layout_minor'), self.flags.get('rand_layout_full')
        gx, gy, gh, gw = -0.72, -0.22, 0.67, 0.72
        if minor or full:                                 # find_dupe.py:101-112
            gh, gw = self.draw('goal_hw', lambda: randomise_hw(RAND_GOAL_MIN_SIZE, RAND_GOAL_MAX_SIZE, self.rng, current_hw=(gh, gw),
                                                               linf_bound=JITTER_TARGET_BOUND if minor else None))
        query_pose = ((-0.33, -0.49), -0.51)
        if self.flags.get('rand_count'):                  # find_dupe.py:122-124
            out_poses = [((0, 0), 0)] * n_out_blocks
        if self.replay is not None and 'poses' in self.replay:
            P = self.replay['poses']
            (cx, cy, _) = P['sensor']
            gx, gy = cx - gw / 2, cy + gh / 2
            robot = _robot(P['robot'][:2], P['robot'][2])
            out_poses = [((x, y), a) for x, y, a in P['outside']]
            query_pose = (P['query'][:2], P['query'][2])
        self.sensor = w.add(GoalRegion(gx, gy, gh, gw, query_colour))
        self.outside_blocks, self.target_set = [], []
        for s, c, (p, a) in zip(out_shapes, out_colours, out_poses):
            blk = w.add(_shape(s, c, p, a))
            self.outside_blocks.append(blk)
            if c == query_colour and s == query_shape:
                self.target_set.append(blk)
        self.query_block = w.add(_shape(query_shape, query_colour, *query_pose))
        self.target_set.append(self.query_block)
        self.distractor_set = [b for b in self.outside_blocks if b not in self.target_set]
        self.robot = w.add(robot)
        ob = self.outside_blocks
        self.slots = [self.sensor, *ob, *[None] * (6 - len(ob)), self.query_block, self.robot]
        if (minor or full) and self.replay is None:       # find_dupe.py:157-196
            from . import placement_ref as pr
            all_ents = [self.sensor, self.robot, *self.outside_blocks]
            lim = dict(rel_pos_linf_limits=JITTER_POS_BOUND, rel_rot_limits=JITTER_ROT_BOUND) if minor else {}
            pr.randomise_all_poses(w, all_ents, [-1, 1, -1, 1], self.rng, rand_rot=[False] + [True] * (len(all_ents) - 1),
                                   ignore_shapes=self.query_block.shapes, **lim)
            # the query block last: onto the (moved) region, then jittered so that it stays mostly inside it
            query_pos_limit = max(0, min(gh, gw) / 2 - SHAPE_RAD / 2)
            if minor:
                query_pos_limit = min(JITTER_POS_BOUND, query_pos_limit)
            pr.shift_bodies(w, self.query_block.bodies, self.main_pose(self.sensor)[:2], self.main_pose(self.query_block)[2])
            pr.randomise_pose(w, self.query_block, [-1, 1, -1, 1], self.rng, rel_pos_linf_limit=query_pos_limit,
                              rel_rot_limit=JITTER_ROT_BOUND if minor else None, ignore=set(self.sensor.shapes))
            self.choices['poses'] = {'sensor': self.main_pose(self.sensor), 'robot': self.main_pose(self.robot),
                                     'outside': [self.main_pose(b) for b in self.outside_blocks], 'query': self.main_pose(self.query_block)}

    def score_on_end_of_traj(self):
        overlap = self.sensor.get_overlapping_ents([self.query_block, *self.outside_blocks])
        n_t = len([e for e in overlap if e in self.target_set])
        n_d = len([e for e in overlap if e in self.distractor_set])
        have_two_shapes = float(n_t >= 2)
        if len(overlap) == 0:
            contamination_rate = 0
        else:
            contamination_rate = n_d / len(overlap)
        return have_two_shapes * (1 - contamination_rate)


class FixColourRef(TaskRef):
    """benchmarks/fix_colour.py:15-40,69-202."""
    name, ep_len = 'FixColour', 60

    def on_reset(self):
        w = self.world
        robot = _robot((0.368, 0.586), 0.718)
        block_colours = ['green', 'green', 'blue']
        block_shapes = ['pentagon', 'square', 'pentagon']
        block_poses = [((0.289, 0.030), 0.307), ((0.133, -0.561), 1.699), ((-0.336, 0.000), -1.529)]
        region_xyhws = [(-0.032, 0.348, 0.427, 0.468), (0.019, -0.391, 0.460, 0.458),
                        (-0.681, 0.196, 0.498, 0.418)]
        region_colours = ['green', 'green', 'red']
        n_regions = len(block_colours)
        if self.flags.get('rand_count'):                  # fix_colour.py:78-82 (MIN_REGIONS, MAX_REGIONS = 2, 3, :9-10)
            n_regions = self.draw('count', lambda: self.rng.randint(2, 3 + 1))
            block_poses = block_poses[:1] * n_regions
            region_xyhws = region_xyhws[:1] * n_regions
        if self.flags.get('rand_colours'):                # fix_colour.py:84-94
            def draw_colours():
                rc = self.rng.choice(SHAPE_COLOURS, size=n_regions).tolist()
                bc = list(rc)
                odd_idx = self.rng.randint(len(bc))
                new_col_idx = self.rng.randint(len(SHAPE_COLOURS) - 1)
                if SHAPE_COLOURS[new_col_idx] == bc[odd_idx]:
                    new_col_idx += 1
                bc[odd_idx] = SHAPE_COLOURS[new_col_idx]
                return rc, bc
            region_colours, block_colours = self.draw('colours', draw_colours)
        if self.flags.get('rand_shapes'):                 # fix_colour.py:97-99
            block_shapes = self.draw('shape_types', lambda: self.rng.choice(RAND_SHAPE_TYPES, size=n_regions).tolist())
        minor, full = self.flags.get('rand_layout_minor'), self.flags.get('rand_layout_full')
        if minor or full:                                 # fix_colour.py:102-113 (MIN / MAX_GOAL_SIZE = 0.4 / 0.5, :13-14)
            hws = self.draw('goal_hw', lambda: [randomise_hw(0.4, 0.5, self.rng, current_hw=hw, linf_bound=JITTER_TARGET_BOUND if minor else None)
                                                for _, _, *hw in region_xyhws])
            region_xyhws = [(x, y, h, w_) for (x, y, _, _), (h, w_) in zip(region_xyhws, hws)]
        if self.replay is not None and 'poses' in self.replay:
            P = self.replay['poses']
            region_xyhws = [(cx - w_ / 2, cy + h / 2, h, w_) for (cx, cy, _), (_, _, h, w_) in zip(P['sensors'], region_xyhws)]
            robot = _robot(P['robot'][:2], P['robot'][2])
            block_poses = [((x, y), a) for x, y, a in P['blocks']]
        self.sensors = [w.add(GoalRegion(*xyhw, col)) for col, xyhw in zip(region_colours, region_xyhws)]
        self.blocks, self.target_blocks = [], []
        for s, c, tc, (p, a) in zip(block_shapes, block_colours, region_colours, block_poses):
            blk = _shape(s, c, p, a)
            self.blocks.append(blk)
            self.target_blocks.append([] if c != tc else [blk])
        for b in self.blocks:
            w.add(b)
        self.robot = w.add(robot)
        pad = lambda lst: list(lst) + [None] * (3 - len(lst))
        self.slots = [*pad(self.sensors), *pad(self.blocks), self.robot]
        if (minor or full) and self.replay is None:       # fix_colour.py:143-187
            from . import placement_ref as pr
            n = len(self.sensors)
            lim = dict(rel_pos_linf_limits=JITTER_POS_BOUND, rel_rot_limits=JITTER_ROT_BOUND) if minor else {}
            block_shapes_all = [sh for b in self.blocks for sh in b.shapes]
            pr.randomise_all_poses(w, [*self.sensors, self.robot], [-1, 1, -1, 1], self.rng, rand_rot=[False] * n + [True],
                                   ignore_shapes=block_shapes_all, **lim)
            for block, sensor in zip(self.blocks, self.sensors):      # blocks onto their regions ...
                pr.shift_bodies(w, block.bodies, self.main_pose(sensor)[:2], self.main_pose(block)[2])
            for block, sensor, (_, _, sh, sw) in zip(self.blocks, self.sensors, region_xyhws):   # ... then jittered inside them
                block_pos_limit = max(0, min(sh, sw) / 2 - SHAPE_RAD)
                if minor:
                    block_pos_limit = min(JITTER_POS_BOUND, block_pos_limit)
                pr.randomise_pose(w, block, [-1, 1, -1, 1], self.rng, rel_pos_linf_limit=block_pos_limit,
                                  rel_rot_limit=JITTER_ROT_BOUND if minor else None, ignore=set(sensor.shapes))
            self.choices['poses'] = {'sensors': [self.main_pose(s_) for s_ in self.sensors], 'robot': self.main_pose(self.robot),
                                     'blocks': [self.main_pose(b) for b in self.blocks]}

    def score_on_end_of_traj(self):
        for sensor, tgt in zip(self.sensors, self.target_blocks):
            overlap = sensor.get_overlapping_ents(self.blocks)
            # list(set) == [..]: equal only for 0 or 1 elements
            if len(overlap) != len(tgt) or any(a is not b for a, b in zip(overlap, tgt)):
                return 0.0
        return 1.0


class _ClusterRef(TaskRef):
    """benchmarks/cluster.py:67-216."""
    ep_len = 240
    by = None

    def on_reset(self):
        w = self.world
        robot = _robot(*self.ROBOT_POSE)
        colours, shape_types, poses = self.COLOURS, self.SHAPES, self.POSES
        n_shapes = len(colours)
        if self.flags.get('rand_shape_count'):           # cluster.py:81-85
            n_shapes = self.draw('count', lambda: self.rng.randint(7, 10 + 1))
            poses = [((0, 0), 0)] * n_shapes
        if self.flags.get('rand_shape_colour'):          # cluster.py:91-100
            def draw_colours():
                cs = list(SHAPE_COLOURS)
                cs.extend([self.rng.choice(SHAPE_COLOURS) for _ in range(n_shapes - len(cs))])
                self.rng.shuffle(cs)
                return cs
            colours = self.draw('colours', draw_colours)
        if self.flags.get('rand_shape_type'):            # cluster.py:102-110
            def draw_types():
                ts = list(RAND_SHAPE_TYPES)
                ts.extend([self.rng.choice(RAND_SHAPE_TYPES) for _ in range(n_shapes - len(ts))])
                self.rng.shuffle(ts)
                return ts
            shape_types = self.draw('shape_types', draw_types)
        if self.replay is not None and 'poses' in self.replay:
            (rx, ry, ra), *bp = self.replay['poses']
            robot = _robot((rx, ry), ra)
            poses = [((x, y), a) for x, y, a in bp]
        self.shape_ents = [w.add(_shape(s, c, p, a))
                           for (p, a), c, s in zip(poses, colours, shape_types)]
        c_values_list = np.asarray(colours if self.by == 'colour' else shape_types, dtype='object')
        self.characteristic_values = np.unique(c_values_list)
        self.blocks_by_characteristic = {}
        for shape, c_value in zip(self.shape_ents, c_values_list):
            self.blocks_by_characteristic.setdefault(c_value, []).append(shape)
        self.robot = w.add(robot)
        n_slots = 10 if self.flags.get('rand_shape_count') else len(self.shape_ents)
        self.slots = [*self.shape_ents, *[None] * (n_slots - len(self.shape_ents)), self.robot]
        if (self.flags.get('rand_layout_minor') or self.flags.get('rand_layout_full')) and self.replay is None:   # cluster.py:148-161
            lim = dict(rel_pos_linf_limits=JITTER_POS_BOUND, rel_rot_limits=JITTER_ROT_BOUND) if self.flags.get('rand_layout_minor') else {}
            self.jitter([self.robot, *self.shape_ents], **lim)

    def score_on_end_of_traj(self):
        nvals = len(self.characteristic_values)
        centroids = np.zeros((nvals, 2))
        for c_idx, c_value in enumerate(self.characteristic_values):
            c_blocks = self.blocks_by_characteristic.get(c_value)
            if not c_blocks:
                centroid = (0, 0)
            else:
                positions = np.asarray([self.block_pos(b) for b in c_blocks])
                centroid = np.mean(positions, axis=0)
            centroids[c_idx] = centroid
        min_margin = 2.0
        n_blocks = 0
        n_correct = 0
        for c_idx, c_value in enumerate(self.characteristic_values):
            for block in self.blocks_by_characteristic.get(c_value, []):
                n_blocks += 1
                block_pos = np.array([list(self.block_pos(block))])
                centroid_sses = np.sum((block_pos - centroids)**2, axis=1)
                indices = np.arange(nvals)
                true_sse, = centroid_sses[indices == c_idx]
                bad_sses = centroid_sses[indices != c_idx]
                nearest_bad_centroid = np.min(bad_sses)
                true_centroid_sse = centroid_sses[c_idx]
                margin = min_margin * true_centroid_sse     # squared distance: reference quirk
                n_correct += int(np.sqrt(true_sse) < np.sqrt(nearest_bad_centroid) - margin)
        frac_correct = float(n_correct) / max(n_blocks, 1)
        thresh = 0.75
        return max(frac_correct - thresh, 0) / (1 - thresh)


class ClusterColourRef(_ClusterRef):
    """benchmarks/cluster.py:219-256."""
    name, by = 'ClusterColour', 'colour'
    ROBOT_POSE = ((0.71692, -0.34374), 0.83693)
    COLOURS = ['blue', 'blue', 'blue', 'green', 'green', 'red', 'yellow', 'yellow']
    SHAPES = ['circle', 'star', 'square', 'pentagon', 'pentagon', 'square', 'star', 'pentagon']
    POSES = [((-0.5147, 0.14149), -0.38871), ((-0.1347, -0.71414), 1.0533),
             ((-0.74247, -0.097592), 1.1571), ((-0.077363, -0.42964), -0.64379),
             ((0.51978, 0.1853), -1.1762), ((-0.5278, -0.21642), 2.9356),
             ((-0.54039, 0.48292), 0.072818), ((-0.16761, 0.64303), -2.3255)]


class ClusterShapeRef(_ClusterRef):
    """benchmarks/cluster.py:259-297."""
    name, by = 'ClusterShape', 'type'
    ROBOT_POSE = ((0.286, -0.202), -1.878)
    COLOURS = ['yellow', 'blue', 'red', 'red', 'green', 'yellow', 'blue', 'green']
    SHAPES = ['square', 'pentagon', 'pentagon', 'pentagon', 'circle', 'star', 'star', 'circle']
    POSES = [((-0.414, 0.297), -1.731), ((0.068, 0.705), 2.184), ((0.821, 0.220), 0.650),
             ((-0.461, -0.749), -2.673), ((0.867, -0.149), -2.215), ((-0.785, -0.140), -0.405),
             ((-0.305, -0.226), 1.341), ((0.758, -0.708), -2.140)]


TASKS = {c.name: c for c in (MoveToCornerRef, MoveToRegionRef, MatchRegionsRef, MakeLineRef,
                             FindDupeRef, FixColourRef, ClusterColourRef, ClusterShapeRef)}
