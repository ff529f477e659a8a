"""Restatement of the reference's pose randomisation (geom.py:116-341,362-384) over the C oracle's shapes.

TEST INFRASTRUCTURE.  `randomise_all_poses` draws from the env's RandomState exactly as pm_randomise_all_poses does
(two uniforms for the position and one for the angle per attempt, entity after entity, each colliding only with the
arena, the entities placed before it and everything that is not being randomised) and tests a candidate pose the way
pm_randomise_pose does: space.shape_query of every shape of the entity = cpCollide(...).count > 0 against every
other shape the ShapeFilters let it see (ref_collide_shapes, the oracle's GJK/EPA narrowphase).

One deliberate difference, shared with the product: the reference moves the robot's finger bodies with
pm_shift_bodies, which leaves the zero-length finger PinJoints with a rounding-level (1e-17) separation whose
direction then drives the first impulses; here the drawn poses are RETURNED and the episode's
world is built at them, so the fingers sit exactly on their anchors as they do in the Demo variant.
"""
import ctypes as C
import math

import numpy as np


class PlacementError(Exception):
    pass


def _bodies(world):
    n = world.L.ref_nbodies(world.h)
    buf = np.zeros((n, 9), dtype=np.float64)
    world.L.ref_get_bodies(world.h, buf.ctypes.data_as(C.POINTER(C.c_double)))
    return buf


def _set_bodies(world, buf):
    world.L.ref_set_bodies(world.h, np.ascontiguousarray(buf).ctypes.data_as(C.POINTER(C.c_double)))


def shift_bodies(world, bodies, position, angle):
    """geom.py:362-384 pm_shift_bodies: rigid transform of `bodies` that puts the first one at (position, angle)."""
    buf = _bodies(world)
    root_angle, root_pos = buf[bodies[0], 2], buf[bodies[0], :2].copy()
    d = angle - root_angle
    c, s = math.cos(d), math.sin(d)
    for b in bodies:
        local_angle_delta = buf[b, 2] - root_angle
        lx, ly = buf[b, 0] - root_pos[0], buf[b, 1] - root_pos[1]
        buf[b, 2] = angle + local_angle_delta
        buf[b, 0] = position[0] + (lx * c - ly * s)           # Vec2d.rotated
        buf[b, 1] = position[1] + (lx * s + ly * c)
    _set_bodies(world, buf)


def _shape_info(world, s):
    out = (C.c_int * 3)()
    world.L.ref_shape_info(world.h, s, out)
    return out[0], out[1], out[2]


def collides(world, shapes, disabled, ignore):
    """any shape of `shapes` touching a shape outside `shapes`, `disabled` (categories = 0) and `ignore`."""
    out = (C.c_double * 19)()
    n = world.L.ref_nshapes(world.h)
    own = set(shapes)
    for s in shapes:
        _, gs, _ = _shape_info(world, s)
        for t in range(n):
            if t in own or t in disabled or t in ignore:
                continue
            bt, gt_, _ = _shape_info(world, t)
            if gs != 0 and gs == gt_:
                continue                                          # cpShapeFilterReject: same non-zero group
            if world.L.ref_collide_shapes(world.h, s, t, out) > 0:
                return True
    return False


def randomise_pose(world, ent, arena_lrbt, rng, rand_pos=True, rand_rot=True, rel_pos_linf_limit=None, rel_rot_limit=None,
                   disabled=(), ignore=()):
    """geom.py:116-262 for the bodies / shapes of `ent`.  Returns the number of rejected attempts."""
    assert rand_pos or rand_rot
    bodies, main = ent.bodies, ent.bodies[0]
    saved = _bodies(world)
    orig_angle, orig_pos = float(saved[main, 2]), (float(saved[main, 0]), float(saved[main, 1]))
    arena_l, arena_r, arena_b, arena_t = arena_lrbt
    if rel_pos_linf_limit is not None:
        pos_x_minmax = (max(arena_l, orig_pos[0] - rel_pos_linf_limit), min(arena_r, orig_pos[0] + rel_pos_linf_limit))
        pos_y_minmax = (max(arena_b, orig_pos[1] - rel_pos_linf_limit), min(arena_t, orig_pos[1] + rel_pos_linf_limit))
    else:
        pos_x_minmax, pos_y_minmax = (arena_l, arena_r), (arena_b, arena_t)
    if rel_rot_limit is not None:
        rot_min, rot_max = orig_angle - rel_rot_limit, orig_angle + rel_rot_limit
    else:
        rot_min, rot_max = -np.pi, np.pi
    max_tries, n_tries = 10000, 0
    while n_tries < max_tries:
        new_pos = (rng.uniform(*pos_x_minmax), rng.uniform(*pos_y_minmax)) if rand_pos else orig_pos
        new_angle = rng.uniform(rot_min, rot_max) if rand_rot else orig_angle
        shift_bodies(world, bodies, new_pos, new_angle)
        if not collides(world, ent.shapes, disabled, ignore):
            break
        n_tries += 1
    else:
        _set_bodies(world, saved)
        raise PlacementError(f'could not place {ent}')
    return n_tries


def randomise_all_poses(world, entities, arena_lrbt, rng, rand_pos=True, rand_rot=True, rel_pos_linf_limits=None,
                        rel_rot_limits=None, ignore_shapes=None, max_retries=10):
    """geom.py:285-341.  Returns [(x, y, angle)] of every entity's main body, in the order of `entities`."""
    n = len(entities)
    lst = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * n
    pos_limits, rot_limits, rand_pos, rand_rot = lst(rel_pos_linf_limits), lst(rel_rot_limits), lst(rand_pos), lst(rand_rot)
    ignore = set(ignore_shapes or ())
    for retry in range(max_retries):
        disabled = set(s for e in entities for s in e.shapes)     # categories = 0: collide with nothing
        for ent, pl, rl, rp, rr in zip(entities, pos_limits, rot_limits, rand_pos, rand_rot):
            disabled -= set(ent.shapes)
            try:
                randomise_pose(world, ent, arena_lrbt, rng, rand_pos=rp, rand_rot=rr, rel_pos_linf_limit=pl, rel_rot_limit=rl,
                               disabled=disabled, ignore=ignore)
            except PlacementError:
                if retry == max_retries - 1:
                    raise
                break
        else:
            break
    buf = _bodies(world)
    return [(float(buf[e.bodies[0], 0]), float(buf[e.bodies[0], 1]), float(buf[e.bodies[0], 2])) for e in entities]
