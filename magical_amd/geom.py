"""Pose randomisation of the Test*Jitter / TestLayout variants: the host-side mirror of the reference's
magical/geom.py:116-341 (pm_randomise_pose, pm_randomise_all_poses).

The reference moves pymunk bodies and asks the space whether the moved shapes touch anything; here the entity poses
of ONE env are a float64[n_entities, 3] array, and the same question goes to the world builder
(mgx_world_placement_collides: arena walls + every enabled entity, ShapeFilter groups honoured).  The random draws
are the reference's, call for call: per attempt rng.uniform for x, for y and for the angle, entity after entity, each
entity colliding only with the ones placed before it and with everything that is not being randomised.
"""
import ctypes as C

import numpy as np

from . import _native as nat


class PlacementError(Exception):
    """geom.py:110-113."""


def placement_collides(env, ent_id, poses, enabled):
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    enabled = np.ascontiguousarray(enabled, dtype=np.uint8)
    return bool(nat.check(env._lib.mgx_world_placement_collides(
        env._world, int(ent_id), poses.ctypes.data_as(C.POINTER(C.c_double)), enabled.ctypes.data_as(C.POINTER(C.c_uint8)))))


def pm_randomise_pose(env, poses, ent_id, enabled, arena_lrbt, rng, rand_pos=True, rand_rot=True, rel_pos_linf_limit=None,
                      rel_rot_limit=None, rejection_tests=()):
    """geom.py:116-262.  `poses[ent_id]` is updated in place; returns the number of rejected attempts."""
    assert rand_pos or rand_rot, 'need to randomise at least one thing, or placement may be impossible'
    orig = poses[ent_id].copy()
    arena_l, arena_r, arena_b, arena_t = arena_lrbt
    if rel_pos_linf_limit is not None:
        assert 0 <= rel_pos_linf_limit
        pos_x_minmax = (max(arena_l, orig[0] - rel_pos_linf_limit), min(arena_r, orig[0] + rel_pos_linf_limit))
        pos_y_minmax = (max(arena_b, orig[1] - rel_pos_linf_limit), min(arena_t, orig[1] + rel_pos_linf_limit))
    else:
        pos_x_minmax, pos_y_minmax = (arena_l, arena_r), (arena_b, arena_t)
    if rel_rot_limit is not None:
        assert 0 <= rel_rot_limit
        rot_min, rot_max = orig[2] - rel_rot_limit, orig[2] + rel_rot_limit
    else:
        rot_min, rot_max = -np.pi, np.pi
    max_tries, n_tries = 10000, 0
    while n_tries < max_tries:
        if rand_pos:
            poses[ent_id, 0] = rng.uniform(*pos_x_minmax)
            poses[ent_id, 1] = rng.uniform(*pos_y_minmax)
        if rand_rot:
            poses[ent_id, 2] = rng.uniform(rot_min, rot_max)
        reject = placement_collides(env, ent_id, poses, enabled)
        if not reject:
            for rejection_test in rejection_tests:
                reject = reject or rejection_test(poses)
                if reject:
                    break
        if not reject:
            break
        n_tries += 1
    else:
        poses[ent_id] = orig
        raise PlacementError(f'could not place entity {ent_id} after {n_tries} attempts')
    return n_tries


def pm_randomise_all_poses(env, poses, entities, arena_lrbt, rng, rand_pos=True, rand_rot=True, rel_pos_linf_limits=None,
                           rel_rot_limits=None, ignore=(), max_retries=10, rejection_tests=()):
    """geom.py:285-341.  entities: the Entity objects to randomise, in order; `ignore`: entities whose shapes never
    count (the reference's ignore_shapes).  `poses` (float64[n_entities, 3], indexed by ent_id) is updated in place."""
    n = len(entities)
    lst = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * n
    pos_limits, rot_limits, rand_pos, rand_rot = lst(rel_pos_linf_limits), lst(rel_rot_limits), lst(rand_pos), lst(rand_rot)
    for retry in range(max_retries):
        enabled = np.ones(len(poses), dtype=np.uint8)
        for e in list(entities) + list(ignore):
            enabled[e.ent_id] = 0                      # categories = 0 until its turn / ignored altogether
        for ent, pl, rl, rp, rr in zip(entities, pos_limits, rot_limits, rand_pos, rand_rot):
            enabled[ent.ent_id] = 1
            try:
                pm_randomise_pose(env, poses, ent.ent_id, enabled, arena_lrbt, rng, rand_pos=rp, rand_rot=rr,
                                  rel_pos_linf_limit=pl, rel_rot_limit=rl, rejection_tests=rejection_tests)
            except PlacementError:
                if retry == max_retries - 1:
                    raise
                break
        else:
            break
    return poses
