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


def placement_collides(env, ent_id, poses, enabled, ent_hw=None):
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    enabled = np.ascontiguousarray(enabled, dtype=np.uint8)
    hw = None if ent_hw is None else np.ascontiguousarray(ent_hw, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))
    return bool(nat.check(env._lib.mgx_world_placement_collides(
        env._world, int(ent_id), poses.ctypes.data_as(C.POINTER(C.c_double)), enabled.ctypes.data_as(C.POINTER(C.c_uint8)), hw)))


def pm_randomise_pose(env, poses, ent_id, enabled, arena_lrbt, rng, rand_pos=True, rand_rot=True, rel_pos_linf_limit=None,
                      rel_rot_limit=None, rejection_tests=(), ent_hw=None):
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
        reject = placement_collides(env, ent_id, poses, enabled, ent_hw)
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
                           rel_rot_limits=None, ignore=(), max_retries=10, rejection_tests=(), native=True, ent_hw=None):
    """geom.py:285-341.  entities: the Entity objects to randomise, in order; `ignore`: entities whose shapes never
    count (the reference's ignore_shapes).  `poses` (float64[n_entities, 3], indexed by ent_id) is updated in place.

    Without custom rejection tests the whole procedure runs natively (mgx_world_randomise_all_poses) on the
    RandomState's own MT19937 stream, which it advances exactly as the Python loop below would."""
    n = len(entities)
    lst = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * n
    pos_limits, rot_limits, rand_pos, rand_rot = lst(rel_pos_linf_limits), lst(rel_rot_limits), lst(rand_pos), lst(rand_rot)
    if native and not rejection_tests and max_retries == 10:
        kind, key, pos, has_gauss, cached = rng.get_state()
        assert kind == 'MT19937'
        key = np.ascontiguousarray(key, dtype=np.uint32).copy()
        cpos = C.c_int(int(pos))
        ents = (C.c_int * n)(*[e.ent_id for e in entities])
        ign = np.zeros(len(poses), dtype=np.uint8)
        for e in ignore:
            ign[e.ent_id] = 1
        u8 = lambda v: np.asarray([1 if x else 0 for x in v], dtype=np.uint8)
        lim = lambda v: np.asarray([-1.0 if x is None else float(x) for x in v], dtype=np.float64)
        rp, rr, pl, rl = u8(rand_pos), u8(rand_rot), lim(pos_limits), lim(rot_limits)
        arena = np.asarray(arena_lrbt, dtype=np.float64)
        assert poses.dtype == np.float64 and poses.flags.c_contiguous
        P8, PD = C.POINTER(C.c_uint8), C.POINTER(C.c_double)
        rc = env._lib.mgx_world_randomise_all_poses(
            env._world, poses.ctypes.data_as(PD), ents, n, ign.ctypes.data_as(P8), arena.ctypes.data_as(PD), rp.ctypes.data_as(P8),
            rr.ctypes.data_as(P8), pl.ctypes.data_as(PD), rl.ctypes.data_as(PD), key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(cpos),
            None if ent_hw is None else np.ascontiguousarray(ent_hw, dtype=np.float64).ctypes.data_as(PD))
        rng.set_state((kind, key, cpos.value, has_gauss, cached))
        if rc < 0:
            raise PlacementError(env._lib.mgx_last_error().decode())
        return poses
    for retry in range(max_retries):
        enabled = np.ones(len(poses), dtype=np.uint8)
        for e in list(entities) + list(ignore):
            enabled[e.ent_id] = 0                      # categories = 0 until its turn / ignored altogether
        for ent, pl, rl, rp, rr in zip(entities, pos_limits, rot_limits, rand_pos, rand_rot):
            enabled[ent.ent_id] = 1
            try:
                pm_randomise_pose(env, poses, ent.ent_id, enabled, arena_lrbt, rng, rand_pos=rp, rand_rot=rr,
                                  rel_pos_linf_limit=pl, rel_rot_limit=rl, rejection_tests=rejection_tests, ent_hw=ent_hw)
            except PlacementError:
                if retry == max_retries - 1:
                    raise
                break
        else:
            break
    return poses


def pm_randomise_all_poses_batch(env, poses, entities, arena_lrbt, rngs, rand_pos=True, rand_rot=True, rel_pos_linf_limits=None,
                                 rel_rot_limits=None, ignore=(), ent_hw=None, env_idx=None, addrs=None):
    """pm_randomise_all_poses for M envs in one native call: poses float64[M, n_entities, 3] (updated in place), rngs the
    M envs' np.random.RandomState objects, whose MT19937 states are advanced in place through their ctypes address.
    The limits are scalars / per-entity lists as in the reference, or float64[M, n] arrays (NaN = no limit) when they
    differ between envs (a block jittered inside its own env's goal region)."""
    m, n = len(rngs), len(entities)
    lst = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * n
    rand_pos, rand_rot = lst(rand_pos), lst(rand_rot)
    per_env = isinstance(rel_pos_linf_limits, np.ndarray) or isinstance(rel_rot_limits, np.ndarray)

    def lim(v):
        if isinstance(v, np.ndarray):
            a = np.asarray(v, dtype=np.float64).reshape(m, n)
        else:
            a = np.asarray([np.nan if x is None else float(x) for x in lst(v)], dtype=np.float64)
            a = np.tile(a, (m, 1)) if per_env else a
        return np.ascontiguousarray(np.where(np.isnan(a), -1.0, a))
    pl, rl = lim(rel_pos_linf_limits), lim(rel_rot_limits)
    if addrs is None:        # (uint64[M] from batch_rng.state_addresses(rngs), if the caller has them already)
        from .batch_rng import state_addresses
        addrs = state_addresses(rngs)
    assert len(addrs) == m
    ents = (C.c_int * n)(*[e.ent_id for e in entities])
    ign = np.zeros(poses.shape[1], dtype=np.uint8)
    for e in ignore:
        ign[e.ent_id] = 1
    u8 = lambda v: np.asarray([1 if x else 0 for x in v], dtype=np.uint8)
    rp, rr = u8(rand_pos), u8(rand_rot)
    arena = np.asarray(arena_lrbt, dtype=np.float64)
    assert poses.dtype == np.float64 and poses.flags.c_contiguous and poses.shape[0] == m
    P8, PD = C.POINTER(C.c_uint8), C.POINTER(C.c_double)
    hw = None if ent_hw is None else np.ascontiguousarray(ent_hw, dtype=np.float64).ctypes.data_as(PD)
    if getattr(env, 'variable_worlds', False):
        # every env is placed in its own episode's world (shape types / entity counts differ between envs)
        assert env_idx is not None and len(env_idx) == m
        idx32 = np.ascontiguousarray(env_idx, dtype=np.int32)
        rc = env._lib.mgx_engine_env_randomise_all_poses_batch(
            env._engine, m, idx32.ctypes.data_as(C.POINTER(C.c_int)), poses.ctypes.data_as(PD), ents, n, ign.ctypes.data_as(P8),
            arena.ctypes.data_as(PD), rp.ctypes.data_as(P8), rr.ctypes.data_as(P8), pl.ctypes.data_as(PD), rl.ctypes.data_as(PD),
            1 if per_env else 0, addrs.ctypes.data_as(C.POINTER(C.c_uint64)), hw)
    else:
        rc = env._lib.mgx_world_randomise_all_poses_batch(
            env._world, m, poses.ctypes.data_as(PD), ents, n, ign.ctypes.data_as(P8), arena.ctypes.data_as(PD), rp.ctypes.data_as(P8),
            rr.ctypes.data_as(P8), pl.ctypes.data_as(PD), rl.ctypes.data_as(PD), 1 if per_env else 0, addrs.ctypes.data_as(C.POINTER(C.c_uint64)), hw)
    if rc < 0:
        raise PlacementError(env._lib.mgx_last_error().decode())
    return poses


def randomise_hw(min_side, max_side, rng, current_hw=None, linf_bound=None):
    """geom.py:344-360: height and width of a goal region, two draws (h, then w).  The reference draws both with one
    rng.uniform(minima, maxima) on length-2 arrays, which numpy evaluates element by element as
    low + (high - low) * random_sample() -- the two scalar calls below (this runs once per env and reset)."""
    assert min_side <= max_side
    lo_h = lo_w = float(min_side)
    hi_h = hi_w = float(max_side)
    if linf_bound is not None:
        assert linf_bound == float(linf_bound) and current_hw is not None and len(current_hw) == 2
        lo_h, hi_h = max(lo_h, current_hw[0] - linf_bound), min(hi_h, current_hw[0] + linf_bound)
        lo_w, hi_w = max(lo_w, current_hw[1] - linf_bound), min(hi_w, current_hw[1] + linf_bound)
    uh, uw = rng.random_sample(2)
    h = lo_h + (hi_h - lo_h) * uh
    w = lo_w + (hi_w - lo_w) * uw
    return h, w
