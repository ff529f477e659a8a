"""Host-side float64 helpers shared by the tasks' score_on_end_of_traj() implementations.

Scores are functions of the final body poses only; they run once per episode on the poses
downloaded from the device (float64).  Everything is batched over the M finished envs, with
operation orders chosen so the result is bit-identical to the reference's per-env numpy code.
"""
import ctypes as C
import math

import numpy as np

from .. import _native as nat
from .. import entities as en


def _row_norm_loop(d):
    out = np.empty(len(d), dtype=np.float64)
    dot, sqrt = np.dot, math.sqrt
    for k, r in enumerate(d):
        out[k] = sqrt(dot(r, r))
    return out


_ROW_NORM_BATCHED = [
    # stacked (1 x 2) @ (2 x 1): numpy routes this through the same dot kernel on the builds seen so far
    lambda d: np.sqrt(np.matmul(d[:, None, :], d[:, :, None])[:, 0, 0]),
    # no FMA in the BLAS in use
    lambda d: np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]),
]
_row_norm_impl = None


def row_norm(d):
    """np.linalg.norm(v) for each row v of d[M, 2].  The reference calls the 1-D form, which numpy evaluates as
    sqrt(v.dot(v)) through BLAS ddot -- with or without FMA depending on the BLAS kernel the CPU selects, so elementwise
    x*x + y*y can differ in the last bit.  The row-by-row call is the definition (3 ms per 4096 rows); a batched
    expression is used instead once it has reproduced the row-by-row results bit for bit on 2048 probe rows in this
    process (a wrong candidate differs on ~8 % of rows), so the result is the reference's either way."""
    global _row_norm_impl
    d = np.ascontiguousarray(d, dtype=np.float64)
    if _row_norm_impl is None:
        rs = np.random.RandomState(20260928)
        probe = rs.uniform(-2.0, 2.0, size=(2048, 2))
        probe[:256] *= rs.uniform(1e-6, 1.0, size=(256, 1))
        want = _row_norm_loop(probe)
        _row_norm_impl = next((f for f in _ROW_NORM_BATCHED if np.array_equal(f(probe), want)), _row_norm_loop)
    return _row_norm_impl(d) if len(d) else np.empty(0, dtype=np.float64)


_NUMPY_DOT_MODES = None


def numpy_dot_modes():
    """(dot_mode, mm_mode) for mgx_engine_score_points, or None where this numpy does something else.  The reference's point scores go
    through two library kernels whose rounding depends on the build: np.dot of a 2-vector with itself (np.linalg.norm, BLAS ddot) and the
    [n, 2] @ [2, 1] product of make_line.py:47.  Each is compared, on 2048 probe rows, with the three ways a sum of two products can be
    rounded -- 0: round(round(x x') + round(y y')), 1: fma(y, y', round(x x')), 2: fma(x, x', round(y y')) -- evaluated EXACTLY (rational
    arithmetic, one rounding where an fma has one); the mode is the one that reproduces every row."""
    global _NUMPY_DOT_MODES
    if _NUMPY_DOT_MODES is None:
        from fractions import Fraction
        rs = np.random.RandomState(20260929)
        a = rs.uniform(-2.0, 2.0, size=(2048, 2)); b = rs.uniform(-2.0, 2.0, size=(2048, 2))
        a[:256] *= rs.uniform(1e-6, 1.0, size=(256, 1))

        def cands(ax, ay, bx, by):
            fx, fy = Fraction(ax) * Fraction(bx), Fraction(ay) * Fraction(by)
            return (ax * bx + ay * by, float(Fraction(ax * bx) + fy), float(fx + Fraction(ay * by)))
        dot_want = np.array([np.dot(r, r) for r in a])
        dot_c = np.array([cands(r[0], r[1], r[0], r[1]) for r in a])
        # the product as longest_line() forms it: offs[n, 2] @ unit[:, None] -> [n, 1], n = 3 or 4 blocks (matmul picks its kernel by
        # shape: a [1, 2] @ [2, 1] product goes another way than these on the builds seen); both shapes must round the same way
        pick = lambda want, c: next((m for m in range(3) if np.array_equal(c[:, m], want)), None)
        d = pick(dot_want, dot_c)
        m = -1
        for n in (4, 3):
            rows = (2048 // n) * n
            mm_want = np.concatenate([np.squeeze(a[i:i + n] @ b[i // n][:, None], axis=1) for i in range(0, rows, n)])
            mm_c = np.array([cands(a[i, 0], a[i, 1], b[i // n, 0], b[i // n, 1]) for i in range(rows)])
            mn = pick(mm_want, mm_c)
            m = mn if m in (-1, mn) else None
        _NUMPY_DOT_MODES = (d, m) if d is not None and m is not None else False
    return _NUMPY_DOT_MODES or None


def entity_shapes(env, ent):
    """[(kind, radius, local_verts[n,2])] of a block entity, from the native world."""
    L, w = env._lib, env._world
    max_shapes, stride = 8, 16
    kinds = (C.c_int * max_shapes)()
    radii = (C.c_double * max_shapes)()
    nverts = (C.c_int * max_shapes)()
    xy = (C.c_double * (max_shapes * stride))()
    n = nat.check(L.mgx_world_entity_shapes(w, ent.ent_id, max_shapes, kinds, radii, nverts, xy, stride))
    out = []
    for s in range(n):
        v = np.array(xy[s * stride:s * stride + 2 * nverts[s]], dtype=np.float64).reshape(-1, 2)
        out.append((kinds[s], radii[s], v))
    return out


def _shape_hits_box(kind, radius, verts, pose, bb):
    """Does the shape (at poses[M,3]) overlap the axis-aligned sensor box?  Mirrors
    space.shape_query(goal_shape) -> cpShapesCollide(...).count > 0 (entities.py:837-838):
    overlap iff the minimum separation between the cores is <= the shape's radius."""
    x, y, a = pose[:, 0], pose[:, 1], pose[:, 2]
    l, b, r, t = (np.broadcast_to(np.asarray(v, dtype=np.float64), x.shape) for v in bb)      # scalars or one box per env
    if kind == 0:   # circle
        dx = np.maximum(np.maximum(l - x, 0.0), x - r)
        dy = np.maximum(np.maximum(b - y, 0.0), y - t)
        return dx * dx + dy * dy <= radius * radius
    c, s = np.cos(a), np.sin(a)
    wx = x[:, None] + (c[:, None] * verts[None, :, 0] - s[:, None] * verts[None, :, 1])
    wy = y[:, None] + (c[:, None] * verts[None, :, 1] + s[:, None] * verts[None, :, 0])
    sep = np.maximum.reduce([l - wx.max(axis=1), wx.min(axis=1) - r, b - wy.max(axis=1), wy.min(axis=1) - t])
    corner_x, corner_y = np.stack([l, r, r, l], axis=1), np.stack([b, b, t, t], axis=1)      # [M, 4]
    n = verts.shape[0]
    for i in range(n):
        j = (i + 1) % n
        ex, ey = wx[:, j] - wx[:, i], wy[:, j] - wy[:, i]
        ln = np.sqrt(ex * ex + ey * ey)
        nx, ny = ey / ln, -ex / ln           # outward normal of a CCW polygon
        d = np.min(nx[:, None] * (corner_x - wx[:, i, None]) + ny[:, None] * (corner_y - wy[:, i, None]), axis=1)
        sep = np.maximum(sep, d)
    return sep <= radius


def shapes_of_type(env, type_id):
    """[(kind, radius, local_verts)] of a block of native shape type `type_id` (the same for every block: SHAPE_RAD)."""
    cache = env.__dict__.setdefault('_shape_library', {})
    if type_id not in cache:
        L = env._lib
        ent = next(e for e in env._entities if isinstance(e, en.Shape))
        types = np.where(env._default_shape_types >= 0, type_id, -1).astype(np.int32)
        v = C.c_void_p()
        nat.check(L.mgx_world_variant(env._world, None, types.ctypes.data_as(C.POINTER(C.c_int)), C.byref(v)))
        try:
            world, env._world = env._world, v           # entity_shapes() reads env._world
            cache[type_id] = entity_shapes(env, ent)
        finally:
            env._world = world
            L.mgx_world_destroy(v)
    return cache[type_id]


def overlapping_ents(env, goal, ents, poses):
    """GoalRegion.get_overlapping_ents(com_overlap=True) (entities.py:821-881), batched:
    bool[M, len(ents)] -- an entity counts iff EVERY one of its shapes overlaps the sensor AND its
    body position lies inside the sensor's bounding box.  In tasks with per-env worlds a block has the shape type of
    its env's episode, and blocks the episode does not have never count."""
    if poses is None:
        # the engine's own episode ends: the sets were computed on the device (k_score; per-env rectangles, worlds, shape
        # types and absent entities included) -- env._overlap = u8 [n_goals, n_entities, M], both bits set = counts
        g = env._goal_ent_idx.index(goal.ent_id)
        return (env._overlap[g][[e.ent_id for e in ents]] == 3).T
    bb = env.goal_bb(goal)
    l, b, r, t = bb
    out = np.zeros((poses.shape[0], len(ents)), dtype=bool)
    for k, ent in enumerate(ents):
        pose = poses[:, ent.body, :]
        inside = (l <= pose[:, 0]) & (r >= pose[:, 0]) & (b <= pose[:, 1]) & (t >= pose[:, 1])
        if env.variable_worlds:
            inside &= env.entity_enabled[env._scoring_envs, ent.ent_id]
            types = env.entity_shape_types[env._scoring_envs, ent.ent_id]
            groups = [(shapes_of_type(env, int(ty)), np.nonzero(inside & (types == ty))[0]) for ty in np.unique(types[inside])]
        else:
            if not hasattr(ent, '_shapes_cache'):
                ent._shapes_cache = entity_shapes(env, ent)
            # the shape tests only matter where the body position is inside the box: evaluate them on those envs alone
            groups = [(ent._shapes_cache, np.nonzero(inside)[0])]
        for shapes, sel in groups:
            if len(sel) == 0:
                continue
            sub_pose = pose[sel]
            sub_bb = tuple(v[sel] if isinstance(v, np.ndarray) and v.ndim else v for v in bb)
            ok = np.ones(len(sel), dtype=bool)
            for kind, radius, verts in shapes:
                ok &= _shape_hits_box(kind, radius, verts, sub_pose, sub_bb)
            out[sel, k] = ok
    return out
