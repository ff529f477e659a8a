"""GJK/EPA restatement vs an independent brute-force geometric check."""
import ctypes as C
import math

import numpy as np

from oracle.entities_ref import BODY_DYNAMIC, RefWorld


def _world_two_polys(va, vb, ra=0.0, rb=0.0):
    w = RefWorld()
    ba = w.body(BODY_DYNAMIC, 1, 1, (0, 0), 0)
    bb = w.body(BODY_DYNAMIC, 1, 1, (0, 0), 0)
    sa = w.poly(ba, va, ra, 0.5)
    sb = w.poly(bb, vb, rb, 0.5)
    return w, ba, bb, sa, sb


def _regular(n, r):
    return [(r * math.cos(2 * math.pi * k / n), r * math.sin(2 * math.pi * k / n)) for k in range(n)]


def _xf(verts, x, y, a):
    c, s = math.cos(a), math.sin(a)
    return np.array([(vx * c - vy * s + x, vx * s + vy * c + y) for vx, vy in verts])


def _sat(A, B):
    """max over face normals of both polys of the separation (exact min-penetration
    axis when overlapping; a lower bound of the distance otherwise)."""
    best = -1e30
    for P, Q in ((A, B), (B, A)):
        n = len(P)
        for i in range(n):
            e = P[(i + 1) % n] - P[i]
            nrm = np.array([e[1], -e[0]]) / np.hypot(*e)
            sep = np.min((Q - P[i]) @ nrm)
            best = max(best, sep)
    return best


def test_poly_poly_penetration_matches_sat():
    rng = np.random.RandomState(0)
    va, vb = _regular(4, 0.15), _regular(5, 0.14)
    w, ba, bb, sa, sb = _world_two_polys(va, vb)
    out = (C.c_double * 19)()
    n_overlap = n_sep = 0
    for _ in range(2000):
        xa, ya, aa = rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(-3, 3)
        xb, yb, ab = rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), rng.uniform(-3, 3)
        bodies = np.zeros((w.L.ref_nbodies(w.h), 9))
        bodies[ba, :3] = (xa, ya, aa)
        bodies[bb, :3] = (xb, yb, ab)
        w.L.ref_set_bodies(w.h, bodies.ctypes.data_as(C.POINTER(C.c_double)))
        cnt = w.L.ref_collide_shapes(w.h, sa, sb, out)
        sep = _sat(_xf(va, xa, ya, aa), _xf(vb, xb, yb, ab))
        if sep > 1e-9:
            assert cnt == 0
            n_sep += 1
        elif sep < -1e-9:
            assert cnt >= 1
            n_overlap += 1
            nx, ny = out[2], out[3]
            assert abs(math.hypot(nx, ny) - 1) < 1e-9
            # deepest contact's penetration == SAT minimum penetration
            depth = min(((out[5 + 7 * k + 2] - out[5 + 7 * k]) * nx + (out[5 + 7 * k + 3] - out[5 + 7 * k + 1]) * ny)
                        for k in range(cnt))
            assert abs(depth - sep) < 1e-9
    assert n_overlap > 200 and n_sep > 200


def test_circle_poly_distance():
    rng = np.random.RandomState(1)
    w = RefWorld()
    bc = w.body(BODY_DYNAMIC, 1, 1, (0, 0), 0)
    bp = w.body(BODY_DYNAMIC, 1, 1, (0, 0), 0)
    sc = w.L.ref_add_circle(w.h, bc, 0.12, 0.5, 0, 0)
    vp = _regular(5, 0.138)
    sp = w.poly(bp, vp, 0.0, 0.5)
    out = (C.c_double * 19)()
    hits = 0
    for _ in range(2000):
        x, y = rng.uniform(-0.35, 0.35, size=2)
        a = rng.uniform(-3, 3)
        bodies = np.zeros((w.L.ref_nbodies(w.h), 9))
        bodies[bc, :3] = (x, y, 0)
        bodies[bp, :3] = (0, 0, a)
        w.L.ref_set_bodies(w.h, bodies.ctypes.data_as(C.POINTER(C.c_double)))
        cnt = w.L.ref_collide_shapes(w.h, sc, sp, out)
        P = _xf(vp, 0, 0, a)
        # signed distance centre -> polygon
        n = len(P)
        dmin, inside = 1e30, True
        for i in range(n):
            e = P[(i + 1) % n] - P[i]
            nrm = np.array([e[1], -e[0]]) / np.hypot(*e)
            if (np.array([x, y]) - P[i]) @ nrm > 0:
                inside = False
            t = np.clip(((np.array([x, y]) - P[i]) @ e) / (e @ e), 0, 1)
            dmin = min(dmin, np.hypot(*(np.array([x, y]) - (P[i] + t * e))))
        sd = -dmin if inside else dmin
        if sd > 0.12 + 1e-9:
            assert cnt == 0
        elif sd < 0.12 - 1e-9:
            assert cnt == 1
            hits += 1
            nx, ny = out[2], out[3]
            depth = (out[7] - out[5]) * nx + (out[8] - out[6]) * ny
            if not inside:
                assert abs(depth - (sd - 0.12)) < 1e-9
    assert hits > 200


def test_block_resting_on_wall_two_contacts():
    """Square pushed flat into the left wall: segment/poly clipping yields two
    contact points on the wall face x = -1."""
    from oracle.env_ref import RefEnv
    e = RefEnv('MoveToCorner')
    e.reset()
    b = e.bodies()
    s = math.sqrt(math.pi) * 0.12
    b[8, :3] = (-1 + s / 2 - 0.004, 0.0, 0.0)
    e.set_bodies(b)
    e.substep()
    c = e.contacts()
    assert len(c) == 1 and c[0, 4] == 2
    assert abs(abs(c[0, 2]) - 1) < 1e-12 and abs(c[0, 3]) < 1e-12
    for k in range(2):
        assert abs(c[0, 5 + 7 * k] + 1.0) < 1e-9      # point on the wall face
