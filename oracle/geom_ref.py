"""Vertex maths restatement (magical/geom.py:13-108) + the two Chipmunk moment
helpers the reference calls (pymunk.moment_for_circle / moment_for_poly ->
cpMomentForCircle / cpMomentForPoly).  Plain Python floats (fp64)."""
import math


def rotated(v, angle):
    """pymunk Vec2d.rotated (CCW)."""
    c, s = math.cos(angle), math.sin(angle)
    return (v[0] * c - v[1] * s, v[0] * s + v[1] * c)


def regular_poly_circumrad(n_sides, side_length):  # geom.py:13-15
    return side_length / (2 * math.sin(math.pi / n_sides))


def regular_poly_circ_rad_to_side_length(n_sides, rad):  # geom.py:18-22
    p_n = math.pi / n_sides
    return 2 * rad * math.sqrt(p_n * math.tan(p_n))


def regular_poly_apothem_to_side_length(n_sides, apothem):  # geom.py:25-27
    return 2 * apothem * math.tan(math.pi / n_sides)


def regular_poly_side_length_to_apothem(n_sides, side_length):  # geom.py:30-32
    return side_length / (2 * math.tan(math.pi / n_sides))


def compute_regular_poly_verts(n_sides, side_length):  # geom.py:35-46
    step_angle = 2 * math.pi / n_sides
    radius = regular_poly_circumrad(n_sides, side_length)
    return [rotated((0, radius), k * step_angle) for k in range(n_sides)]


def compute_star_verts(n_points, out_radius, in_radius):  # geom.py:49-63
    verts = []
    for k in range(n_points):
        verts.append(rotated((0, out_radius), k * 2 * math.pi / n_points))
        verts.append(rotated((0, in_radius), (2 * k + 1) * math.pi / n_points))
    return verts


def rect_verts(w, h):  # geom.py:101-108, CCW from top right
    return [(w / 2, h / 2), (-w / 2, h / 2), (-w / 2, -h / 2), (w / 2, -h / 2)]


def moment_for_circle(mass, inner, outer, offset=(0, 0)):
    """cpMomentForCircle."""
    return mass * (0.5 * (inner * inner + outer * outer)
                   + (offset[0]**2 + offset[1]**2))


def moment_for_poly(mass, verts, offset=(0, 0)):
    """cpMomentForPoly (ignores the bevel radius, as Chipmunk 7.0 does)."""
    sum1 = sum2 = 0.0
    n = len(verts)
    for i in range(n):
        v1 = (verts[i][0] + offset[0], verts[i][1] + offset[1])
        j = (i + 1) % n
        v2 = (verts[j][0] + offset[0], verts[j][1] + offset[1])
        a = v2[0] * v1[1] - v2[1] * v1[0]
        b = (v1[0] * v1[0] + v1[1] * v1[1]) + (v1[0] * v2[0] + v1[1] * v2[1]) \
            + (v2[0] * v2[0] + v2[1] * v2[1])
        sum1 += a * b
        sum2 += a
    return (mass * sum1) / (6.0 * sum2)


def convex_hull(points):
    """Monotone-chain hull, CCW (stands in for autogeom.to_convex_hull; only
    the vertex SET matters for the moment computed from it)."""
    pts = sorted(set(points))
    if len(pts) <= 2:
        return pts

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return lower[:-1] + upper[:-1]


def star_convex_parts(star_verts):
    """Stand-in for autogeom.convex_decomposition(star, 0) (entities.py:653-654),
    whose partition cannot be observed here (UNPINNED): 5 tip triangles
    (in_{k-1}, out_k, in_k) followed by the inner pentagon.  The union is the
    same star."""
    n = len(star_verts) // 2
    outs = star_verts[0::2]
    ins = star_verts[1::2]
    parts = [[ins[(k - 1) % n], outs[k], ins[k]] for k in range(n)]
    parts.append(list(ins))
    return parts
