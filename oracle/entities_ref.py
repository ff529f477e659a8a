"""World-model restatement: magical/entities.py -> calls into magical_ref.c.

TEST INFRASTRUCTURE.  Each builder cites the reference lines it follows.  The
physics objects pymunk would create (bodies / shapes / constraints) become
ref_add_* calls; the gym_render Geom tree becomes a flat painter's-order list.
"""
import ctypes as C
import math

from . import geom_ref as gt
from ._lib import lib
from .style_ref import (COLOURS_RGB, ROBOT_LINE_THICKNESS,
                        SHAPE_LINE_THICKNESS, darken_rgb, lighten_rgb)

BODY_STATIC, BODY_KINEMATIC, BODY_DYNAMIC = 0, 1, 2
G_POLY, G_LINELOOP = 0, 1
X_WORLD, X_BODY, X_EYE = 0, 1, 2
NAN = float('nan')

SHAPE_TYPES = ('triangle', 'square', 'pentagon', 'hexagon', 'octagon',
               'circle', 'star')  # entities.py:545-554
SHAPE_COLOURS = ('red', 'green', 'blue', 'yellow')  # entities.py:575-581
RAND_SHAPE_TYPES = ('square', 'pentagon', 'star', 'circle')   # entities.py:568-574: what the rand_shape* branches draw from


def _flat(verts):
    arr = (C.c_double * (2 * len(verts)))()
    for i, (x, y) in enumerate(verts):
        arr[2 * i] = x
        arr[2 * i + 1] = y
    return arr


def make_circle_verts(radius, res):  # gym_render.py:438-446
    return [(math.cos(2 * math.pi * i / res) * radius,
             math.sin(2 * math.pi * i / res) * radius) for i in range(res)]


def make_rect_verts(width, height):  # gym_render.py:449-457
    rad_h, rad_w = height / 2, width / 2
    return [(-rad_w, rad_h), (rad_w, rad_h), (rad_w, -rad_h), (-rad_w, -rad_h)]


def make_finger_vertices(upper_arm_len, forearm_len, thickness, side_sign):
    """entities.py:193-214."""
    up_shift = upper_arm_len / 2
    upper_arm_vertices = gt.rect_verts(thickness, upper_arm_len)
    forearm_vertices = gt.rect_verts(thickness, forearm_len)
    upper_start = (side_sign * thickness / 2, upper_arm_len / 2)
    forearm_offset_unrot = (-side_sign * thickness / 2, forearm_len / 2)
    rot_angle = side_sign * math.pi / 8
    off = gt.rotated(forearm_offset_unrot, rot_angle)
    forearm_trans = (upper_start[0] + off[0], upper_start[1] + off[1] + up_shift)
    forearm_final = []
    for v in forearm_vertices:
        rv = gt.rotated(v, rot_angle)
        forearm_final.append((rv[0] + forearm_trans[0], rv[1] + forearm_trans[1]))
    upper_final = [(x, y + up_shift) for x, y in upper_arm_vertices]
    return upper_final, forearm_final


class PhysVars:
    """base_env.py:49-57: defaults, and the uniform ranges `rand_dynamics` samples from (phys_vars.py:84-108)."""
    robot_pos_joint_max_force = 3
    robot_rot_joint_max_force = 1
    robot_finger_max_force = 4
    shape_trans_joint_max_force = 1.5
    shape_rot_joint_max_force = 0.1
    BOUNDS = (('robot_pos_joint_max_force', (2.2, 3.5)), ('robot_rot_joint_max_force', (0.7, 1.5)),
              ('robot_finger_max_force', (2.5, 4.5)), ('shape_trans_joint_max_force', (1.0, 1.8)),
              ('shape_rot_joint_max_force', (0.07, 0.15)))

    @classmethod
    def sample(cls, rng):
        """PhysicsVariables.sample: one rng.uniform(lower, upper) per variable, in declaration order."""
        pv = cls()
        for name, (lo, hi) in cls.BOUNDS:
            setattr(pv, name, rng.uniform(lo, hi))
        return pv


class RefWorld:
    """One pm.Space + gym_render.Viewer geom list (base_env.py:177-234)."""

    def __init__(self, phys_vars=None, phys_iter=10, backend=None):
        # backend 'pymunk': the same tables instantiated as real pymunk objects (oracle/pymunk_backend.py duck-types the C
        # library's ref_* API); raises ImportError where pymunk is not importable
        if backend == 'pymunk':
            from .pymunk_backend import PymunkBackend
            self.L = PymunkBackend()
        else:
            assert backend is None, backend
            self.L = lib()
        self.h = self.L.ref_new()
        self.phys_vars = phys_vars or PhysVars()
        # base_env.py:194-196
        self.L.ref_set_space(self.h, phys_iter, 0.01)
        bg = lighten_rgb(COLOURS_RGB['grey'], times=4)  # base_env.py:186
        self.L.ref_set_bg(self.h, *bg)
        # space.static_body
        self.static_body = self.L.ref_add_body(self.h, BODY_STATIC, 0, 0, 0, 0, 0)
        self._group_ctr = 999  # entities.py:60-66
        self.entities = []

    def __del__(self):
        try:
            self.L.ref_free(self.h)
        except Exception:
            pass

    def generate_group_id(self):
        self._group_ctr += 1
        return self._group_ctr

    # -- thin wrappers
    def body(self, btype, mass=0.0, moment=0.0, pos=(0, 0), angle=0.0):
        return self.L.ref_add_body(self.h, btype, mass, moment, pos[0], pos[1], angle)

    def poly(self, body, verts, radius=0.0, friction=0.0, group=0, sensor=0):
        return self.L.ref_add_poly(self.h, body, len(verts), _flat(verts), radius,
                                   friction, group, sensor)

    def geom_poly(self, verts, rgb, xform=X_WORLD, body=0, eye_base=(0, 0),
                  eye_body=-1, eye_pre=(0, 0)):
        return self.L.ref_add_geom(self.h, G_POLY, len(verts), _flat(verts), *rgb,
                                   xform, body, eye_base[0], eye_base[1], eye_body,
                                   eye_pre[0], eye_pre[1], 0.0, 0)

    def geom_lineloop(self, verts, rgb, width, stipple=0, xform=X_WORLD, body=0):
        return self.L.ref_add_geom(self.h, G_LINELOOP, len(verts), _flat(verts),
                                   *rgb, xform, body, 0, 0, -1, 0, 0, width, stipple)

    def add(self, ent):
        self.entities.append(ent)
        ent.setup(self)
        return ent


class ArenaBoundaries:
    """entities.py:493-537."""

    def __init__(self, left=-1, right=1, top=1, bottom=-1, seg_rad=1):
        self.left, self.right, self.top, self.bottom = left, right, top, bottom
        self.seg_rad = seg_rad
        self.shapes, self.bodies = [], []

    def setup(self, w):
        arena_body = w.body(BODY_STATIC)
        rad = self.seg_rad
        points = [(self.left - rad, self.top + rad), (self.right + rad, self.top + rad),
                  (self.right + rad, self.bottom - rad), (self.left - rad, self.bottom - rad)]
        for a, b in zip(points, points[1:] + points[:1]):
            self.shapes.append(w.L.ref_add_segment(w.h, arena_body, a[0], a[1], b[0], b[1], rad, 0.8))
        width, height = self.right - self.left, self.top - self.bottom
        rect = make_rect_verts(width, height)
        w.geom_poly(rect, (1, 1, 1))
        # PolyLine attrs are enabled in reverse add order (gym_render.py:306-311):
        # LineWidth(0.01) first, then the PolyLine's own LineWidth(1) -> 1 px.
        w.geom_lineloop(rect, COLOURS_RGB['grey'], 1.0)


class Robot:
    """entities.py:217-490."""

    def __init__(self, radius, init_pos, init_angle, mass=1.0):
        self.radius, self.init_pos, self.init_angle, self.mass = radius, init_pos, init_angle, mass
        self.finger_rot_limit_outer = math.pi / 8
        self.finger_rot_limit_inner = 0.0
        self.shapes, self.bodies = [], []

    def setup(self, w):
        L, h, pv = w.L, w.h, w.phys_vars
        inertia = gt.moment_for_circle(self.mass, 0, self.radius)
        body = self.robot_body = w.body(BODY_DYNAMIC, self.mass, inertia, self.init_pos, self.init_angle)
        control = self.control_body = w.body(BODY_KINEMATIC, 0, 0, self.init_pos, self.init_angle)
        j = L.ref_add_pivot(h, control, body, 0, 0, 0, 0)                 # :255-258
        L.ref_joint_params(h, j, NAN, 0.0, pv.robot_pos_joint_max_force)
        j = L.ref_add_gear(h, control, body, 0.0, 1.0)                    # :259-263
        L.ref_joint_params(h, j, 0.0, 2.5, pv.robot_rot_joint_max_force)
        self.pupil_bodies = []
        for _eye_side in [-1, 1]:                                          # :267-277
            eye_mass = self.mass / 10
            eye_inertia = gt.moment_for_circle(eye_mass, 0, self.radius)
            eye_body = w.body(BODY_DYNAMIC, eye_mass, eye_inertia, (0, 0), self.init_angle)
            L.ref_add_spring(h, body, eye_body, 0, 0.1, 3e-3)  # max_bias/max_force unused by springs
            self.pupil_bodies.append(eye_body)
        finger_thickness = 0.25 * self.radius                              # :280-282
        finger_upper_length = 1.1 * self.radius
        finger_lower_length = 0.7 * self.radius
        self.finger_bodies, self.finger_motors = [], []
        finger_vertices, finger_inner_vertices = [], []
        for finger_side in [-1, 1]:                                        # :288-354
            finger_verts = make_finger_vertices(finger_upper_length, finger_lower_length,
                                                finger_thickness, finger_side)
            finger_vertices.append(finger_verts)
            inner = make_finger_vertices(finger_upper_length - ROBOT_LINE_THICKNESS * 2,
                                         finger_lower_length - ROBOT_LINE_THICKNESS * 2,
                                         finger_thickness - ROBOT_LINE_THICKNESS * 2, finger_side)
            inner = [[(x, y + ROBOT_LINE_THICKNESS) for x, y in box] for box in inner]
            finger_inner_vertices.append(inner)
            if finger_side < 0:
                lower_rot_lim, upper_rot_lim = -self.finger_rot_limit_inner, self.finger_rot_limit_outer
            else:
                lower_rot_lim, upper_rot_lim = -self.finger_rot_limit_outer, self.finger_rot_limit_inner
            finger_mass = self.mass / 8
            finger_inertia = gt.moment_for_poly(finger_mass, finger_verts[0] + finger_verts[1])
            delta = upper_rot_lim if finger_side < 0 else lower_rot_lim
            finger_rel_pos = (finger_side * self.radius * 0.45, self.radius * 0.1)
            rel_rot = gt.rotated(finger_rel_pos, self.init_angle)
            fpos = (self.init_pos[0] + rel_rot[0], self.init_pos[1] + rel_rot[1])
            fb = w.body(BODY_DYNAMIC, finger_mass, finger_inertia, fpos, self.init_angle + delta)
            self.finger_bodies.append(fb)
            j = L.ref_add_pin(h, body, fb, finger_rel_pos[0], finger_rel_pos[1], 0, 0)
            L.ref_joint_params(h, j, 0.0, NAN, NAN)
            j = L.ref_add_limit(h, body, fb, lower_rot_lim, upper_rot_lim)
            L.ref_joint_params(h, j, 0.0, NAN, NAN)
            j = L.ref_add_motor(h, body, fb, 0.0)
            L.ref_joint_params(h, j, NAN, 0.0, pv.robot_finger_max_force)
            self.finger_motors.append(j)
        robot_group = 1                                                    # :358-375
        self.shapes.append(L.ref_add_circle(h, body, self.radius, 0.5, robot_group, 0))
        for fb, fverts in zip(self.finger_bodies, finger_vertices):
            for sub in fverts:
                self.shapes.append(w.poly(fb, sub, 0.0, 5.0, robot_group, 0))
        self.bodies = [body, control, *self.pupil_bodies, *self.finger_bodies]
        L.ref_set_robot(h, body, control, self.finger_bodies[0], self.finger_bodies[1],
                        self.finger_motors[0], self.finger_motors[1], self.radius,
                        self.finger_rot_limit_outer, self.finger_rot_limit_inner)
        # graphics (:377-437)
        robot_colour = COLOURS_RGB['grey']
        dark = darken_rgb(robot_colour)
        light = lighten_rgb(robot_colour, 4)
        for fb, fverts in zip(self.finger_bodies, finger_vertices):
            for sub in fverts:
                w.geom_poly(sub, robot_colour, X_BODY, fb)
        for fb, iverts in zip(self.finger_bodies, finger_inner_vertices):
            for sub in iverts:
                w.geom_poly(sub, light, X_BODY, fb)
        w.geom_poly(make_circle_verts(self.radius, 100), dark, X_BODY, body)
        w.geom_poly(make_circle_verts(self.radius - ROBOT_LINE_THICKNESS, 100), robot_colour, X_BODY, body)
        for x_sign, eye_body in zip([-1, 1], self.pupil_bodies):
            base = (x_sign * 0.4 * self.radius, 0.3 * self.radius)
            w.geom_poly(make_circle_verts(0.2 * self.radius, 20), (1.0, 1.0, 1.0), X_EYE, body,
                        eye_base=base)
            w.geom_poly(make_circle_verts(0.12 * self.radius, 10), (0.1, 0.1, 0.1), X_EYE, body,
                        eye_base=base, eye_body=eye_body, eye_pre=(0, self.radius * 0.07))


class Shape:
    """entities.py:584-761."""

    def __init__(self, shape_type, colour_name, shape_size, init_pos, init_angle, mass=0.5):
        self.shape_type, self.colour_name, self.shape_size = shape_type, colour_name, shape_size
        self.colour = COLOURS_RGB[colour_name]
        self.init_pos, self.init_angle, self.mass = init_pos, init_angle, mass
        self.shapes, self.bodies = [], []

    def setup(self, w):
        L, h, pv = w.L, w.h, w.phys_vars
        st = self.shape_type
        if st == 'square':                                                 # :620-635
            side_len = math.sqrt(math.pi) * self.shape_size
            hw = side_len / 2
            verts = [(hw, -hw), (hw, hw), (-hw, hw), (-hw, -hw)]          # cpBoxShapeInit2 order
            # body mass/moment come from shape.mass (cpBodyAccumulateMassFromShapes)
            inertia = self.mass * gt.moment_for_poly(1.0, verts)
            body = w.body(BODY_DYNAMIC, self.mass, inertia, self.init_pos, self.init_angle)
            shapes = [w.poly(body, verts, 0.01 * side_len, 0.5)]
        elif st == 'circle':                                               # :636-645
            inertia = gt.moment_for_circle(self.mass, 0, self.shape_size)
            body = w.body(BODY_DYNAMIC, self.mass, inertia, self.init_pos, self.init_angle)
            shapes = [L.ref_add_circle(h, body, self.shape_size, 0.5, 0, 0)]
        elif st == 'star':                                                 # :646-668
            star_npoints = 5
            star_out_rad = 1.3 * self.shape_size
            star_in_rad = 0.5 * star_out_rad
            star_verts = gt.compute_star_verts(star_npoints, star_out_rad, star_in_rad)
            convex_parts = gt.star_convex_parts(star_verts)
            star_hull = gt.convex_hull(star_verts)
            inertia = gt.moment_for_poly(self.mass, star_hull)
            body = w.body(BODY_DYNAMIC, self.mass, inertia, self.init_pos, self.init_angle)
            group = w.generate_group_id()
            shapes = [w.poly(body, part, 0.0, 0.5, group) for part in convex_parts]
        else:                                                              # :669-697
            factor, num_sides = {'triangle': (0.8, 3), 'pentagon': (1.0, 5),
                                 'hexagon': (1.0, 6), 'octagon': (1.0, 8)}[st]
            side_len = factor * gt.regular_poly_circ_rad_to_side_length(num_sides, self.shape_size)
            poly_verts = gt.compute_regular_poly_verts(num_sides, side_len)
            inertia = gt.moment_for_poly(self.mass, poly_verts)
            body = w.body(BODY_DYNAMIC, self.mass, inertia, self.init_pos, self.init_angle)
            shapes = [w.poly(body, poly_verts, 0.0, 0.5)]
        self.shape_body = body
        self.bodies = [body]
        self.shapes = shapes
        j = L.ref_add_pivot(h, w.static_body, body, 0, 0, 0, 0)           # :703-707
        L.ref_joint_params(h, j, NAN, 0.0, pv.shape_trans_joint_max_force)
        j = L.ref_add_gear(h, w.static_body, body, 0.0, 1.0)              # :708-711
        L.ref_joint_params(h, j, NAN, 0.0, pv.shape_rot_joint_max_force)
        # drawing (:713-757)
        T = SHAPE_LINE_THICKNESS
        if st == 'square':
            inner = [make_rect_verts(side_len - 2 * T, side_len - 2 * T)]
            outer = [make_rect_verts(side_len, side_len)]
        elif st == 'circle':
            inner = [make_circle_verts(self.shape_size - T, 100)]
            outer = [make_circle_verts(self.shape_size, 100)]
        elif st == 'star':
            short = gt.compute_star_verts(star_npoints, star_out_rad - T, star_in_rad - T)
            inner = gt.star_convex_parts(short)
            outer = convex_parts
        else:
            apothem = gt.regular_poly_side_length_to_apothem(num_sides, side_len)
            short_side = gt.regular_poly_apothem_to_side_length(num_sides, apothem - T)
            inner = [gt.compute_regular_poly_verts(num_sides, short_side)]
            outer = [poly_verts]
        dark = darken_rgb(self.colour)
        for g in outer:
            w.geom_poly(g, dark, X_BODY, body)
        for g in inner:
            w.geom_poly(g, self.colour, X_BODY, body)


class GoalRegion:
    """entities.py:769-886."""

    def __init__(self, x, y, h, w, colour_name):
        self.x, self.y, self.h, self.w, self.colour_name = x, y, h, w, colour_name
        self.base_colour = COLOURS_RGB[colour_name]
        self.shapes, self.bodies = [], []

    def setup(self, w):
        self.world = w
        pos = (self.x + self.w / 2, self.y - self.h / 2)
        self.goal_body = w.body(BODY_STATIC, 0, 0, pos, 0.0)
        hw, hh = self.w / 2, self.h / 2
        verts = [(hw, -hh), (hw, hh), (-hw, hh), (-hw, -hh)]
        self.goal_shape = w.poly(self.goal_body, verts, 0.0, 0.0, 0, 1)
        self.shapes, self.bodies = [self.goal_shape], [self.goal_body]
        self.bb = (pos[0] - hw, pos[1] - hh, pos[0] + hw, pos[1] + hh)  # l b r t
        rect = make_rect_verts(self.w, self.h)
        w.geom_poly(rect, lighten_rgb(self.base_colour, times=2), X_BODY, self.goal_body)
        w.geom_lineloop(rect, self.base_colour, 2.5, 0x00FF, X_BODY, self.goal_body)

    def get_overlapping_ents(self, ents, com_overlap=True):
        """entities.py:821-881 with com_overlap=True: an entity counts iff EVERY
        one of its shapes (a) collides with the sensor box (cpShapesCollide
        count > 0) and (b) has its body position inside the sensor's BB."""
        w = self.world
        out = (C.c_double * 19)()
        bodies = (C.c_double * (9 * w.L.ref_nbodies(w.h)))()
        w.L.ref_get_bodies(w.h, bodies)
        l, b, r, t = self.bb
        result = []
        for ent in ents:
            ok = len(ent.shapes) > 0
            for s in ent.shapes:
                cnt = w.L.ref_collide_shapes(w.h, self.goal_shape, s, out)
                bx, by = bodies[9 * ent.bodies[0]], bodies[9 * ent.bodies[0] + 1]
                inside = (l <= bx and r >= bx and b <= by and t >= by)
                if not (cnt > 0 and inside):
                    ok = False
                    break
            if ok:
                result.append(ent)
        return result
