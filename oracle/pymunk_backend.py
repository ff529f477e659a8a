"""pymunk-backed capture path: the repo's OWN world tables driven through the third-party `pymunk` API.

TEST INFRASTRUCTURE (never imported by the product).  The physics of the reference lives in pymunk 5.6 / Chipmunk2D 7, which is
not vendored under /root/reference and is not installed in the build container or on the GPU box (probe result: DESIGN.md
section 6), so `oracle/magical_ref.c` -- a restatement of cpSpaceStep -- is PARITY UNPINNED for poses.  This module is the
only lever that can pin it: wherever `import pymunk` succeeds it instantiates the same entity tables the C oracle is built
from (`oracle/entities_ref.py`, `oracle/tasks_ref.py`) as real pymunk bodies / shapes / constraints and records what the real
engine does with them.  No file of the reference is imported, read or copied; the pymunk calls below mirror the reference's
call sites, cited per method:

    space            magical/base_env.py:194-196  (pm.Space(), collision_slop = 0.01, iterations = 10), :236-243 (space.step)
    robot            magical/entities.py:243-374  (bodies, PivotJoint / GearJoint to the kinematic control body, eye
                                                   DampedRotarySprings, per finger PinJoint + RotaryLimitJoint + SimpleMotor,
                                                   ShapeFilter(group=...), frictions)
    Robot.update     magical/entities.py:439-479  (control body angle / velocity, motor rates)
    arena            magical/entities.py:506-517  (four pm.Segment on a static body, friction 0.8)
    blocks           magical/entities.py:620-711  (Poly / Circle / star parts, PivotJoint + GearJoint to space.static_body)

`PymunkBackend` duck-types the ctypes library of the C oracle (the `ref_*` functions `entities_ref.RefWorld` calls on `w.L`), so
`RefEnv(task, backend='pymunk')` builds and steps a pymunk world with the unchanged table code.  What pymunk does not expose
(bias velocities, per-axis accumulated joint impulses) is reported as NaN, never invented.

    python -m oracle.pymunk_backend --probe                      # is pymunk importable here?  which version?
    python -m oracle.pymunk_backend --capture [out.json]         # per-substep states, masses, star parts, scores of fixed tapes
                                                                 # (default: tests/golden/pymunk_capture.json, the fixture path)

tests/test_pymunk_backend.py runs the comparison (C oracle vs pymunk, substep by substep) when pymunk is importable and skips
LOUDLY otherwise; bench.py's cpu_baseline uses it (`kind: "pymunk"`) under the same condition.  The capture FILE closes the loop for
boxes without pymunk: ONE run of the --capture command anywhere pymunk 5.6 / 5.7 imports, committed as tests/golden/pymunk_capture.json,
and tests/test_pymunk_capture.py holds both the C oracle (CPU) and the HIP engine (-m gpu) to the real engine's states.
"""
import json
import math
import sys

import numpy as np

NAN = float('nan')
BODY_STATIC, BODY_KINEMATIC, BODY_DYNAMIC = 0, 1, 2
SH_CIRCLE, SH_SEGMENT, SH_POLY = 0, 1, 2
SUPPORTED = ('5.6', '5.7')       # the reference pins pymunk==5.6.*; 5.7 is the same Chipmunk (7.0.3)


def probe():
    """(available, version or reason).  Never raises."""
    try:
        import pymunk
    except Exception as ex:      # ImportError, or a broken cffi / shared library
        return False, f'{type(ex).__name__}: {ex}'
    return True, getattr(pymunk, 'version', getattr(pymunk, '__version__', '?'))


def available():
    return probe()[0]


class PymunkBackend:
    """The `ref_*` API of oracle/libmagical_ref.so on top of pymunk.  One instance = one world; the "handle" it hands out is
    itself.  Bodies, shapes and joints are numbered in creation order exactly like the C oracle numbers them."""

    def __init__(self):
        ok, why = probe()
        if not ok:
            raise ImportError(f'pymunk is not importable here ({why}): the pymunk-backed capture path cannot run; '
                              'the C oracle stays PARITY UNPINNED for poses')
        import pymunk
        self.pm = pymunk
        self.space = None
        self.bodies, self.shapes, self.joints = [], [], []
        self.shape_type, self.shape_body = [], []
        self.robot = None
        self.rel_turn_angle = self.target_speed = self.target_finger_angle = 0.0
        self.episode_steps = 0

    # ------------------------------------------------------------ space (base_env.py:194-196)
    def ref_new(self):
        self.space = self.pm.Space()
        return self

    def ref_free(self, h):
        self.space = None

    def ref_set_space(self, h, iterations, slop):
        self.space.collision_slop = slop
        self.space.iterations = iterations

    def ref_set_bg(self, h, r, g, b):
        pass                                    # no painter behind this backend

    def ref_set_gjk_warm(self, h, on):
        pass                                    # Chipmunk always warm-starts its GJK from the cached arbiter

    # ------------------------------------------------------------ bodies
    def ref_add_body(self, h, btype, mass, moment, x, y, angle):
        pm = self.pm
        if btype == BODY_DYNAMIC:
            body = pm.Body(mass, moment)                                # entities.py:243-246, 267-271, 321-327, 620-697
        elif btype == BODY_KINEMATIC:
            body = pm.Body(body_type=pm.Body.KINEMATIC)                 # entities.py:250-253
        else:
            # the first body of a world is space.static_body (base_env.py / entities.py:703); further static bodies
            # (arena, goal regions) are bodies of their own (entities.py:506, 790)
            body = self.space.static_body if not self.bodies else pm.Body(body_type=pm.Body.STATIC)
        body.position = (x, y)
        body.angle = angle
        if body is not self.space.static_body:
            self.space.add(body)
        self.bodies.append(body)
        return len(self.bodies) - 1

    # ------------------------------------------------------------ shapes
    def _add_shape(self, shape, kind, body, friction, group, sensor):
        shape.friction = friction
        if group:
            shape.filter = self.pm.ShapeFilter(group=group)             # entities.py:358-375, 660-668
        if sensor:
            shape.sensor = True                                         # entities.py:795-797
        self.space.add(shape)
        self.shapes.append(shape); self.shape_type.append(kind); self.shape_body.append(body)
        return len(self.shapes) - 1

    def ref_add_circle(self, h, body, radius, friction, group, sensor):
        return self._add_shape(self.pm.Circle(self.bodies[body], radius), SH_CIRCLE, body, friction, group, sensor)

    def ref_add_poly(self, h, body, n, xy, radius, friction, group, sensor):
        verts = [(xy[2 * i], xy[2 * i + 1]) for i in range(n)]
        return self._add_shape(self.pm.Poly(self.bodies[body], verts, radius=radius), SH_POLY, body, friction, group, sensor)

    def ref_add_segment(self, h, body, ax, ay, bx, by, radius, friction):
        return self._add_shape(self.pm.Segment(self.bodies[body], (ax, ay), (bx, by), radius), SH_SEGMENT, body, friction, 0, 0)

    def ref_add_geom(self, *args):
        return 0

    # ------------------------------------------------------------ constraints
    def _add_joint(self, j):
        self.space.add(j)
        self.joints.append(j)
        return len(self.joints) - 1

    def ref_add_pivot(self, h, a, b, ax, ay, bx, by):                   # entities.py:255-258, 703-707
        return self._add_joint(self.pm.PivotJoint(self.bodies[a], self.bodies[b], (ax, ay), (bx, by)))

    def ref_add_gear(self, h, a, b, phase, ratio):                      # entities.py:259-263, 708-711
        return self._add_joint(self.pm.GearJoint(self.bodies[a], self.bodies[b], phase, ratio))

    def ref_add_spring(self, h, a, b, rest, k, damping):                # entities.py:272-277
        return self._add_joint(self.pm.DampedRotarySpring(self.bodies[a], self.bodies[b], rest, k, damping))

    def ref_add_pin(self, h, a, b, ax, ay, bx, by):                     # entities.py:334-341
        return self._add_joint(self.pm.PinJoint(self.bodies[a], self.bodies[b], (ax, ay), (bx, by)))

    def ref_add_limit(self, h, a, b, lo, hi):                           # entities.py:343-346
        return self._add_joint(self.pm.RotaryLimitJoint(self.bodies[a], self.bodies[b], lo, hi))

    def ref_add_motor(self, h, a, b, rate):                             # entities.py:349-354
        return self._add_joint(self.pm.SimpleMotor(self.bodies[a], self.bodies[b], rate))

    def ref_joint_params(self, h, j, error_bias, max_bias, max_force):
        c = self.joints[j]
        if not math.isnan(error_bias):
            c.error_bias = error_bias
        if not math.isnan(max_bias):
            c.max_bias = max_bias
        if not math.isnan(max_force):
            c.max_force = max_force

    def ref_set_robot(self, h, robot_body, control_body, finger_l, finger_r, motor_l, motor_r, radius, lim_outer, lim_inner):
        self.robot = dict(body=robot_body, control=control_body, fingers=(finger_l, finger_r), motors=(motor_l, motor_r),
                          radius=radius, lim_outer=lim_outer, lim_inner=lim_inner)

    # ------------------------------------------------------------ env step
    def ref_set_action(self, h, action):                                # entities.py:148-190, 439-457
        ud, lr, grip = action % 3, (action // 3) % 3, action // 9
        r = self.robot['radius']
        self.rel_turn_angle, self.target_speed = 0.0, 0.0
        if ud == 1:
            self.target_speed += 4.0 * r
        if ud == 2:
            self.target_speed -= 3.0 * r
        if lr == 1:
            self.rel_turn_angle += 1.5
        if lr == 2:
            self.rel_turn_angle -= 1.5
        self.target_finger_angle = self.robot['lim_outer'] if grip == 0 else -self.robot['lim_inner']

    def ref_robot_update(self, h):                                      # entities.py:459-479
        if self.robot is None:
            return
        rb, cb = self.bodies[self.robot['body']], self.bodies[self.robot['control']]
        cb.angle = rb.angle + self.rel_turn_angle
        cb.velocity = rb.rotation_vector.cpvrotate((0.0, self.target_speed))
        for f in range(2):
            side = -1.0 if f == 0 else 1.0
            fb = self.bodies[self.robot['fingers'][f]]
            angle_error = (fb.angle - rb.angle) + side * self.target_finger_angle
            target_rate = max(-1.0, min(1.0, angle_error * 10))
            if abs(target_rate) < 1e-4:
                target_rate = 0.0
            self.joints[self.robot['motors'][f]].rate = target_rate

    def ref_space_step(self, h, dt):
        self.space.step(dt)

    def ref_substep(self, h, dt):                                       # base_env.py:236-243
        self.ref_robot_update(h)
        self.space.step(dt)

    def ref_step(self, h, action, fps):                                 # base_env.py:255-270
        self.ref_set_action(h, action)
        dt = (1.0 / fps) / 10
        for _ in range(10):
            self.ref_substep(h, dt)
        self.episode_steps += 1

    # ------------------------------------------------------------ state access
    def ref_nbodies(self, h):
        return len(self.bodies)

    def ref_nshapes(self, h):
        return len(self.shapes)

    def ref_njoints(self, h):
        return len(self.joints)

    def ref_episode_steps(self, h):
        return self.episode_steps

    @staticmethod
    def _buf(out):
        """ctypes array / pointer or numpy array -> something indexable for writing."""
        return out

    def ref_get_bodies(self, h, out):
        """out[nbodies][9] = x y a vx vy w vbx vby wb; the bias velocities are internal to Chipmunk: NaN."""
        for i, b in enumerate(self.bodies):
            vals = (b.position.x, b.position.y, b.angle, b.velocity.x, b.velocity.y, b.angular_velocity, NAN, NAN, NAN)
            for k, v in enumerate(vals):
                out[9 * i + k] = v

    def ref_set_bodies(self, h, arr):
        for i, b in enumerate(self.bodies):
            if b.body_type == self.pm.Body.STATIC:
                continue
            b.position = (arr[9 * i], arr[9 * i + 1])
            b.angle = arr[9 * i + 2]
            b.velocity = (arr[9 * i + 3], arr[9 * i + 4])
            b.angular_velocity = arr[9 * i + 5]
            self.space.reindex_shapes_for_body(b)

    def ref_get_body_mass(self, h, out):
        for i, b in enumerate(self.bodies):
            dyn = b.body_type == self.pm.Body.DYNAMIC
            out[2 * i] = 1.0 / b.mass if dyn else 0.0
            out[2 * i + 1] = 1.0 / b.moment if dyn else 0.0

    def ref_get_joint_acc(self, h, out):
        """Chipmunk exposes only |accumulated impulse| (cpConstraintGetImpulse): column 0 holds that magnitude, column 1 NaN."""
        for i, c in enumerate(self.joints):
            out[2 * i] = c.impulse
            out[2 * i + 1] = NAN

    def _arbiters(self):
        """The space's active arbiters, once each, in the order bodies / arbiters are met (pymunk has no cpSpaceEachArbiter)."""
        seen, rows = set(), []

        def visit(arb):
            sa, sb = arb.shapes
            key = (id(sa), id(sb))
            if key in seen or (key[1], key[0]) in seen:
                return
            seen.add(key)
            rows.append(arb)
        for b in self.bodies:
            b.each_arbiter(visit)
        return rows

    def ref_narbiters(self, h):
        return len(self._arbiters())

    def ref_get_contacts(self, h, out, max_rows):
        """rows of [shape_a, shape_b, nx, ny, count, (p1x p1y p2x p2y jn jt hash) x 2]; per-point impulses and hashes are not
        exposed by pymunk 5.x (only the arbiter's total impulse): NaN."""
        index = {id(s): i for i, s in enumerate(self.shapes)}
        rows = 0
        for arb in self._arbiters():
            if rows >= max_rows:
                break
            cps = arb.contact_point_set
            sa, sb = arb.shapes
            o = [0.0] * 19
            o[0], o[1], o[2], o[3], o[4] = index[id(sa)], index[id(sb)], cps.normal.x, cps.normal.y, len(cps.points)
            for k, p in enumerate(cps.points[:2]):
                o[5 + 7 * k: 12 + 7 * k] = [p.point_a.x, p.point_a.y, p.point_b.x, p.point_b.y, NAN, NAN, NAN]
            for k, v in enumerate(o):
                out[19 * rows + k] = v
            rows += 1
        return rows

    def ref_collide_shapes(self, h, i, j, out):
        """cpShapesCollide on the current poses (GoalRegion.get_overlapping_ents, entities.py:821-881)."""
        sa, sb = (i, j) if self.shape_type[i] <= self.shape_type[j] else (j, i)
        for s in (sa, sb):
            self.shapes[s].cache_bb()
        cps = self.shapes[sa].shapes_collide(self.shapes[sb])
        o = [0.0] * 19
        o[0], o[1], o[2], o[3], o[4] = sa, sb, cps.normal.x, cps.normal.y, len(cps.points)
        for k, p in enumerate(cps.points[:2]):
            o[5 + 7 * k: 12 + 7 * k] = [p.point_a.x, p.point_a.y, p.point_b.x, p.point_b.y, NAN, NAN, NAN]
        for k, v in enumerate(o):
            out[k] = v
        return len(cps.points)

    def ref_shape_info(self, h, s, out):
        sh = self.shapes[s]
        out[0], out[1], out[2] = self.shape_body[s], int(sh.filter.group), int(bool(sh.sensor))

    def ref_shape_world(self, h, s, out_xy, out_r, out_type):
        sh, kind = self.shapes[s], self.shape_type[s]
        body = self.bodies[self.shape_body[s]]
        if kind == SH_CIRCLE:
            pts = [body.local_to_world(sh.offset)]
        elif kind == SH_SEGMENT:
            pts = [body.local_to_world(sh.a), body.local_to_world(sh.b)]
        else:
            pts = [body.local_to_world(v) for v in sh.get_vertices()]
        for i, p in enumerate(pts):
            out_xy[2 * i], out_xy[2 * i + 1] = p.x, p.y
        _set_scalar(out_r, sh.radius)
        _set_scalar(out_type, kind)
        return len(pts)

    def ref_render(self, *args):
        raise NotImplementedError('the pymunk backend has no painter: render with the C oracle on the captured poses')


def _set_scalar(ref, value):
    """Write through a ctypes byref() / pointer or a one-element list."""
    try:
        ref._obj.value = value                  # ctypes.byref(x)
    except AttributeError:
        try:
            ref.contents.value = value          # ctypes.pointer(x)
        except AttributeError:
            ref[0] = value


# ---------------------------------------------------------------------------------------------------------------------
# What the reference's OWN construction calls give for masses and moments (entities.py builds some bodies from shape masses),
# next to the formulas the oracle's tables use -- and pymunk's decomposition of the star.

def reference_mass_table(shape_size=0.12, mass=0.5, robot_radius=0.2, robot_mass=1.0):
    """{name: (mass, moment)} as pymunk computes them along the reference's construction paths."""
    import pymunk as pm
    from . import geom_ref as gt
    out = {}
    out['robot'] = (robot_mass, pm.moment_for_circle(robot_mass, 0, robot_radius))            # entities.py:243-246
    out['eye'] = (robot_mass / 10, pm.moment_for_circle(robot_mass / 10, 0, robot_radius))    # entities.py:267-271
    # square: pm.Body() + Poly.create_box(..., radius) with shape.mass = mass (entities.py:620-635): mass and moment are
    # accumulated from the shape, bevel radius included
    side = math.sqrt(math.pi) * shape_size
    body = pm.Body()
    box = pm.Poly.create_box(body, (side, side), 0.01 * side)
    box.mass = mass
    space = pm.Space(); space.add(body, box)
    out['square'] = (body.mass, body.moment)
    out['circle'] = (mass, pm.moment_for_circle(mass, 0, shape_size))                         # entities.py:636-645
    for name, (factor, n) in {'triangle': (0.8, 3), 'pentagon': (1.0, 5), 'hexagon': (1.0, 6), 'octagon': (1.0, 8)}.items():
        side_len = factor * gt.regular_poly_circ_rad_to_side_length(n, shape_size)
        verts = gt.compute_regular_poly_verts(n, side_len)
        out[name] = (mass, pm.moment_for_poly(mass, verts))                                   # entities.py:669-697
    star = gt.compute_star_verts(5, 1.3 * shape_size, 0.65 * shape_size)
    out['star'] = (mass, pm.moment_for_poly(mass, gt.convex_hull(star)))                      # entities.py:646-668
    return {k: (float(m), float(i)) for k, (m, i) in out.items()}


def reference_star_parts(shape_size=0.12):
    """pm.autogeometry.convex_decomposition of the star outline, as entities.py:655-660 calls it (closed loop, tolerance 0)."""
    import pymunk as pm
    from pymunk import autogeometry
    from . import geom_ref as gt
    star = gt.compute_star_verts(5, 1.3 * shape_size, 0.65 * shape_size)
    parts = autogeometry.convex_decomposition(list(star) + [star[0]], 0)
    return [[(float(v[0]), float(v[1])) for v in part] for part in parts]


# ---------------------------------------------------------------------------------------------------------------------
# Capture: per-substep states of fixed tapes

def fixed_tape(task, steps, seed=0):
    import zlib
    return np.random.RandomState(zlib.crc32(task.encode()) % 10000 + seed).randint(0, 18, size=steps).astype(int)


def capture_task(task, steps=20, backend='pymunk', seed=0):
    """One Demo episode piece of `task` under `backend` ('pymunk' or 'c'): body states [steps * 10 + 1, n_bodies, 6] after
    every substep (x y angle vx vy w), inverse masses / moments, the score of the state reached."""
    from .env_ref import FPS, RefEnv
    env = RefEnv(task, backend=None if backend == 'c' else backend)
    env.reset()
    tape = fixed_tape(task, steps, seed)
    states = [env.bodies()[:, :6].copy()]
    for a in tape:
        env.set_action(int(a))
        for _ in range(10):
            env.substep(1.0 / FPS / 10)
            states.append(env.bodies()[:, :6].copy())
    return {'task': task, 'backend': backend, 'tape': [int(a) for a in tape], 'states': np.array(states),
            'mass': env.body_mass(), 'score': float(env.task.score_on_end_of_traj())}


CAPTURE_FIXTURE = 'tests/golden/pymunk_capture.json'       # where the consumer tests (tests/test_pymunk_capture.py) look for a capture


def capture_all(steps=20, backend='pymunk'):
    """The capture file's content.  backend 'pymunk': the real engine (what pins the oracle); 'c': the C oracle capturing itself --
    only to exercise the consumer tests' plumbing where pymunk cannot be imported (such a file says so and pins nothing)."""
    from .tasks_ref import TASKS
    if backend == 'pymunk':
        ok, version = probe()
        if not ok:
            raise ImportError(version)
        out = {'backend': 'pymunk', 'pymunk_version': str(version), 'mass_table': reference_mass_table(), 'star_parts': reference_star_parts(), 'tasks': {}}
    else:
        out = {'backend': 'c', 'pymunk_version': None, 'mass_table': None, 'star_parts': None, 'tasks': {}}
    for task in TASKS:
        rec = capture_task(task, steps, backend=backend)
        out['tasks'][task] = {'tape': rec['tape'], 'score': rec['score'], 'mass': rec['mass'].tolist(),
                              'states_hex': [[[float(v).hex() for v in body] for body in st] for st in rec['states']]}
    return out


def load_capture(path):
    """A capture file back as arrays: {'backend', 'pymunk_version', 'mass_table', 'star_parts', 'tasks': {task: {'tape' int[T],
    'score', 'mass' f64[n_bodies, 2], 'states' f64[10 T + 1, n_bodies, 6]}}} -- states after every substep, bit for bit (hex)."""
    with open(path) as f:
        cap = json.load(f)
    for rec in cap['tasks'].values():
        rec['states'] = np.array([[[float.fromhex(v) for v in body] for body in st] for st in rec.pop('states_hex')], dtype=np.float64)
        rec['tape'] = np.array(rec['tape'], dtype=int)
        rec['mass'] = np.array(rec['mass'], dtype=np.float64)
    return cap


def main(argv):
    ok, info = probe()
    if '--probe' in argv or len(argv) == 0:
        print(json.dumps({'pymunk_importable': ok, 'detail': str(info)}))
        return 0 if ok else 3
    if '--capture' in argv:
        k = argv.index('--capture')
        path = argv[k + 1] if k + 1 < len(argv) and not argv[k + 1].startswith('--') else CAPTURE_FIXTURE
        backend = argv[argv.index('--backend') + 1] if '--backend' in argv else 'pymunk'
        if backend == 'pymunk' and not ok:
            print(f'pymunk is not importable here ({info}): nothing captured', file=sys.stderr)
            return 3
        with open(path, 'w') as f:
            json.dump(capture_all(backend=backend), f)
        print(f'wrote {path} (' + (f'pymunk {info}' if backend == 'pymunk' else 'the C oracle capturing itself: pins nothing') + ')')
        return 0
    print(__doc__)
    return 2


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
