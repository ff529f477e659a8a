"""Known-answer tests pinning the CPU oracle (oracle/) -- the reference's own
tests hold no numeric vectors, so these analytic cases are what
the restatement is anchored on (SURVEY.md §4 'what our build's tests must add').
"""
import math

import numpy as np
import pytest

from oracle import geom_ref as gt
from oracle.entities_ref import make_finger_vertices
from oracle.env_ref import RefEnv

DT = 1.0 / 8 / 10
ROBOT, CONTROL, EYE_L, EYE_R, FINGER_L, FINGER_R, BLOCK = 2, 3, 4, 5, 6, 7, 8   # MoveToCorner


def _idx(e):
    r = e.task.robot
    return (r.robot_body, r.control_body, r.pupil_bodies[0], r.pupil_bodies[1],
            r.finger_bodies[0], r.finger_bodies[1])


def test_appendix_d_constants():
    # SURVEY.md Appendix D (recomputed there from the reference formulas)
    s = math.sqrt(math.pi) * 0.12
    assert abs(s - 0.2126945) < 1e-7
    hw = s / 2
    sq = [(hw, -hw), (hw, hw), (-hw, hw), (-hw, -hw)]
    assert abs(0.5 * gt.moment_for_poly(1.0, sq) - 0.00376991) < 1e-8
    pent_side = gt.regular_poly_circ_rad_to_side_length(5, 0.12)
    assert abs(pent_side - 0.1621555) < 1e-7
    pv = gt.compute_regular_poly_verts(5, pent_side)
    assert abs(gt.moment_for_poly(0.5, pv) - 0.00366110) < 1e-8
    tri = gt.compute_regular_poly_verts(3, 0.8 * gt.regular_poly_circ_rad_to_side_length(3, 0.12))
    assert abs(gt.moment_for_poly(0.5, tri) - 0.00278600) < 1e-8
    star = gt.compute_star_verts(5, 0.156, 0.078)
    assert abs(gt.moment_for_poly(0.5, gt.convex_hull(star)) - 0.00468269) < 1e-8
    assert abs(gt.moment_for_circle(0.5, 0, 0.12) - 0.0036) < 1e-12
    fl = make_finger_vertices(0.22, 0.14, 0.05, -1)
    fr = make_finger_vertices(0.22, 0.14, 0.05, 1)
    assert abs(gt.moment_for_poly(0.125, fl[0] + fl[1]) - 0.00500649) < 1e-8
    assert abs(gt.moment_for_poly(0.125, fr[0] + fr[1]) - 0.00498302) < 1e-8
    assert abs(fl[1][0][0] - 0.07477) < 1e-5 and abs(fl[1][0][1] - 0.33021) < 1e-5


def test_world_counts_movetocorner():
    e = RefEnv('MoveToCorner')
    e.reset()
    # static_body, arena, robot, control, 2 eyes, 2 fingers, block
    assert e.L.ref_nbodies(e.h) == 9
    assert e.L.ref_nshapes(e.h) == 10      # 4 walls + circle + 4 finger quads + block
    assert e.L.ref_njoints(e.h) == 12
    m = e.body_mass()
    assert m[ROBOT, 0] == 1.0 and abs(m[ROBOT, 1] - 1 / 0.02) < 1e-9
    assert m[CONTROL, 0] == 0 and m[CONTROL, 1] == 0
    assert abs(m[EYE_L, 1] - 1 / 0.002) < 1e-9
    assert abs(m[BLOCK, 0] - 2.0) < 1e-12


def test_world_counts_clustercolour():
    e = RefEnv('ClusterColour')
    e.reset()
    assert e.L.ref_nbodies(e.h) == 2 + 8 + 6
    # walls 4 + blocks (1+6+1+1+1+1+6+1) + robot 5
    assert e.L.ref_nshapes(e.h) == 4 + 18 + 5
    assert e.L.ref_njoints(e.h) == 16 + 10


def test_robot_forward_speed_and_accel():
    """PivotJoint(control, robot) max_force 3, mass 1 -> accel 3 u/s^2 up to
    4*r = 0.8 u/s forward, 0.6 back (entities.py:439-447).
    Fingers (2 x 0.125) are dragged along, so the early acceleration is a bit
    below 3 and the first substep gives exactly 3*dt (pins inactive: zero
    length)."""
    e = RefEnv('MoveToRegion')   # no blocks in the way
    e.reset()
    ROBOT, CONTROL, EYE_L, EYE_R, FINGER_L, FINGER_R = _idx(e)
    e.set_bodies(_place_robot(e, 0.5, -0.7, 0.0))
    e.set_action(1)
    e.substep()
    b = e.bodies()
    assert abs(np.hypot(b[ROBOT, 3], b[ROBOT, 4]) - 3 * DT) < 1e-12
    assert abs(b[ROBOT, 3]) < 1e-12          # forward is body +y at angle 0
    for _ in range(79):
        e.substep()
    b = e.bodies()
    assert abs(b[ROBOT, 4] - 0.8) < 1e-3
    e.set_action(2)
    for _ in range(100):
        e.substep()
    b = e.bodies()
    assert abs(b[ROBOT, 4] + 0.6) < 1e-3


def test_robot_turn_rate():
    """GearJoint max_bias 2.5 rad/s, max_force 1, I=0.02 (entities.py:259-263)."""
    e = RefEnv('MoveToRegion')
    e.reset()
    ROBOT, CONTROL, EYE_L, EYE_R, FINGER_L, FINGER_R = _idx(e)
    e.set_bodies(_place_robot(e, 0.0, 0.0, 0.0))
    e.set_action(3)   # LEFT
    for _ in range(40):
        e.substep()
    assert abs(e.bodies()[ROBOT, 5] - 2.5) < 2e-3
    e.set_action(6)   # RIGHT
    for _ in range(60):
        e.substep()
    assert abs(e.bodies()[ROBOT, 5] + 2.5) < 2e-3


def test_finger_limits():
    """RotaryLimitJoint: left finger in [0, pi/8], right in [-pi/8, 0]
    (entities.py:307-312,343-346); CLOSE drives both to 0, OPEN to +-pi/8."""
    e = RefEnv('MoveToRegion')
    e.reset()
    ROBOT, CONTROL, EYE_L, EYE_R, FINGER_L, FINGER_R = _idx(e)
    e.set_bodies(_place_robot(e, 0.0, 0.0, 0.0))
    for _ in range(10):
        e.step(9)   # Close
    b = e.bodies()
    assert abs((b[FINGER_L, 2] - b[ROBOT, 2])) < 5e-3
    assert abs((b[FINGER_R, 2] - b[ROBOT, 2])) < 5e-3
    for _ in range(10):
        e.step(0)   # Open
    b = e.bodies()
    assert abs((b[FINGER_L, 2] - b[ROBOT, 2]) - math.pi / 8) < 5e-3
    assert abs((b[FINGER_R, 2] - b[ROBOT, 2]) + math.pi / 8) < 5e-3
    # pins hold the finger roots on the robot
    root = np.array([b[ROBOT, 0] - 0.09, b[ROBOT, 1] + 0.02])
    assert np.linalg.norm(b[FINGER_L, :2] - root) < 1e-3


def test_block_friction_deceleration():
    """Block PivotJoint to static body: max_force 1.5, m 0.5 -> |dv| = 3*dt per
    substep; GearJoint max_force 0.1 -> |dw| = 0.1/I*dt (entities.py:703-711)."""
    e = RefEnv('MoveToCorner')
    e.reset()
    b = e.bodies()
    b[BLOCK, 3:6] = (0.3, 0.4, 3.0)
    e.set_bodies(b)
    inertia = 1.0 / e.body_mass()[BLOCK, 1]
    for k in range(1, 6):
        e.substep()
        bb = e.bodies()
        assert abs(np.hypot(bb[BLOCK, 3], bb[BLOCK, 4]) - (0.5 - 3.0 * DT * k)) < 1e-12
        assert abs(bb[BLOCK, 5] - (3.0 - 0.1 / inertia * DT * k)) < 1e-12
    for _ in range(40):
        e.substep()
    bb = e.bodies()
    assert abs(bb[BLOCK, 3]) < 1e-12 and abs(bb[BLOCK, 5]) < 1e-12


def test_walls_contain_robot():
    """Arena inner faces at +-1 (entities.py:502-517); slop 0.01."""
    e = RefEnv('MoveToRegion')
    e.reset()
    ROBOT, CONTROL, EYE_L, EYE_R, FINGER_L, FINGER_R = _idx(e)
    e.set_bodies(_place_robot(e, 0.0, 0.5, math.pi))   # facing down, fingers trail
    for _ in range(40):
        e.step(2)  # backwards = +y
    b = e.bodies()
    assert 0.78 < b[ROBOT, 1] <= 0.8 + 0.011
    assert e.L.ref_narbiters(e.h) >= 1


def test_block_pushed_and_score_bitexact_formula():
    e = RefEnv('MoveToCorner')
    e.reset()
    done = False
    rng = np.random.RandomState(0)
    n = 0
    while not done:
        _, done, info = e.step(rng.randint(18))
        n += 1
    assert n == 80
    x, y = e.bodies()[BLOCK, :2]
    d = np.linalg.norm(np.asarray([-1.0, 1.0]) - np.asarray([x, y]))
    expect = min(1.0, max(0.0, np.sqrt(2) - d) / (np.sqrt(2) - np.sqrt(2) / 2))
    assert info['eval_score'] == expect


def test_determinism_and_gjk_cache_independence():
    tape = np.random.RandomState(1).randint(0, 18, size=60)
    outs = []
    for warm in (True, True, False):
        e = RefEnv('ClusterColour', gjk_warm=warm)
        e.reset()
        for a in tape:
            e.step(a)
        outs.append(e.bodies().copy())
    assert np.array_equal(outs[0], outs[1])
    # warm-started vs cold GJK reach the same minimum-separation axis; only the
    # path differs, so trajectories agree to round-off amplified by contacts
    assert np.abs(outs[0] - outs[2])[:, :3].max() < 1e-6


@pytest.mark.parametrize('task', ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine',
                                  'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape'])
def test_rollout_length_all_tasks(task):
    """Mirror of the reference's only test (tests/test_rollout_preproc.py:17-36)."""
    e = RefEnv(task)
    e.reset()
    rng = np.random.RandomState(42)
    done, n = False, 0
    while not done:
        _, done, info = e.step(rng.randint(18))
        n += 1
        assert np.isfinite(e.bodies()).all()
    assert n == e.max_episode_steps
    assert 0.0 <= info['eval_score'] <= 1.0


def _place_robot(e, x, y, angle):
    ROBOT, CONTROL, EYE_L, EYE_R, FINGER_L, FINGER_R = _idx(e)
    b = e.bodies()
    b[ROBOT, :3] = (x, y, angle)
    b[CONTROL, :3] = (x, y, angle)
    b[EYE_L, 2] = b[EYE_R, 2] = angle
    c, s = math.cos(angle), math.sin(angle)
    for idx, side in ((FINGER_L, -1), (FINGER_R, 1)):
        ax, ay = side * 0.2 * 0.45, 0.2 * 0.1     # entities.py:324-325, bit-exact with the pin anchor
        b[idx, 0] = (ax * c - ay * s) + x
        b[idx, 1] = (ax * s + ay * c) + y
        b[idx, 2] = angle + (math.pi / 8 if side < 0 else -math.pi / 8)
    b[:, 3:] = 0
    return b
