"""The reference-derived pin: tests/golden/reference_vectors.json holds OUTPUTS OF THE REFERENCE'S OWN CODE (qxcv/magical's
style.py, phys_vars.py, and the pymunk-free functions / methods of geom.py, entities.py, base_env.py, benchmarks/*.py,
executed in the build container by tests/golden/make_reference_vectors.py).  Here the oracle (CPU tests) and the product
(host code on CPU where it needs no device, the engine under `-m gpu`) are checked against it, bit for bit.

What this pins: palette, force-limit sampling, polygon sizing, goal-size draws, the action table, the env-name grammar and
the whole registration table, longest_line, and the arithmetic of every score_on_end_of_traj() (for the region tasks: given
the overlap set).  What it cannot pin: anything computed inside pymunk / Chipmunk, pyglet / GL or cv2.
"""
import ctypes as C
import itertools
import json
import math
import os
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    with open(os.path.join(HERE, 'golden', 'reference_vectors.json')) as f:
        return json.load(f)


FX = _load()


def unhex(v):
    if isinstance(v, list):
        return [unhex(x) for x in v]
    return float.fromhex(v)


def arr(v):
    return np.asarray(unhex(v), dtype=np.float64)


# ====================================================================== oracle + product host code (CPU)
def test_fixture_is_current_with_the_reference():
    """In the build container the fixture is regenerated from /root/reference and must equal the committed file."""
    if not os.path.isdir('/root/reference/magical'):
        pytest.skip('reference only exists in the build container')
    import subprocess
    import sys
    import tempfile
    gen = os.path.join(HERE, 'golden', 'make_reference_vectors.py')
    src = open(gen).read().replace("OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_vectors.json')", 'OUT = os.environ["MGX_REFVEC_OUT"]')
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'v.json')
        subprocess.check_call([sys.executable, '-c', src], env=dict(os.environ, MGX_REFVEC_OUT=out), stdout=subprocess.DEVNULL)
        assert json.load(open(out)) == FX


def test_style_oracle_and_product_palette():
    from oracle import style_ref
    st = FX['style']
    for n in st['colour_names']:
        assert list(style_ref.COLOURS_RGB[n]) == unhex(st['COLOURS_RGB'][n]), n
        assert list(style_ref.darken_rgb(style_ref.COLOURS_RGB[n])) == unhex(st['darken_rgb'][n]), n
        for t in (1, 2, 4):
            assert list(style_ref.lighten_rgb(style_ref.COLOURS_RGB[n], t)) == unhex(st['lighten_rgb'][str(t)][n]), (n, t)
    for k in ('GOAL_LINE_THICKNESS', 'SHAPE_LINE_THICKNESS', 'ROBOT_LINE_THICKNESS', 'ARENA_ZOOM_OUT'):
        assert getattr(style_ref, k) == unhex(st[k]), k
    # product: the RGB8 palette the rasteriser paints with (GL float colour -> u8 framebuffer: round to nearest; that
    # conversion is the GL driver's, i.e. unpinned) in the three roles outline / fill / goal interior
    from magical_amd import _native, entities as en
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('HIP library not built')
    L = _native.lib()
    u8 = lambda c: [int(v * 255.0 + 0.5) for v in unhex(c)]
    for name, cid in en.COLOUR_ID.items():
        name = str(getattr(name, 'value', name))
        for role, want in ((0, u8(st['darken_rgb'][name])), (1, u8(st['COLOURS_RGB'][name])), (2, u8(st['lighten_rgb']['2'][name]))):
            rgb = _native.check(L.mgx_world_palette(cid, role))
            assert [rgb & 0xFF, (rgb >> 8) & 0xFF, (rgb >> 16) & 0xFF] == want, (name, role)
    assert u8(st['lighten_rgb']['4']['grey']) == [231, 231, 234]          # the background (base_env.py:186)


def test_phys_vars_oracle_and_product_draws():
    from oracle.entities_ref import PhysVars
    from magical_amd.base_env import PhysicsVariables
    pv = FX['phys_vars']
    assert [n for n, _ in PhysVars.BOUNDS] == pv['names'] == list(PhysicsVariables.NAMES)
    assert [float(getattr(PhysVars, n)) for n in pv['names']] == unhex(pv['defaults']) == PhysicsVariables.defaults()
    for n, (lo, hi) in PhysVars.BOUNDS:
        assert [lo, hi] == unhex(pv['bounds'][n]) == list(getattr(PhysicsVariables, n)[1])
    for seed, draws in pv['samples'].items():
        r1, r2 = np.random.RandomState(int(seed)), np.random.RandomState(int(seed))
        for want in draws:
            o = PhysVars.sample(r1)
            assert [getattr(o, n) for n in pv['names']] == unhex(want), seed
            assert PhysicsVariables.sample(r2) == unhex(want), seed


def test_geom_oracle_and_product():
    from oracle import geom_ref, tasks_ref
    from magical_amd import geom
    g = FX['geom']
    lens = unhex(g['lengths'])
    names = {'regular_poly_circumrad': 'regular_poly_circumrad', 'regular_poly_circ_rad_to_side_length': 'regular_poly_circ_rad_to_side_length',
             'regular_poly_apothem_to_side_legnth': 'regular_poly_apothem_to_side_length', 'regular_poly_side_length_to_apothem': 'regular_poly_side_length_to_apothem'}
    for ref_name, ours in names.items():
        for i, n in enumerate(g['n_sides']):
            for j, l in enumerate(lens):
                assert getattr(geom_ref, ours)(n, l) == unhex(g[ref_name][i][j]), (ref_name, n, l)
    for c in g['randomise_hw']:
        for fn in (tasks_ref.randomise_hw, geom.randomise_hw):
            rng = np.random.RandomState(c['seed'])
            bound = None if c['linf_bound'] is None else unhex(c['linf_bound'])
            draws = [list(map(float, fn(unhex(c['min']), unhex(c['max']), rng, current_hw=tuple(unhex(c['current_hw'])), linf_bound=bound))) for _ in range(3)]
            assert draws == unhex(c['draws']), (fn.__module__, c['seed'])
            assert int(rng.randint(0, 2 ** 31 - 1)) == c['next_u32']             # the stream advanced exactly as far


def test_entities_tables():
    from magical_amd import entities as en
    e = FX['entities']
    assert {m.name: int(m) for m in en.RobotAction} == e['RobotAction']
    assert [[i, [int(f) for f in fl], nm] for i, fl, nm in en.ACTION_NUMS_FLAGS_NAMES] == e['ACTION_NUMS_FLAGS_NAMES']
    for flags, i in e['FLAGS_TO_ACTION_ID']:
        assert en.FLAGS_TO_ACTION_ID[tuple(en.RobotAction(f) for f in flags)] == i
    assert [m.value for m in en.ShapeType] == e['ShapeType'] and [m.value for m in en.ShapeColour] == e['ShapeColour']
    assert [str(getattr(v, 'value', v)) for v in en.SHAPE_TYPE_NAMES] == e['SHAPE_TYPES']
    assert [str(getattr(v, 'value', v)) for v in en.SHAPE_COLOUR_NAMES] == e['SHAPE_COLOURS']
    # the id = 9 * [close] + 3 * lr + ud decode the step kernel uses (csrc/mgx_sim.h ph_control) is this table
    A = e['RobotAction']
    for i, (ud, lr, grip), _ in e['ACTION_NUMS_FLAGS_NAMES']:
        assert (i % 3, (i // 3) % 3, i // 9) == ({A['NONE']: 0, A['UP']: 1, A['DOWN']: 2}[ud], {A['NONE']: 0, A['LEFT']: 1, A['RIGHT']: 2}[lr],
                                                 {A['OPEN']: 0, A['CLOSE']: 1}[grip])
    from oracle import tasks_ref
    assert list(tasks_ref.SHAPE_COLOURS) == e['SHAPE_COLOURS'] and list(tasks_ref.RAND_SHAPE_TYPES) == e['SHAPE_TYPES']


def test_registry_equals_the_reference_table():
    import magical_amd as m
    from magical_amd import benchmarks as b
    m.register_envs()
    r = FX['registry']
    assert m.ALL_REGISTERED_ENVS == r['ALL_REGISTERED_ENVS']
    assert b.AVAILABLE_PREPROCESSORS == r['preprocessors'] and list(b.DEFAULT_RES) == r['DEFAULT_RES']
    assert {k: (list(v) if isinstance(v, tuple) else v) for k, v in b.COMMON_KWARGS.items()} == r['common_kwargs']
    assert [[k, list(v)] for k, v in m.DEMO_ENVS_TO_TEST_ENVS_MAP.items()] == r['DEMO_ENVS_TO_TEST_ENVS_MAP']
    for epoint, name, ep_len, kwargs in r['specs']:
        spec = b._SPECS[name]
        mod, cls = epoint.split(':')
        assert mod == f'magical.benchmarks.{spec["module"]}' and cls == spec['cls'], name
        assert spec['ep_len'] == ep_len, name
        assert sorted(k for k, v in kwargs.items() if v) == sorted(spec['flags']), name
        for p in r['preprocessors']:
            sp = b._SPECS[m.update_magical_env_name(name, preproc=p)]
            assert (sp['module'], sp['cls'], sp['ep_len'], sp['flags'], sp['preproc']) == (spec['module'], spec['cls'], ep_len, spec['flags'], p)
    for name, want in r['EnvName'].items():
        if isinstance(want, str):
            with pytest.raises((ValueError, AssertionError)):
                m.EnvName(name)
        else:
            e = m.EnvName(name)
            assert {k: getattr(e, k) for k in want} == want, name
    for name, kw, want in r['update_magical_env_name']:
        assert m.update_magical_env_name(name, **kw) == want
    # the oracle's task table: episode lengths
    from oracle.tasks_ref import TASKS
    for epoint, name, ep_len, kwargs in r['specs']:
        assert TASKS[name.split('-')[0]].ep_len == ep_len


def test_longest_line_oracle_and_product():
    from oracle import tasks_ref
    from magical_amd.benchmarks import make_line as ml
    cases = FX['make_line']['longest_line']
    assert ml.INLIER_RAD_MULT == unhex(FX['make_line']['INLIER_RAD_MULT']) and ml.MAX_SEP_RADS == unhex(FX['make_line']['MAX_SEP_RADS'])
    assert (ml.MIN_BLOCKS, ml.MAX_BLOCKS) == (FX['make_line']['MIN_BLOCKS'], FX['make_line']['MAX_BLOCKS'])
    by_shape = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for k, c in enumerate(cases):
            pts = arr(c['points']).reshape(-1, 2)
            a, s = unhex(c['inlier_dist']), unhex(c['max_separation'])
            assert tasks_ref.longest_line(pts, a, s) == c['out'], k
            assert ml.longest_line(pts, a, s) == c['out'], k
            by_shape.setdefault((len(pts), a, s), []).append((pts, c['out']))
        n_batched = 0
        for (n, a, s), lst in by_shape.items():         # the batched form the engine's scoring uses
            if n >= 1:
                got = ml.longest_line_batch(np.stack([p for p, _ in lst]), a, s)
                assert got.tolist() == [o for _, o in lst], (n, a, s)
                n_batched += len(lst)
    assert n_batched > 800


def _ref_env(task):
    from tests.util import new_ref
    return new_ref(task)


def _place(r, ents_xy):
    """Set block positions in an oracle env: {entity: (x, y)}."""
    b = r.bodies()
    for ent, (x, y) in ents_xy.items():
        b[ent.shape_body, 0], b[ent.shape_body, 1] = x, y
    r.set_bodies(b)


def test_scores_oracle_point_tasks():
    """MoveToCorner (+ DebugReward), MakeLine, ClusterColour / ClusterShape: the oracle's restated scores on the fixture's
    block positions equal what the reference's own method bodies returned."""
    r = _ref_env('MoveToCorner')
    for c in FX['move_to_corner']:
        b = r.bodies()
        b[r.task.shape.shape_body, :2] = unhex(c['block'])
        b[r.task.robot.robot_body, :2] = unhex(c['robot'])
        r.set_bodies(b)
        assert float(r.task.score_on_end_of_traj()) == unhex(c['score'])
        assert float(r.task.debug_shaped_reward()) == unhex(c['debug_shaped_reward'])
    r = _ref_env('MakeLine')
    n4 = 0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for c in FX['make_line']['scores']:
            pts = arr(c['points']).reshape(-1, 2)
            if len(pts) != len(r.task.blocks):
                continue
            _place(r, dict(zip(r.task.blocks, pts)))
            assert float(r.task.score_on_end_of_traj()) == unhex(c['score'])
            n4 += 1
    assert n4 > 100
    for task in ('ClusterColour', 'ClusterShape'):
        r = _ref_env(task)
        d = FX['cluster']['demo'][task]
        key = 'colour_name' if task == 'ClusterColour' else 'shape_type'
        assert [str(getattr(e, key)) for e in r.task.shape_ents] == d['labels']
        assert [str(e.colour_name) for e in r.task.shape_ents] == d['block_colours'] and [str(e.shape_type) for e in r.task.shape_ents] == d['block_shapes']
        for e, pose in zip(r.task.shape_ents, unhex(d['block_poses'])):
            assert [e.init_pos[0], e.init_pos[1], e.init_angle] == pose
        for c in d['cases']:
            _place(r, dict(zip(r.task.shape_ents, arr(c['pos']))))
            assert float(r.task.score_on_end_of_traj()) == unhex(c['score'])


def _region_layout(bb, inside, k, n):
    """A position for block k of n: inside the box (near its centre, spread a little) or far outside the arena."""
    l, b, r, t = bb
    if inside:
        return ((l + r) / 2 + 0.02 * (k - n / 2), (b + t) / 2 + 0.015 * ((k * 7) % 5 - 2))
    return (5.0 + k, 5.0)


def test_scores_oracle_region_tasks():
    """MatchRegions / FindDupe / FixColour (Demo worlds): blocks are placed so that the oracle's own overlap query yields the
    fixture's overlap set; the score must then equal the reference's arithmetic on that set."""
    r = _ref_env('MatchRegions')
    T, D = r.task.target_shapes, r.task.distractor_shapes
    bb = r.task.sensor.bb
    n = 0
    for c in FX['match_regions']:
        if (c['n_targets'], c['n_distractors']) != (len(T), len(D)):
            continue
        ents = T + D
        inside = [i < c['targets_in'] for i in range(len(T))] + [i < c['distractors_in'] for i in range(len(D))]
        _place(r, {e: _region_layout(bb, ins, k, len(ents)) for k, (e, ins) in enumerate(zip(ents, inside))})
        assert float(r.task.score_on_end_of_traj()) == unhex(c['score']), c
        n += 1
    assert n == (len(T) + 1) * (len(D) + 1)
    r = _ref_env('FindDupe')
    blocks = [r.task.query_block, *r.task.outside_blocks]
    T = [b_ for b_ in blocks if b_ in r.task.target_set]
    D = [b_ for b_ in blocks if b_ in r.task.distractor_set]
    assert len(T) + len(D) == len(blocks)
    bb = r.task.sensor.bb
    n = 0
    for c in FX['find_dupe']:
        if (c['n_targets'], c['n_distractors']) != (len(T), len(D)):
            continue
        ents = T + D
        inside = [i < c['targets_in'] for i in range(len(T))] + [i < c['distractors_in'] for i in range(len(D))]
        _place(r, {e: _region_layout(bb, ins, k, len(ents)) for k, (e, ins) in enumerate(zip(ents, inside))})
        assert float(r.task.score_on_end_of_traj()) == unhex(c['score']), c
        n += 1
    assert n == (len(T) + 1) * (len(D) + 1)
    r = _ref_env('FixColour')
    sensors, blocks, keep = r.task.sensors, r.task.blocks, [len(t) == 1 for t in r.task.target_blocks]
    n = 0
    for c in FX['fix_colour']:
        if c['keep'] != keep:
            continue
        _place(r, {blk: _region_layout(sensors[w].bb if w >= 0 else (0, 0, 0, 0), w >= 0, k, len(blocks))
                   for k, (blk, w) in enumerate(zip(blocks, c['block_region']))})
        assert float(r.task.score_on_end_of_traj()) == unhex(c['score']), c
        n += 1
    assert n == (len(sensors) + 1) ** len(blocks)


# ====================================================================== the engine (GPU)
def _make(name, n, **kw):
    import magical_amd
    return magical_amd.make(name, n_envs=n, device='cuda:0', **kw)


def _set_xy(env, bodies, ent, xy):
    bodies[:, ent.body, 0], bodies[:, ent.body, 1] = xy[:, 0], xy[:, 1]


def _score(env, bodies, device=True):
    """Host score of the poses written into the device state; where the task's score also exists on the device
    (mgx_engine_score_points: MoveToCorner, MakeLine, Cluster*) the kernel's result must equal it bit for bit -- so every
    point-task case of the fixture pins the device kernel on the reference's own output as well."""
    env.set_bodies(bodies)
    env._scoring_envs = np.arange(env.n_envs)
    host = env.score_on_end_of_traj(env.get_poses())
    if device and env._device_point_scores():
        env._enqueue_point_scores(None)
        dev = env._score_dev.cpu().numpy()
        assert dev.tolist() == np.asarray(host, dtype=np.float64).tolist(), ('device score differs from the host score', int((dev != host).sum()))
    return host


@pytest.mark.gpu
def test_gpu_rand_dynamics_draws_equal_the_reference():
    """MoveToCorner-TestDynamics: env k of a batch seeded s draws from RandomState(s + k) what the reference's
    PhysicsVariables.sample() draws (first and second episode), and the limits reach the device as impulses per substep."""
    pv = FX['phys_vars']
    n, ep = 32, 2
    env = _make('MoveToCorner-TestDynamics-v0', n, max_episode_steps=ep)
    env.seed(0)
    env.reset()
    row = env._info('physvar_row')
    for episode in range(2):
        want = np.array([unhex(pv['samples'][str(k)][episode]) for k in range(n)])
        assert np.array_equal(env.phys_vars, want), episode
        dev = env.state_f[row:row + 5].cpu().numpy().T
        assert np.array_equal(dev, (want * (1.0 / 8 / 10)).astype(np.float32)), episode
        for _ in range(ep):
            env.step(np.zeros(n, dtype=np.int32))
    env.close()


@pytest.mark.gpu
def test_gpu_rendered_colours_equal_the_reference_palette():
    """The allocentric frame of every task shows the reference's colours (style.py through make_reference_vectors.py) where
    they must be: background, block fill at each block's centre, goal interior at each region's centre."""
    import torch
    from magical_amd import entities as en
    from tests.util import TASKS
    st = FX['style']
    u8 = lambda c: [int(v * 255.0 + 0.5) for v in unhex(c)]
    for task in TASKS:
        env = _make(f'{task}-Demo-v0', 1)
        env.reset()
        buf = torch.zeros((1, 96, 96, 3), dtype=torch.uint8, device='cuda:0')
        frame = env.render_frames(buf, view='allo', layout='frame')[0].cpu().numpy()
        poses = env.get_poses()[0]
        # allocentric camera: [-1.02, 1.02]^2 -> 96 px, row 0 at the top (gym_render.py:176-182, style.ARENA_ZOOM_OUT)
        px = lambda x, y: (int((1.02 - y) / 2.04 * 96), int((x + 1.02) / 2.04 * 96))
        assert (frame == 255).all(axis=-1).sum() > 96 * 96 // 4, task     # the arena floor is white (entities.py:531); the clear colour
        # lighten_rgb(grey, 4) only shows in the 1-pixel zoom-out margin, blended with the grey bounds
        robot = env._robot
        for ent in env._entities:
            if isinstance(ent, en.Shape):
                x, y = poses[ent.body, :2]
                if np.hypot(x - poses[robot.body, 0], y - poses[robot.body, 1]) < 0.4:
                    continue
                assert frame[px(x, y)].tolist() == u8(st['COLOURS_RGB'][str(getattr(ent.colour_name, 'value', ent.colour_name))]), (task, ent.ent_id)
        env.close()


@pytest.mark.gpu
def test_gpu_point_task_scores_equal_the_reference():
    """The engine's batched score_on_end_of_traj() for MoveToCorner, MakeLine, ClusterColour, ClusterShape on poses written into
    the device state equals the reference's own method on the same positions, bit for bit; the DebugReward env's shaped reward
    (computed on the device in fp64) agrees to rounding."""
    cases = FX['move_to_corner']
    env = _make('MoveToCorner-Demo-DebugReward-v0', len(cases), dtype='f64')
    env.reset()
    b = env.get_bodies()
    shape = env._MoveToCornerEnv__shape_ref
    _set_xy(env, b, shape, np.array([unhex(c['block']) for c in cases]))
    _set_xy(env, b, env._robot, np.array([unhex(c['robot']) for c in cases]))
    got = _score(env, b)
    assert got.tolist() == [unhex(c['score']) for c in cases]
    rew = env.debug_shaped_reward().cpu().numpy()
    assert np.abs(rew - np.array([unhex(c['debug_shaped_reward']) for c in cases])).max() < 1e-12
    env.close()
    cases = [c for c in FX['make_line']['scores'] if len(c['points']) == 4]
    env = _make('MakeLine-Demo-v0', len(cases))
    env.reset()
    b = env.get_bodies()
    pts = np.array([unhex(c['points']) for c in cases])
    for k, blk in enumerate(env._blocks):
        _set_xy(env, b, blk, pts[:, k])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        got = _score(env, b)
    assert got.tolist() == [unhex(c['score']) for c in cases]
    env.close()
    for task in ('ClusterColour', 'ClusterShape'):
        d = FX['cluster']['demo'][task]
        env = _make(f'{task}-Demo-v0', len(d['cases']))
        env.reset()
        ents = env._BaseClusterEnv__shape_ents
        assert [str(getattr(e.colour_name, 'value', e.colour_name)) for e in ents] == d['block_colours']
        assert [str(getattr(e.shape_type, 'value', e.shape_type)) for e in ents] == d['block_shapes']
        assert np.array_equal(env.default_entity_poses()[[e.ent_id for e in ents]], arr(d['block_poses']))
        b = env.get_bodies()
        pos = np.array([unhex(c['pos']) for c in d['cases']])
        for k, e in enumerate(ents):
            _set_xy(env, b, e, pos[:, k])
        got = _score(env, b)
        assert got.tolist() == [unhex(c['score']) for c in d['cases']], task
        env.close()


@pytest.mark.gpu
def test_gpu_cluster_score_with_random_memberships():
    """cluster.py:166-216 on random label assignments (7..10 blocks, the reference's np.unique class order): the product's
    per-env class tables (the TestAll machinery: counts, colours) reproduce the reference's score on the same positions."""
    from magical_amd import entities as en
    cases = FX['cluster']['random_labels']
    env = _make('ClusterColour-TestAll-v0', len(cases))
    env.seed(3)
    env.reset()
    ents = env._BaseClusterEnv__shape_ents
    ids = [e.ent_id for e in ents]
    values = sorted({l for c in cases for l in c['labels']})                  # np.unique order of the str-valued enum
    assert values == [str(getattr(v, 'value', v)) for v in env._BaseClusterEnv__characteristic_values]
    b = env.get_bodies()
    for k, c in enumerate(cases):
        n = len(c['labels'])
        env.entity_enabled[k, ids] = [i < n for i in range(len(ids))]
        env._class_env[k, :n] = [values.index(l) for l in c['labels']]
        pos = arr(c['pos'])
        for i in range(n):
            b[k, ents[i].body, :2] = pos[i]
    got = _score(env, b, device=False)       # (this test edits the HOST's presence table only: the device has its own, below)
    assert got.tolist() == [unhex(c['score']) for c in cases]
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['MoveToCorner-Demo-v0', 'MoveToCorner-TestAll-v0', 'MakeLine-Demo-v0', 'MakeLine-TestCountPlus-v0', 'MakeLine-TestAll-v0',
                                  'ClusterColour-Demo-v0', 'ClusterColour-TestColour-v0', 'ClusterColour-TestAll-v0', 'ClusterShape-TestCountPlus-v0',
                                  'ClusterShape-TestAll-LoRes4E-v0'])
def test_gpu_device_point_scores_equal_the_host_scores_over_rollouts(name):
    """info['eval_score'] of the point tasks comes off the device whole (mgx_engine_score_points, no pose rows downloaded): over
    rollouts of Demo and Test* variants (per-env worlds: absent blocks, redrawn classes) it equals, bit for bit, the host's
    score_on_end_of_traj() -- the restatement pinned on the reference's method bodies above -- on the same final poses, across
    two auto-resets; a numpy whose BLAS rounds differently would make the env fall back to the host path, which is asserted not
    to be the case here."""
    from magical_amd.benchmarks._scoring import numpy_dot_modes
    assert numpy_dot_modes() is not None
    n, ep = 192, 14
    dev = _make(name, n, max_episode_steps=ep)
    host = _make(name, n, max_episode_steps=ep, device_scores=False)
    assert dev._device_point_scores() and not host._device_point_scores()
    dev.seed(6); host.seed(6); dev.reset(); host.reset()
    rs = np.random.RandomState(8)
    ends = 0
    for s in range(2 * ep):
        a = np.where(rs.rand(n) < 0.7, 1, rs.randint(0, 18, size=n)).astype(np.int32)       # mostly forward: blocks get shoved around
        _, _, dd, di = dev.step(a)
        _, _, dh, hi = host.step(a)
        assert np.array_equal(dd, dh)
        assert di['eval_score'].tolist() == hi['eval_score'].tolist(), (name, s, int((di['eval_score'] != hi['eval_score']).sum()))
        ends += int(dd.any())
    assert ends == 2
    dev.close(); host.close()


@pytest.mark.gpu
def test_gpu_region_task_scores_equal_the_reference():
    """MatchRegions / FindDupe / FixColour: every overlap set of the Demo worlds, produced by really placing the blocks in or
    out of the regions on the device; the product's overlap query + arithmetic gives the reference's score for that set."""
    env = _make('MatchRegions-Demo-v0', 1)
    T, D = env._MatchRegionsEnv__target_shapes, env._MatchRegionsEnv__distractor_shapes
    bb = env._MatchRegionsEnv__sensor_ref.bb
    env.close()
    cases = [c for c in FX['match_regions'] if (c['n_targets'], c['n_distractors']) == (len(T), len(D))]
    env = _make('MatchRegions-Demo-v0', len(cases))
    env.reset()
    T, D = env._MatchRegionsEnv__target_shapes, env._MatchRegionsEnv__distractor_shapes
    b = env.get_bodies()
    for k, c in enumerate(cases):
        ents = T + D
        inside = [i < c['targets_in'] for i in range(len(T))] + [i < c['distractors_in'] for i in range(len(D))]
        for j, (e, ins) in enumerate(zip(ents, inside)):
            b[k, e.body, :2] = _region_layout(bb, ins, j, len(ents))
    assert _score(env, b).tolist() == [unhex(c['score']) for c in cases]
    env.close()

    env = _make('FindDupe-Demo-v0', 1)
    blocks, is_t = env._FindDupeEnv__all_blocks, env._FindDupeEnv__is_target
    nt, nd = int(is_t.sum()), int((~is_t).sum())
    bb = env._FindDupeEnv__sensor_ref.bb
    env.close()
    cases = [c for c in FX['find_dupe'] if (c['n_targets'], c['n_distractors']) == (nt, nd)]
    env = _make('FindDupe-Demo-v0', len(cases))
    env.reset()
    blocks, is_t = env._FindDupeEnv__all_blocks, env._FindDupeEnv__is_target
    T = [e for e, t in zip(blocks, is_t) if t]
    D = [e for e, t in zip(blocks, is_t) if not t]
    b = env.get_bodies()
    for k, c in enumerate(cases):
        ents = T + D
        inside = [i < c['targets_in'] for i in range(len(T))] + [i < c['distractors_in'] for i in range(len(D))]
        for j, (e, ins) in enumerate(zip(ents, inside)):
            b[k, e.body, :2] = _region_layout(bb, ins, j, len(ents))
    assert _score(env, b).tolist() == [unhex(c['score']) for c in cases]
    env.close()

    env = _make('FixColour-Demo-v0', 1)
    keep = [bool(k) for k in env._keep]
    env.close()
    cases = [c for c in FX['fix_colour'] if c['keep'] == keep]
    env = _make('FixColour-Demo-v0', len(cases))
    env.reset()
    b = env.get_bodies()
    for k, c in enumerate(cases):
        for j, (blk, w) in enumerate(zip(env._blocks, c['block_region'])):
            b[k, blk.body, :2] = _region_layout(env._sensors[w].bb if w >= 0 else (0, 0, 0, 0), w >= 0, j, len(env._blocks))
    assert _score(env, b).tolist() == [unhex(c['score']) for c in cases]
    env.close()
