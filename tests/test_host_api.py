"""Host-side (no GPU) checks: env-name registry, action table, C-ABI export list, scoring parity with the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_matches_reference_name_surface():
    import magical_amd as m
    m.register_envs()
    assert m.register_envs() is False                         # idempotent (benchmarks/__init__.py:396-399)
    assert len(m.ALL_REGISTERED_ENVS) == 366                  # 60 x (1 + 5) + 6
    assert len(set(m.ALL_REGISTERED_ENVS)) == 366
    assert 'MoveToCorner-Demo-LoRes4E-v0' in m.ALL_REGISTERED_ENVS
    assert 'ClusterColour-TestAll-LoResCHW4E-v0' in m.ALL_REGISTERED_ENVS
    assert len(m.DEMO_ENVS_TO_TEST_ENVS_MAP) == 48            # 8 tasks x 6 name forms
    assert 'MoveToCorner-TestAll-v0' in m.DEMO_ENVS_TO_TEST_ENVS_MAP['MoveToCorner-Demo-v0']
    e = m.EnvName('MatchRegions-TestJitter-LoRes4A-v0')
    assert (e.task, e.variant, e.preproc, e.version, e.is_test) == ('MatchRegions', 'TestJitter', 'LoRes4A', 'v0', True)
    assert e.demo_env_name == 'MatchRegions-Demo-LoRes4A-v0'
    assert m.update_magical_env_name('FixColour-Demo-v0', preproc='LoRes4E', variant='TestAll') == 'FixColour-TestAll-LoRes4E-v0'
    with pytest.raises(ValueError):
        m.EnvName('NotAnEnv')


@pytest.mark.skipif(not os.path.isdir('/root/reference/magical'), reason='reference only exists in the build container')
def test_registry_names_equal_the_reference_source():
    """Every name string the reference registers appears in ours (parsed from its source text, not imported)."""
    import magical_amd as m
    m.register_envs()
    src = open('/root/reference/magical/benchmarks/__init__.py').read()
    base_names = set(re.findall(r"'([A-Za-z]+-(?:Demo|Test[A-Za-z]*)-v0)'", src))
    assert len(base_names) == 60
    assert base_names <= set(m.ALL_REGISTERED_ENVS)


def test_package_top_level_names_are_the_references(tmp_path):
    """`import magical_amd as magical` offers every name magical/__init__.py:2-8 exports, and the demo-file helpers behave as the
    reference's do (saved_trajectories.py:52-60: the preprocessor name goes before the version; reference_demos.py:27-33: a
    directory that holds the download marker is accepted as it is)."""
    import magical_amd as magical
    ref_names = ['ALL_REGISTERED_ENVS', 'AVAILABLE_PREPROCESSORS', 'DEMO_ENVS_TO_TEST_ENVS_MAP', 'register_envs', 'try_download_demos',
                 'load_demos', 'preprocess_demos_with_wrapper', 'splice_in_preproc_name', '__version__']
    if os.path.isdir('/root/reference/magical'):         # (build container only: the list above is what the reference's source says)
        import ast
        tree = ast.parse(open('/root/reference/magical/__init__.py').read())
        imported = {a.asname or a.name for node in tree.body if isinstance(node, ast.ImportFrom) for a in node.names}
        assert imported == set(ref_names), imported ^ set(ref_names)
    for name in ref_names:
        assert hasattr(magical, name) and name in magical.__all__, name
    assert magical.splice_in_preproc_name('MoveToCorner-Demo-v0', 'LoRes4E') == 'MoveToCorner-Demo-LoRes4E-v0'
    from magical_amd import saved_trajectories as st
    assert magical.load_demos is st.load_demos and magical.preprocess_demos_with_wrapper is st.preprocess_demos_with_wrapper
    from magical_amd.reference_demos import DONE_FILE, DownloadError
    with pytest.raises(DownloadError):
        magical.try_download_demos(str(tmp_path / 'demos'))           # nothing is fetched (no networking in this engine) ...
    (tmp_path / 'have').mkdir(); (tmp_path / 'have' / DONE_FILE).write_text('x')
    assert magical.try_download_demos(str(tmp_path / 'have')) is None      # ... a directory the reference has filled is taken as it is


def test_action_table():
    from magical_amd import entities as en
    A = en.RobotAction
    assert len(en.ACTION_NUMS_FLAGS_NAMES) == 18
    assert en.ACTION_ID_TO_FLAGS[0] == (A.NONE, A.NONE, A.OPEN)
    assert en.ACTION_ID_TO_FLAGS[4] == (A.UP, A.LEFT, A.OPEN) and en.ACTION_NUMS_FLAGS_NAMES[4][2] == 'UpLeftOpen'
    assert en.ACTION_ID_TO_FLAGS[17] == (A.DOWN, A.RIGHT, A.CLOSE) and en.ACTION_NUMS_FLAGS_NAMES[17][2] == 'DownRightClose'
    for i in range(18):
        assert en.FLAGS_TO_ACTION_ID[en.ACTION_ID_TO_FLAGS[i]] == i


def test_c_abi_exports_every_declared_symbol():
    """libmagical_hip.so loads (no GPU needed) and exports every function include/mgx.h declares."""
    from magical_amd import _native
    header = open(os.path.join(ROOT, 'include', 'mgx.h')).read()
    declared = set(re.findall(r'\b(mgx_[a-z_0-9]+)\s*\(', header))
    assert declared == set(_native.EXPORTED_SYMBOLS), declared ^ set(_native.EXPORTED_SYMBOLS)
    # development hooks live in their own header and are not part of the boundary
    debug = set(re.findall(r'\b(mgx_[a-z_0-9]+)\s*\(', open(os.path.join(ROOT, 'include', 'mgx_debug.h')).read()))
    assert debug and all(d.startswith('mgx_engine_debug_') or d.startswith('mgx_debug_') for d in debug) and not (debug & declared)
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('HIP library not built in this checkout (run `python __graft_entry__.py`)')
    L = ctypes.CDLL(_native.LIB_PATH)
    for sym in declared | debug:
        assert hasattr(L, sym), sym


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import magical_amd
    with pytest.raises(Exception) as ei:
        magical_amd.make('MoveToCorner-Demo-v0', n_envs=4, device='cpu')
    assert 'no CPU fallback' in str(ei.value)


def test_world_builder_matches_oracle_tables():
    """The C++ world builder (product) and the Python entity restatement (oracle) agree on masses, inertias,
    initial poses and collision geometry for every task."""
    from magical_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('HIP library not built')
    from tests.util import TASKS, new_ref, ref_body_index, ref_entities_as_tuples
    L = _native.lib()
    for task in TASKS:
        ref = new_ref(task)
        w = ctypes.c_void_p()
        _native.check(L.mgx_world_create(ctypes.byref(w)))
        for ent in ref_entities_as_tuples(ref):
            if ent[0] == 'robot':
                _native.check(L.mgx_world_add_robot(w, *ent[1:]))
            elif ent[0] == 'shape':
                _native.check(L.mgx_world_add_shape(w, *ent[1:]))
            else:
                _native.check(L.mgx_world_add_goal(w, *ent[1:]))
        _native.check(L.mgx_world_finalize(w, 100))
        out = ctypes.c_int()
        _native.check(L.mgx_world_info(w, _native.INFO['n_bodies'], ctypes.byref(out)))
        nb = out.value
        idx = ref_body_index(ref)
        assert nb == len(idx) + 1
        mass = (ctypes.c_double * (2 * nb))()
        pose = (ctypes.c_double * (3 * nb))()
        _native.check(L.mgx_world_body_table(w, mass, pose))
        rm, rb = ref.body_mass()[idx], ref.bodies()[idx]
        assert np.allclose(np.array(mass).reshape(nb, 2)[1:], rm, rtol=1e-14, atol=0)
        got_pose = np.array(pose).reshape(nb, 3)[1:]
        mask = np.ones_like(got_pose, dtype=bool)
        assert np.allclose(got_pose[mask], rb[:, :3][mask], rtol=0, atol=1e-15)
        _native.check(L.mgx_world_info(w, _native.INFO['n_joints'], ctypes.byref(out)))
        assert out.value == ref.L.ref_njoints(ref.h)
        L.mgx_world_destroy(w)


def test_world_variants_keep_indices_and_match_oracle_worlds():
    """mgx_world_variant (Test*Shape / CountPlus worlds): absent blocks keep their entity index and an inert body with its
    state rows, so body indices and the host-addressed state rows are those of the full world; the present entities'
    masses / inertias / joint count equal the oracle's world built with only those entities."""
    from magical_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('HIP library not built')
    from oracle.env_ref import RefEnv
    from tests.util import ref_body_index
    L = _native.lib()
    info = lambda w, key: (lambda out: (_native.check(L.mgx_world_info(w, _native.INFO[key], ctypes.byref(out))), out.value)[1])(ctypes.c_int())
    flags = dict(rand_shape_colour=True, rand_shape_type=True, rand_shape_count=True, rand_layout_full=True)
    st_id = {'triangle': 0, 'square': 1, 'pentagon': 2, 'hexagon': 3, 'octagon': 4, 'circle': 5, 'star': 6}
    w = ctypes.c_void_p()
    _native.check(L.mgx_world_create(ctypes.byref(w)))
    for k in range(10):                                      # the ten block slots of Cluster*-TestCountPlus, then the robot
        _native.check(L.mgx_world_add_shape(w, 1, 0, 0.0, 0.0, 0.0))
    _native.check(L.mgx_world_add_robot(w, 0.286, -0.202, -1.878))
    _native.check(L.mgx_world_finalize(w, 100))
    full = {k: info(w, k) for k in ('n_bodies', 'state_rows_p', 'physvar_row', 'n_shapes', 'n_joints')}
    seen_counts = set()
    for seed in range(6):
        ref = RefEnv('ClusterShape', seed=seed, **flags)
        ref.reset()
        slots = ref.task.slots
        enabled = np.array([s is not None for s in slots], dtype=np.uint8)
        types = np.array([st_id[str(s.shape_type)] if s is not None and hasattr(s, 'shape_type') else -1 for s in slots], dtype=np.int32)
        seen_counts.add(int(enabled.sum()))
        v = ctypes.c_void_p()
        _native.check(L.mgx_world_variant(w, enabled.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), types.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(v)))
        assert {k: info(v, k) for k in ('n_bodies', 'state_rows_p', 'physvar_row')} == {k: full[k] for k in ('n_bodies', 'state_rows_p', 'physvar_row')}
        assert info(v, 'n_joints') == ref.L.ref_njoints(ref.h)
        nb = full['n_bodies']
        mass = (ctypes.c_double * (2 * nb))()
        _native.check(L.mgx_world_body_table(v, mass, None))
        mass = np.array(mass).reshape(nb, 2)
        body = ctypes.c_int()
        want = ref.body_mass()
        for e, s in enumerate(slots):
            _native.check(L.mgx_world_entity(v, e, None, ctypes.byref(body), None, None))
            if s is None:
                assert np.all(mass[body.value] == 0.0)                       # inert
            else:
                assert np.allclose(mass[body.value], want[s.bodies[0]], rtol=1e-14, atol=0), (seed, e)
        L.mgx_world_destroy(v)
    assert len(seen_counts) > 1
    # the robot cannot be absent; unknown shape types are rejected
    bad = np.ones(11, dtype=np.uint8); bad[10] = 0
    v = ctypes.c_void_p()
    assert L.mgx_world_variant(w, bad.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), None, ctypes.byref(v)) < 0
    L.mgx_world_destroy(w)


def test_palette_table_matches_colorsys():
    """The RGB8 palette baked into mgx_world.cpp equals style.py evaluated with colorsys (oracle/style_ref.py)."""
    from oracle.style_ref import COLOURS_RGB, darken_rgb, lighten_rgb, to_u8
    src = open(os.path.join(ROOT, 'magical_amd', 'csrc', 'mgx_world.cpp')).read()

    def table(name):
        body = re.search(name + r'\[4\] = \{(.*?)\};', src).group(1)
        return [tuple(int(v) for v in t.split(',')) for t in re.findall(r'\{(\d+, \d+, \d+)\}', body)]
    order = ['red', 'green', 'blue', 'yellow']
    assert table('BASE') == [to_u8(COLOURS_RGB[c]) for c in order]
    assert table('DARK') == [to_u8(darken_rgb(COLOURS_RGB[c])) for c in order]
    assert table('LIGHT2') == [to_u8(lighten_rgb(COLOURS_RGB[c], 2)) for c in order]


def test_spaces_standins():
    """The gym.spaces stand-ins carry what the reference's users read from them (base_env.py:97-109)."""
    from magical_amd import spaces
    a = spaces.Discrete(18)
    assert a.n == 18 and a.contains(17) and not a.contains(18) and 0 <= a.sample(np.random.RandomState(0)) < 18
    b = spaces.Box(0, 255, (96, 96, 12), 'uint8')
    assert b.shape == (96, 96, 12) and b.dtype == np.uint8 and b.low.min() == 0 and b.high.max() == 255
    assert b.contains(np.zeros((96, 96, 12), dtype=np.uint8)) and not b.contains(np.zeros((96, 96, 3), dtype=np.uint8))
    d = spaces.Dict([('allo', b), ('ego', b)])
    assert list(d.spaces) == ['allo', 'ego'] and d['ego'] == b
    assert d.contains(d.sample(np.random.RandomState(1)))
    assert spaces.to_gym(a) is a or type(spaces.to_gym(a)).__name__ == 'Discrete'


def test_native_pose_sampler_matches_python_and_numpy_stream():
    """mgx_world_randomise_all_poses draws from the RandomState's MT19937 state exactly what the Python mirror of
    geom.py:116-341 draws through rng.uniform: same poses, and the stream ends in the same place (no GPU needed: the
    world builder is host code)."""
    from magical_amd import _native, geom
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('HIP library not built')
    L = _native.lib()

    class Ent:
        def __init__(self, i):
            self.ent_id = i

    class Shim:          # what geom.py needs of an env
        _lib = L
    w = ctypes.c_void_p()
    _native.check(L.mgx_world_create(ctypes.byref(w)))
    Shim._world = w
    layout = [('robot', 0.7, -0.3, 0.8)] + [('shape', t, c, x, y, a) for (t, c, x, y, a) in [
        (5, 2, -0.51, 0.14, -0.39), (6, 2, -0.13, -0.71, 1.05), (1, 2, -0.74, -0.1, 1.16), (2, 1, -0.08, -0.43, -0.64),
        (2, 1, 0.52, 0.19, -1.18), (1, 0, -0.53, -0.22, 2.94), (6, 3, -0.54, 0.48, 0.07), (2, 3, -0.17, 0.64, -2.33)]]
    poses0 = []
    for ent in layout:
        if ent[0] == 'robot':
            _native.check(L.mgx_world_add_robot(w, ent[1], ent[2], ent[3])); poses0.append(ent[1:4])
        else:
            _native.check(L.mgx_world_add_shape(w, ent[1], ent[2], ent[3], ent[4], ent[5])); poses0.append(ent[3:6])
    _native.check(L.mgx_world_finalize(w, 100))
    ents = [Ent(i) for i in range(len(layout))]
    for seed, limits in [(0, (None, None)), (1, (0.025, 0.05 * np.pi)), (2, (None, None)), (3, (0.3, None))]:
        r1, r2 = np.random.RandomState(seed), np.random.RandomState(seed)
        r1.uniform(); r2.uniform()        # start mid-stream
        p1 = geom.pm_randomise_all_poses(Shim, np.array(poses0, dtype=np.float64), ents, [-1, 1, -1, 1], r1,
                                         rel_pos_linf_limits=limits[0], rel_rot_limits=limits[1], native=True)
        p2 = geom.pm_randomise_all_poses(Shim, np.array(poses0, dtype=np.float64), ents, [-1, 1, -1, 1], r2,
                                         rel_pos_linf_limits=limits[0], rel_rot_limits=limits[1], native=False)
        assert np.array_equal(p1, p2), (seed, p1 - p2)
        assert r1.randint(1 << 30) == r2.randint(1 << 30)        # both streams advanced identically
        if limits[0] is None:
            assert np.abs(p1 - np.array(poses0)).max() > 0.2     # a full layout really moves things
        # the batch entry point working on the live RandomState states: same poses, same stream positions
        r3 = [np.random.RandomState(seed), np.random.RandomState(seed + 100)]
        for r in r3:
            r.uniform()
        pb = np.ascontiguousarray(np.tile(np.array(poses0, dtype=np.float64), (2, 1, 1)))
        geom.pm_randomise_all_poses_batch(Shim, pb, ents, [-1, 1, -1, 1], r3, rel_pos_linf_limits=limits[0], rel_rot_limits=limits[1])
        assert np.array_equal(pb[0], p1) and not np.array_equal(pb[1], p1)
        r4 = np.random.RandomState(seed); r4.uniform()
        geom.pm_randomise_all_poses(Shim, np.array(poses0, dtype=np.float64), ents, [-1, 1, -1, 1], r4,
                                    rel_pos_linf_limits=limits[0], rel_rot_limits=limits[1], native=False)
        assert r3[0].randint(1 << 30) == r4.randint(1 << 30)
        # nothing overlaps, nothing pokes through the walls
        en_all = np.ones(len(layout), dtype=np.uint8)
        for e in ents:
            assert not geom.placement_collides(Shim, e.ent_id, p1, en_all)
    L.mgx_world_destroy(w)


def _placement_world(L, _native):
    class Ent:
        def __init__(self, i):
            self.ent_id = i

    class Shim:          # what geom.py needs of an env
        _lib = L
    w = ctypes.c_void_p()
    _native.check(L.mgx_world_create(ctypes.byref(w)))
    Shim._world = w
    poses0 = []
    for (x, y, h, wd, c) in [(-0.9, 0.9, 0.5, 0.6, 0), (0.2, 0.8, 0.45, 0.5, 2)]:
        _native.check(L.mgx_world_add_goal(w, x, y, h, wd, c)); poses0.append((x + wd / 2, y - h / 2, 0.0))
    _native.check(L.mgx_world_add_robot(w, 0.6, -0.4, 0.8)); poses0.append((0.6, -0.4, 0.8))
    for (t, c, x, y, a) in [(5, 2, -0.51, 0.14, -0.39), (6, 2, -0.13, -0.71, 1.05), (1, 2, -0.74, -0.1, 1.16), (2, 1, -0.08, -0.43, -0.64),
                            (4, 1, 0.52, 0.19, -1.18), (0, 0, -0.53, -0.62, 2.94), (6, 3, 0.1, 0.3, 0.07)]:
        _native.check(L.mgx_world_add_shape(w, t, c, x, y, a)); poses0.append((x, y, a))
    _native.check(L.mgx_world_finalize(w, 100))
    return Shim, w, [Ent(i) for i in range(len(poses0))], np.array(poses0, dtype=np.float64)


def test_native_pose_sampler_batch_on_the_host_pool_equals_the_calls_one_by_one():
    """mgx_world_randomise_all_poses_batch spreads its envs over the persistent host pool (mgx_api.hip HostPool) and answers its
    collision queries from world-space shapes it holds per entity (only the entity being placed is rebuilt per attempt); the
    single call and the Python loop over mgx_world_placement_collides (geom.py:116-341 restated) rebuild everything per query.
    Goal regions with per-env sizes, an ignored entity, per-env limits: same poses, same stream positions -- twice, so that
    the second burst runs on pool threads that already hold buffers from the first."""
    from magical_amd import _native, geom
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('HIP library not built')
    L = _native.lib()
    Shim, w, ents, poses0 = _placement_world(L, _native)
    m, ne = 200, len(ents)
    rs = np.random.RandomState(7)
    for rep in range(2):
        ent_hw = np.zeros((m, ne, 2)); ent_hw[:, :2] = 0.3 + 0.3 * rs.rand(m, 2, 2)
        order = ents[:2] + ents[2:]                     # regions first, then the robot, then the blocks (find_dupe.py:84-112's order)
        pl = np.where(rs.rand(m, ne) < 0.4, 0.2 + 0.3 * rs.rand(m, ne), np.nan)
        rl = np.where(rs.rand(m, ne) < 0.4, 0.1 + rs.rand(m, ne), np.nan)
        rngs = [np.random.RandomState(1000 * rep + k) for k in range(m)]
        twins = [np.random.RandomState(1000 * rep + k) for k in range(m)]
        batch = np.ascontiguousarray(np.tile(poses0, (m, 1, 1)))
        geom.pm_randomise_all_poses_batch(Shim, batch, order, [-1, 1, -1, 1], rngs, rel_pos_linf_limits=pl, rel_rot_limits=rl,
                                          ignore=[ents[-1]], ent_hw=ent_hw)
        en_all = np.ones(ne, dtype=np.uint8); en_all[-1] = 0
        for k in range(m):
            lim = lambda row: [None if np.isnan(x) else float(x) for x in row]
            one = geom.pm_randomise_all_poses(Shim, poses0.copy(), order, [-1, 1, -1, 1], twins[k], rel_pos_linf_limits=lim(pl[k]),
                                              rel_rot_limits=lim(rl[k]), ignore=[ents[-1]], ent_hw=ent_hw[k], native=(k % 8 != 0))
            assert np.array_equal(batch[k], one), (rep, k)
            assert rngs[k].randint(1 << 30) == twins[k].randint(1 << 30)
            for e in ents[:-1]:
                assert not geom.placement_collides(Shim, e.ent_id, batch[k], en_all, ent_hw=ent_hw[k])
    L.mgx_world_destroy(w)


def _placement_batch_in_child(q):
    from magical_amd import _native, geom
    L = _native.lib()
    Shim, w, ents, poses0 = _placement_world(L, _native)
    rngs = [np.random.RandomState(k) for k in range(128)]
    batch = np.ascontiguousarray(np.tile(poses0, (128, 1, 1)))
    geom.pm_randomise_all_poses_batch(Shim, batch, ents, [-1, 1, -1, 1], rngs)
    q.put(float(batch.sum()))


def test_host_pool_survives_a_fork():
    """The pool's threads do not exist in a forked child: it has to start its own instead of waiting for the parent's."""
    import multiprocessing as mp
    from magical_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('HIP library not built')
    ctx = mp.get_context('fork')
    q0 = ctx.Queue()
    _placement_batch_in_child(q0)                 # the parent's pool exists from here on
    want = q0.get(timeout=10)
    q = ctx.Queue()
    p = ctx.Process(target=_placement_batch_in_child, args=(q,))
    p.start()
    try:
        got = q.get(timeout=60)
    finally:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
    assert got == want


def test_make_line_batched_score_is_bit_identical():
    """longest_line_batch == longest_line (the per-env mirror of make_line.py:31-71) on random layouts, near-collinear
    layouts around the inlier threshold, and degenerate ones (coincident points)."""
    from magical_amd.benchmarks.make_line import longest_line, longest_line_batch
    rs = np.random.RandomState(0)
    inlier, sep = 0.12 * 1.5, 0.12 * 3.5
    pts = [rs.uniform(-1, 1, size=(4000, 4, 2))]
    base = rs.uniform(-0.8, 0.8, size=(4000, 1, 2)); d = rs.uniform(-1, 1, size=(4000, 1, 2)); d /= np.linalg.norm(d, axis=2, keepdims=True)
    t = np.sort(rs.uniform(0, 1.2, size=(4000, 4, 1)), axis=1)
    near = base + t * d + rs.normal(0, 0.12, size=(4000, 4, 2)) * rs.choice([0.0, 0.5, 1.0, 1.5], size=(4000, 1, 1))
    pts.append(near)
    deg = rs.uniform(-1, 1, size=(200, 4, 2)); deg[:, 1] = deg[:, 0]; deg[:100, 3] = deg[:100, 2]
    pts.append(deg)
    for p in pts:
        got = longest_line_batch(np.ascontiguousarray(p), inlier, sep)
        with np.errstate(divide='ignore', invalid='ignore'):
            want = np.array([longest_line(np.ascontiguousarray(q), inlier, sep) for q in p])
        assert np.array_equal(got, want), np.nonzero(got != want)[0][:5]
        assert len(np.unique(want)) >= 2


def test_cheap_draws_advance_the_stream_like_the_reference_calls():
    """The per-env draws replace rng.choice / rng.uniform by their cheaper constituents (entities.draw_choice,
    PhysicsVariables.sample, geom.randomise_hw): same values, same stream position, for scalar and sized draws."""
    from magical_amd import entities as en, geom
    from magical_amd.base_env import PhysicsVariables
    names = en.SHAPE_COLOUR_NAMES
    for seed in range(50):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        want = [a.choice(np.asarray(names, dtype='object')) for _ in range(3)] + a.choice(names, size=5).tolist() + [a.choice(names) for _ in range(4)]
        got = [en.draw_choice(b, names) for _ in range(3)] + en.draw_choice(b, names, size=5) + en.draw_choice(b, names, size=4)
        assert [str(x) for x in want] == [str(x) for x in got]
        want_pv = [float(a.uniform(*getattr(PhysicsVariables, n)[1])) for n in PhysicsVariables.NAMES]
        assert PhysicsVariables.sample(b) == want_pv
        for kw in (dict(), dict(current_hw=(0.7, 0.6), linf_bound=0.0075)):
            minima, maxima = np.asarray((0.5, 0.5)), np.asarray((0.8, 0.8))
            if kw:
                minima = np.maximum(minima, np.asarray(kw['current_hw']) - kw['linf_bound']); maxima = np.minimum(maxima, np.asarray(kw['current_hw']) + kw['linf_bound'])
            h, w = a.uniform(minima, maxima)                       # geom.py:344-360
            assert geom.randomise_hw(0.5, 0.8, b, **kw) == (h, w)
        assert a.random_sample() == b.random_sample()


def test_batched_rng_primitives_equal_numpy():
    """mgx_rng_bounded / doubles / shuffle_batch (batch_rng.BatchRng) draw from live np.random.RandomState streams exactly what
    rng.randint / rng.choice / rng.random_sample / rng.shuffle draw, and leave the streams where numpy leaves them."""
    from magical_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('HIP library not built')
    from magical_amd.batch_rng import BatchRng
    m = 150
    rngs = [np.random.RandomState(100 + k) for k in range(m)]
    refs = [np.random.RandomState(100 + k) for k in range(m)]
    brng = BatchRng(rngs)
    for rnd in range(25):
        rs = np.random.RandomState(rnd)
        n = int(rs.randint(1, 40))
        counts = rs.randint(0, 7, size=m)
        out = brng.randint(n, counts=counts)
        for k, r in enumerate(refs):
            want = r.randint(0, n, size=counts[k]) if counts[k] else []
            assert list(out[k, :counts[k]]) == list(want)
        d = brng.random_sample(3)
        for k, r in enumerate(refs):
            assert np.array_equal(d[k], r.random_sample(3))
        items = rs.randint(0, 12, size=m)
        perm = brng.shuffle(items)
        for k, r in enumerate(refs):
            lst = list(range(100, 100 + items[k]))
            r.shuffle(lst)
            assert lst == [100 + p for p in perm[k, :items[k]]]
        one = brng.randint(4)[:, 0]                      # rng.choice(seq of 4) indexes with this
        names = np.asarray(['a', 'b', 'c', 'd'], dtype='object')
        for k, r in enumerate(refs):
            assert names[one[k]] == r.choice(names)
        rows = np.nonzero(rs.rand(m) < 0.4)[0]           # a subset of the envs draws (per-env bounds)
        sub = brng.randint(3, rows=rows)[:, 0]
        for v, k in zip(sub, rows):
            assert v == refs[k].randint(3)
    assert all(a.randint(1 << 30) == b.randint(1 << 30) for a, b in zip(rngs, refs))
    # PhysicsVariables.sample for a batch
    from magical_amd.base_env import PhysicsVariables
    got = PhysicsVariables.sample_batch(brng)
    assert np.array_equal(got, np.array([PhysicsVariables.sample(r) for r in refs]))


def test_batched_rng_primitives_on_the_host_pool_equal_numpy():
    """The same primitives at a batch size that is spread over the host pool (>= 1024 streams per call, mgx_api.hip rng_batch):
    every stream still draws what numpy draws, and an error in one env is reported."""
    from magical_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('HIP library not built')
    from magical_amd.batch_rng import BatchRng
    m = 2500
    rngs = [np.random.RandomState(5000 + k) for k in range(m)]
    refs = [np.random.RandomState(5000 + k) for k in range(m)]
    brng = BatchRng(rngs)
    rs = np.random.RandomState(0)
    for rnd in range(3):
        counts = rs.randint(0, 5, size=m)
        out = brng.randint(7, counts=counts)
        d = brng.random_sample(2)
        items = rs.randint(0, 9, size=m)
        perm = brng.shuffle(items)
        for k, r in enumerate(refs):
            assert list(out[k, :counts[k]]) == list(r.randint(0, 7, size=counts[k]) if counts[k] else [])
            assert np.array_equal(d[k], r.random_sample(2))
            lst = list(range(items[k])); r.shuffle(lst)
            assert lst == list(perm[k, :items[k]])
    assert all(a.randint(1 << 30) == b.randint(1 << 30) for a, b in zip(rngs, refs))
    with pytest.raises(AssertionError):
        BatchRng(rngs + rngs[:1])                        # the same stream twice in one batch
    L = _native.lib()
    bad = brng.addrs.copy(); bad[m // 2] = 0
    out = np.zeros((m, 1), dtype=np.int32)
    rc = L.mgx_rng_bounded_batch(m, bad.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), None, 1, 3, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), 1)
    assert rc < 0 and b'MT19937' in L.mgx_last_error()


def test_bench_window_plan_gives_every_window_its_share_of_episode_ends():
    """bench.py: a K-step timed window shorter than an episode contains the episode end of n * K / ep envs (their clocks set
    ahead so that the end falls in the middle of the window) and no end of the others, neither in the window nor in the
    warm-up; windows that cannot be placed inside one episode, or that span episodes, are left alone."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    for ep in (40, 80, 240):
        for K in (1, 5, 20, 39, 79, 80, 400):
            for W in (0, 5, 20, 60):
                n = 4096
                preroll, share, ahead = bench.window_plan(K, W, ep, n)
                if K >= ep or W + K >= ep:
                    assert (preroll, share, ahead) == (0, 0, 0)
                    continue
                assert share == round(n * K / ep) and preroll >= ep
                # clocks are set when every env starts its second episode (preroll step ep); the window starts preroll - ep + W steps later
                start = preroll - ep + W
                first, last = start, start + K - 1                       # steps of the window, counted from the setting of the clocks
                assert start + K < ep and start >= W                      # the untouched envs' clock (0 then) stays below ep to the window's end
                end_step = ep - 1 - ahead                                 # step (since the clocks were set) at which the others finish
                assert first <= end_step <= last                          # inside the window
                assert end_step >= start                                  # not during the warm-up


def test_numpy_dot_modes_probe_is_self_consistent():
    """_scoring.numpy_dot_modes(): which rounding this numpy's ddot / small matmul take (handed to mgx_engine_score_points).  The mode it
    reports must reproduce the live primitives on rows it has not seen, and the three candidates must really differ somewhere."""
    from fractions import Fraction
    from magical_amd.benchmarks._scoring import numpy_dot_modes
    modes = numpy_dot_modes()
    assert modes is not None, 'this numpy rounds a 2-element dot product in a way the device kernel does not know: host scoring would be used'
    dot_mode, mm_mode = modes
    rs = np.random.RandomState(77)
    a, b = rs.uniform(-3, 3, size=(600, 2)), rs.uniform(-3, 3, size=(600, 2))

    def cand(mode, ax, ay, bx, by):
        if mode == 0:
            return ax * bx + ay * by
        if mode == 1:
            return float(Fraction(ax * bx) + Fraction(ay) * Fraction(by))
        return float(Fraction(ax) * Fraction(bx) + Fraction(ay * by))
    assert [cand(dot_mode, r[0], r[1], r[0], r[1]) for r in a] == [float(np.dot(r, r)) for r in a]
    for n in (3, 4):        # the [n, 2] @ [2, 1] products of make_line.py:47 (a [1, 2] @ [2, 1] product takes another kernel)
        rows = (600 // n) * n
        got = np.concatenate([np.squeeze(a[i:i + n] @ b[i // n][:, None], axis=1) for i in range(0, rows, n)])
        assert [cand(mm_mode, a[i, 0], a[i, 1], b[i // n, 0], b[i // n, 1]) for i in range(rows)] == got.tolist(), n
    differ = sum(len({cand(m, x[0], x[1], y[0], y[1]) for m in range(3)}) > 1 for x, y in zip(a, b))
    assert differ > 10


def test_kernels_never_ask_where_a_wavefront_sits():
    """Wavefronts are context-switched on the target system and come back in another slot (round 5: an LDS table indexed by HW_ID's
    SIMD and slot bits hung one env-step in ~20 000, tools/dev/hang_hunt.py).  Product code must not read HW_ID / XCC_ID: every
    s_getreg in csrc/ has to sit under a development-build guard."""
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'magical_amd', 'csrc')
    dev_guards = ('MGX_RASTER_CLOCKS', 'MGX_RASTER_PROBE', 'MGX_STEP_PROBE', 'MGX_HANG_DEBUG')
    offenders = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith(('.hip', '.h', '.inc', '.cpp')):
            continue
        stack = []          # one entry per open #if: True when that block is a development-only one
        for no, line in enumerate(open(os.path.join(csrc, name), encoding='utf-8'), 1):
            s = line.strip()
            if re.match(r'#\s*if', s):
                stack.append(any(g in s for g in dev_guards) and not s.startswith('#ifndef'))
            elif re.match(r'#\s*elif', s) and stack:
                stack[-1] = any(g in s for g in dev_guards)
            elif re.match(r'#\s*else', s) and stack:
                stack[-1] = False
            elif re.match(r'#\s*endif', s) and stack:
                stack.pop()
            elif 's_getreg' in s and not s.startswith('//') and not any(stack):
                offenders.append('%s:%d' % (name, no))
    assert not offenders, offenders
