"""Generates tests/golden/reference_vectors.json by RUNNING the reference's own pure-Python code.

Runs only where /root/reference exists (the build container).  pymunk, gym, pyglet and cv2 are absent, so the
reference package cannot be imported as a whole; but a good part of the hot path's host arithmetic is plain
Python / numpy with no third-party dependency, and that part is executed here, unmodified:

  * `style.py` and `phys_vars.py` import stand-alone (stdlib only) and are loaded by path;
  * single functions / classes / constants of modules whose *imports* need pymunk or gym are cut out of the module's
    syntax tree at run time (`ast`), compiled and executed against the real numpy / math / re / itertools -- no
    stand-in is written for anything: a definition that touches pymunk (Vec2d, Space, shape queries) is simply not
    taken.  Methods are re-wrapped in a class of the same name so that `self.__x` mangles as in the reference.
  * `score_on_end_of_traj()` bodies are run on synthetic `self` objects that carry nothing but input data (block
    positions, overlap sets): what is pinned is the reference's scoring ARITHMETIC, not its pymunk queries.

The fixture holds inputs and outputs only (doubles as C99 hex strings, so they round-trip exactly); no reference
source text is copied.  CPU tests check the oracle against it, GPU tests check the product against it; only the
JSON travels to the GPU box.
"""
import ast
import collections
import enum
import importlib.util
import itertools
import json
import math
import os
import re
import types
import warnings

import numpy as np

REF = '/root/reference/magical'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_vectors.json')


def hx(v):
    """float (or nested sequence of floats) -> hex strings."""
    if isinstance(v, (list, tuple, np.ndarray)):
        return [hx(x) for x in v]
    return float(v).hex()


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def tree_of(rel):
    with open(os.path.join(REF, rel)) as f:
        return ast.parse(f.read(), filename=rel)


def cut(rel, names, namespace):
    """Execute the top-level definitions `names` (functions, classes, assignments) of a reference module in `namespace`."""
    found = set()
    body = []
    for node in tree_of(rel).body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            body.append(node); found.add(node.name)
        elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in node.targets):
            body.append(node); found.update(t.id for t in node.targets if isinstance(t, ast.Name))
    missing = set(names) - found
    assert not missing, (rel, missing)
    exec(compile(ast.Module(body=body, type_ignores=[]), os.path.join(REF, rel), 'exec'), namespace)
    return namespace


def cut_methods(rel, cls_name, methods, namespace, attrs=()):
    """Class `cls_name` reduced to the given methods (and class-level assignments `attrs`), so that private names mangle as in
    the reference; bases are dropped."""
    for node in tree_of(rel).body:
        if isinstance(node, ast.ClassDef) and node.name == cls_name:
            keep = [n for n in node.body if (isinstance(n, ast.FunctionDef) and n.name in methods) or
                    (isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in attrs for t in n.targets))]
            assert {n.name for n in keep if isinstance(n, ast.FunctionDef)} == set(methods), (rel, cls_name)
            new = ast.ClassDef(name=cls_name, bases=[], keywords=[], body=keep, decorator_list=[])
            mod = ast.fix_missing_locations(ast.Module(body=[new], type_ignores=[]))
            exec(compile(mod, os.path.join(REF, rel), 'exec'), namespace)
            return namespace[cls_name]
    raise KeyError(cls_name)


def cut_function_locals(rel, func, wanted, namespace):
    """Run the plain data-building statements of `func` (assignments and .extend() calls whose expressions use nothing but
    literals and names already built) and return the locals in `wanted`: register_envs() builds its table of env
    specifications that way before it hands them to gym."""
    for node in tree_of(rel).body:
        if isinstance(node, ast.FunctionDef) and node.name == func:
            ns = dict(namespace)
            for st in node.body:
                ok = isinstance(st, ast.Assign) or (isinstance(st, ast.Expr) and isinstance(st.value, ast.Call) and
                                                   isinstance(st.value.func, ast.Attribute) and st.value.func.attr == 'extend')
                if not ok:
                    continue
                names = {n.id for n in ast.walk(st) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
                if not names <= set(ns) | {'dict'}:
                    continue
                exec(compile(ast.fix_missing_locations(ast.Module(body=[st], type_ignores=[])), rel, 'exec'), ns)
            return {k: ns[k] for k in wanted}
    raise KeyError(func)


class Obj:
    """Plain attribute bag: carries input data into a reference method."""
    def __init__(self, **kw):
        self.__dict__.update(kw)


def block_at(x, y):
    return Obj(shape_body=Obj(position=Obj(x=float(x), y=float(y))))


def main():
    out = collections.OrderedDict()
    out['_about'] = ('outputs of qxcv/magical code run in the build container by tests/golden/make_reference_vectors.py; '
                     'doubles are C99 hex strings')

    # ------------------------------------------------------------------ style.py (whole module)
    style = load_by_path('ref_style', 'style.py')
    names = list(style.COLOURS_RGB)
    out['style'] = {
        'colour_names': names,
        'COLOURS_RGB': {n: hx(style.COLOURS_RGB[n]) for n in names},
        'darken_rgb': {n: hx(style.darken_rgb(style.COLOURS_RGB[n])) for n in names},
        'lighten_rgb': {str(t): {n: hx(style.lighten_rgb(style.COLOURS_RGB[n], times=t)) for n in names} for t in (1, 2, 4)},
        'GOAL_LINE_THICKNESS': hx(style.GOAL_LINE_THICKNESS), 'SHAPE_LINE_THICKNESS': hx(style.SHAPE_LINE_THICKNESS),
        'ROBOT_LINE_THICKNESS': hx(style.ROBOT_LINE_THICKNESS), 'ARENA_ZOOM_OUT': hx(style.ARENA_ZOOM_OUT),
    }

    # ------------------------------------------------------------------ phys_vars.py (whole module) + base_env.PhysicsVariables
    pv = load_by_path('ref_phys_vars', 'phys_vars.py')
    ns = cut('base_env.py', ['PhysicsVariables'], {'PhysicsVariablesBase': pv.PhysicsVariablesBase, 'PhysVar': pv.PhysVar})
    PV = ns['PhysicsVariables']
    var_names = list(PV.variables)
    d = PV.defaults()
    seeds = list(range(32)) + [1234, 1235, 1236, 4242, 2 ** 31 - 1, 2 ** 32 - 1]
    samples = {}
    for s in seeds:
        # two consecutive draws from one stream: an env's first and second episode (base_env.py:198-203)
        rng = np.random.RandomState(s)
        samples[str(s)] = []
        for _ in range(2):
            inst = PV.sample(rng)
            samples[str(s)].append(hx([getattr(inst, n) for n in var_names]))
    out['phys_vars'] = {'names': var_names, 'defaults': hx([getattr(d, n) for n in var_names]),
                        'bounds': {n: hx([PV.variables[n].lower, PV.variables[n].upper]) for n in var_names},
                        'samples': samples}

    # ------------------------------------------------------------------ geom.py: the pymunk-free functions
    g = cut('geom.py', ['regular_poly_circumrad', 'regular_poly_circ_rad_to_side_length', 'regular_poly_apothem_to_side_legnth',
                        'regular_poly_side_length_to_apothem', 'randomise_hw'], {'math': math, 'np': np})
    lens = [0.12, 0.2, 0.12 * 2 ** 0.5, 1.0, 0.0371]
    out['geom'] = {
        'lengths': hx(lens),
        'n_sides': list(range(3, 11)),
        'regular_poly_circumrad': [[hx(g['regular_poly_circumrad'](n, l)) for l in lens] for n in range(3, 11)],
        'regular_poly_circ_rad_to_side_length': [[hx(g['regular_poly_circ_rad_to_side_length'](n, l)) for l in lens] for n in range(3, 11)],
        'regular_poly_apothem_to_side_legnth': [[hx(g['regular_poly_apothem_to_side_legnth'](n, l)) for l in lens] for n in range(3, 11)],
        'regular_poly_side_length_to_apothem': [[hx(g['regular_poly_side_length_to_apothem'](n, l)) for l in lens] for n in range(3, 11)],
    }
    hw_cases = []
    # (min_side, max_side, current_hw, linf_bound): the goal-region draws of MoveToRegion / MatchRegions / FindDupe
    # (0.5..0.8) and FixColour (0.4..0.5), full and jittered (JITTER_TARGET_BOUND = 0.05 * 0.3 / 2, base_env.py:68-76)
    jt = 0.05 * (0.8 - 0.5) / 2
    for (lo, hi, cur, bound) in [(0.5, 0.8, (0.76, 0.75), None), (0.5, 0.8, (0.76, 0.75), jt), (0.5, 0.8, (0.7, 0.6), jt),
                                 (0.5, 0.8, (0.67, 0.72), None), (0.4, 0.5, (0.427, 0.468), jt), (0.4, 0.5, (0.498, 0.418), None)]:
        for s in range(8):
            rng = np.random.RandomState(1000 + s)
            draws = [g['randomise_hw'](lo, hi, rng, current_hw=cur, linf_bound=bound) for _ in range(3)]
            hw_cases.append({'min': hx(lo), 'max': hx(hi), 'current_hw': hx(cur), 'linf_bound': None if bound is None else hx(bound),
                             'seed': 1000 + s, 'draws': hx(draws), 'next_u32': int(rng.randint(0, 2 ** 31 - 1))})
    out['geom']['randomise_hw'] = hw_cases

    # ------------------------------------------------------------------ entities.py: enums and the action table
    en_ns = cut('entities.py', ['RobotAction', 'ACTION_NUMS_FLAGS_NAMES', 'ACTION_ID_TO_FLAGS', 'FLAGS_TO_ACTION_ID', 'ShapeType',
                                'ShapeColour', 'SHAPE_TYPES', 'SHAPE_COLOURS'], {'enum': enum, 'np': np})
    out['entities'] = {
        'RobotAction': {m.name: int(m) for m in en_ns['RobotAction']},
        'ACTION_NUMS_FLAGS_NAMES': [[i, [int(f) for f in flags], name] for i, flags, name in en_ns['ACTION_NUMS_FLAGS_NAMES']],
        'FLAGS_TO_ACTION_ID': [[[int(f) for f in flags], i] for flags, i in en_ns['FLAGS_TO_ACTION_ID'].items()],
        'ShapeType': [m.value for m in en_ns['ShapeType']], 'ShapeColour': [m.value for m in en_ns['ShapeColour']],
        'SHAPE_TYPES': [m.value for m in en_ns['SHAPE_TYPES']], 'SHAPE_COLOURS': [m.value for m in en_ns['SHAPE_COLOURS']],
    }
    en_mod = types.SimpleNamespace(**{k: en_ns[k] for k in ('ShapeType', 'ShapeColour', 'SHAPE_TYPES', 'SHAPE_COLOURS', 'RobotAction')})

    # ------------------------------------------------------------------ benchmarks/__init__.py: names and the registry table
    b = cut('benchmarks/__init__.py', ['_ENV_NAME_RE', 'EnvName', 'update_magical_env_name', 'DEFAULT_RES'], {'re': re, 'collections': collections})
    EnvName, update_name = b['EnvName'], b['update_magical_env_name']
    # keys of DEFAULT_PREPROC_ENTRY_POINT_WRAPPERS, in order (the values are gym wrapper factories: not evaluated)
    preprocs = None
    for node in tree_of('benchmarks/__init__.py').body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == 'DEFAULT_PREPROC_ENTRY_POINT_WRAPPERS' for t in node.targets):
            preprocs = [elt.elts[0].value for elt in node.value.args[0].elts]
    assert preprocs and all(isinstance(p, str) for p in preprocs)
    loc = cut_function_locals('benchmarks/__init__.py', 'register_envs', ['env_epoint_suffix_kwargs', 'common_kwargs', 'debug_mtc_suffix', 'mtc_ep_len'],
                              {'DEFAULT_RES': b['DEFAULT_RES']})
    specs = loc['env_epoint_suffix_kwargs']
    all_names = []
    for epoint, name, ep_len, kwargs in specs:       # the registration loop's order (benchmarks/__init__.py:979-999)
        all_names.append(name)
        for p in preprocs:
            all_names.append(update_name(name, preproc=p))
    dbg = f"MoveToCorner-Demo-{loc['debug_mtc_suffix']}-v0"       # :1030-1047
    all_names.append(dbg)
    all_names += [f"MoveToCorner-Demo-{loc['debug_mtc_suffix']}-{p}-v0" for p in preprocs]
    demo_to_test = {}
    for n in all_names[:len(specs) * (1 + len(preprocs))]:
        e = EnvName(n)
        if e.is_test:
            demo_to_test.setdefault(e.demo_env_name, []).append(e.env_name)
    parse_cases = {}
    for n in all_names + ['NotAnEnv', 'MoveToCorner-v0', 'MoveToCorner-Demo', 'A-Demo-B-C-v12', 'X-TestFoo-v3', '-Demo-v0', 'A-Train-v0']:
        try:
            e = EnvName(n)
            parse_cases[n] = {'task': e.task, 'variant': e.variant, 'preproc': e.preproc, 'version': e.version, 'is_test': e.is_test,
                              'demo_env_name': e.demo_env_name, 'env_name': e.env_name}
        except (ValueError, AssertionError) as ex:
            parse_cases[n] = type(ex).__name__
    upd = []
    for n, kw in [('FixColour-Demo-v0', dict(preproc='LoRes4E', variant='TestAll')), ('MoveToCorner-TestAll-LoRes4A-v0', dict(task='MakeLine')),
                  ('ClusterShape-Demo-LoResStack-v0', dict(version='v3')), ('FindDupe-TestJitter-v0', dict(variant='Demo', preproc='LoResCHW4E'))]:
        upd.append([n, kw, update_name(n, **kw)])
    out['registry'] = {
        'preprocessors': preprocs, 'DEFAULT_RES': list(b['DEFAULT_RES']),
        'common_kwargs': {k: (list(v) if isinstance(v, tuple) else v) for k, v in loc['common_kwargs'].items()},
        'specs': [[ep, n, int(l), kw] for ep, n, l, kw in specs],
        'ALL_REGISTERED_ENVS': all_names,
        'DEMO_ENVS_TO_TEST_ENVS_MAP': sorted([k, v] for k, v in demo_to_test.items()),
        'EnvName': parse_cases, 'update_magical_env_name': upd,
    }

    # ------------------------------------------------------------------ make_line.py: longest_line + the score
    ml = cut('benchmarks/make_line.py', ['longest_line', 'INLIER_RAD_MULT', 'MAX_SEP_RADS', 'MIN_BLOCKS', 'MAX_BLOCKS'], {'np': np, 'it': itertools})
    longest_line = ml['longest_line']
    SHAPE_RAD = 0.2 * 0.6                        # BaseEnv.ROBOT_RAD * 0.6 (base_env.py:63-65)
    inl, sep = ml['INLIER_RAD_MULT'] * SHAPE_RAD, ml['MAX_SEP_RADS'] * SHAPE_RAD
    rs = np.random.RandomState(20260928)
    ll_cases = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for k in range(1100):
            n = int(rs.randint(0, 7)) if k < 1000 else int(rs.randint(2, 6))
            kind = k % 5
            if kind == 0 or n < 2:
                pts = rs.uniform(-1, 1, size=(n, 2))
            elif kind in (1, 2):              # near-collinear: along a random line, jittered around the inlier distance
                p0, th = rs.uniform(-0.5, 0.5, 2), rs.uniform(0, math.pi)
                t = np.sort(rs.uniform(-0.9, 0.9, n)) if kind == 1 else np.cumsum(rs.uniform(0.2, 0.55, n)) - 0.8
                off = rs.normal(0, inl * (0.5 if kind == 1 else 1.0), n)
                pts = p0 + np.outer(t, [math.cos(th), math.sin(th)]) + np.outer(off, [-math.sin(th), math.cos(th)])
            elif kind == 3:                   # clusters / coincident points (0/0 in the unit vector)
                pts = rs.uniform(-1, 1, size=(n, 2)); pts[rs.randint(n)] = pts[rs.randint(n)]
            else:                             # on a grid: exact ties in the projections
                pts = rs.randint(-3, 4, size=(n, 2)) * 0.25
            if k >= 1000:                     # other thresholds than the task's
                a, b_ = float(rs.uniform(0.01, 0.5)), float(rs.uniform(0.05, 1.5))
            else:
                a, b_ = inl, sep
            pts = np.asarray(pts, dtype=np.float64).reshape(n, 2)
            ll_cases.append({'points': hx(pts.tolist()), 'inlier_dist': hx(a), 'max_separation': hx(b_), 'out': int(longest_line(pts, a, b_))})
    MakeLine = cut_methods('benchmarks/make_line.py', 'MakeLineEnv', ['score_on_end_of_traj'], {'np': np, 'longest_line': longest_line})
    ml_scores = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for c in ll_cases[:1000]:
            pts = np.array([[float.fromhex(v) for v in p] for p in c['points']]).reshape(-1, 2)
            if len(pts) in (3, 4):            # MIN_BLOCKS..MAX_BLOCKS
                self = Obj(_blocks=[block_at(x, y) for x, y in pts], inlier_dist=inl, max_sep=sep)
                ml_scores.append({'points': c['points'], 'score': hx(MakeLine.score_on_end_of_traj(self))})
    out['make_line'] = {'INLIER_RAD_MULT': hx(ml['INLIER_RAD_MULT']), 'MAX_SEP_RADS': hx(ml['MAX_SEP_RADS']), 'MIN_BLOCKS': ml['MIN_BLOCKS'],
                        'MAX_BLOCKS': ml['MAX_BLOCKS'], 'inlier_dist': hx(inl), 'max_sep': hx(sep), 'longest_line': ll_cases, 'scores': ml_scores}

    # ------------------------------------------------------------------ move_to_corner.py: score + shaped debug reward
    MTC = cut_methods('benchmarks/move_to_corner.py', 'MoveToCornerEnv', ['score_on_end_of_traj', 'debug_shaped_reward'], {'np': np})
    mtc = []
    pts = np.concatenate([rs.uniform(-1.1, 1.1, size=(300, 2)), np.array([[-1.0, 1.0], [1.0, -1.0], [0.0, 0.0], [-0.5, 0.5], [-1.0, 0.0], [0.1, -0.65]]),
                          np.array([[-1.0 + math.sqrt(2) / 2 * math.cos(a), 1.0 + math.sqrt(2) / 2 * math.sin(a)] for a in np.linspace(-1.5, 0, 14)])])
    robots = rs.uniform(-1, 1, size=(len(pts), 2))
    for (x, y), (rx, ry) in zip(pts, robots):
        self = MTC.__new__(MTC)
        self._MoveToCornerEnv__shape_ref = Obj(shape_body=Obj(position=(float(x), float(y))))
        self._robot = Obj(robot_body=Obj(position=(float(rx), float(ry))))
        mtc.append({'block': hx([x, y]), 'robot': hx([rx, ry]), 'score': hx(self.score_on_end_of_traj()), 'debug_shaped_reward': hx(self.debug_shaped_reward())})
    out['move_to_corner'] = mtc

    # ------------------------------------------------------------------ cluster.py: score on the Demo memberships and on random ones
    Cluster = cut_methods('benchmarks/cluster.py', 'BaseClusterEnv', ['score_on_end_of_traj'], {'np': np})
    cl_cls = {}
    for cname in ('ClusterColourEnv', 'ClusterShapeEnv'):
        for node in tree_of('benchmarks/cluster.py').body:
            if isinstance(node, ast.ClassDef) and node.name == cname:
                ns2 = {'en': en_mod}
                body = [n for n in node.body if isinstance(n, ast.Assign) and n.targets[0].id in ('DEFAULT_BLOCK_COLOURS', 'DEFAULT_BLOCK_SHAPES', 'DEFAULT_BLOCK_POSES', 'DEFAULT_ROBOT_POSE')]
                exec(compile(ast.Module(body=body, type_ignores=[]), 'cluster.py', 'exec'), ns2)
                cl_cls[cname] = ns2

    def cluster_score(labels, pos):
        # the index on_reset() builds (cluster.py:131-147): np.unique of the labels, blocks listed per label in block order
        self = Cluster.__new__(Cluster)
        c_values_list = np.asarray(labels, dtype='object')
        self._BaseClusterEnv__characteristic_values = np.unique(c_values_list)
        by = {}
        for (x, y), v in zip(pos, c_values_list):
            by.setdefault(v, []).append(block_at(x, y))
        self._BaseClusterEnv__blocks_by_characteristic = by
        return self.score_on_end_of_traj()

    def cluster_positions(labels):
        n = len(labels)
        mode = rs.randint(4)
        if mode == 0:
            return rs.uniform(-0.9, 0.9, size=(n, 2))
        cent = {v: rs.uniform(-0.7, 0.7, 2) for v in sorted(set(labels))}
        spread = [0.02, 0.08, 0.2][rs.randint(3)]
        pos = np.array([cent[v] + rs.normal(0, spread, 2) for v in labels])
        if mode == 2:                        # a few strays
            for _ in range(rs.randint(1, 3)):
                pos[rs.randint(n)] = rs.uniform(-0.9, 0.9, 2)
        return pos
    cl = {'demo': {}, 'random_labels': []}
    for task, cname, key in (('ClusterColour', 'ClusterColourEnv', 'DEFAULT_BLOCK_COLOURS'), ('ClusterShape', 'ClusterShapeEnv', 'DEFAULT_BLOCK_SHAPES')):
        labels = [v.value for v in cl_cls[cname][key]]
        cases = []
        for _ in range(250):
            pos = cluster_positions(labels)
            cases.append({'pos': hx(pos.tolist()), 'score': hx(cluster_score(cl_cls[cname][key], pos))})
        cl['demo'][task] = {'labels': labels, 'block_colours': [v.value for v in cl_cls[cname]['DEFAULT_BLOCK_COLOURS']],
                            'block_shapes': [v.value for v in cl_cls[cname]['DEFAULT_BLOCK_SHAPES']],
                            'block_poses': hx([[p[0][0], p[0][1], p[1]] for p in cl_cls[cname]['DEFAULT_BLOCK_POSES']]),
                            'robot_pose': hx([cl_cls[cname]['DEFAULT_ROBOT_POSE'][0][0], cl_cls[cname]['DEFAULT_ROBOT_POSE'][0][1], cl_cls[cname]['DEFAULT_ROBOT_POSE'][1]]),
                            'cases': cases}
    cols = list(en_ns['SHAPE_COLOURS'])
    for _ in range(200):
        n = int(rs.randint(7, 11))
        labels = cols + [cols[rs.randint(4)] for _ in range(n - 4)]
        rs.shuffle(labels)
        pos = cluster_positions([v.value for v in labels])
        cl['random_labels'].append({'labels': [v.value for v in labels], 'pos': hx(pos.tolist()), 'score': hx(cluster_score(labels, pos))})
    out['cluster'] = cl

    # ------------------------------------------------------------------ region tasks: the arithmetic on a given overlap set
    class Sensor:
        def __init__(self, overlap):
            self.overlap = overlap

        def get_overlapping_ents(self, com_overlap, ent_index):
            assert com_overlap is True
            return set(self.overlap)
    MR = cut_methods('benchmarks/match_regions.py', 'MatchRegionsEnv', ['score_on_end_of_traj'], {})
    FD = cut_methods('benchmarks/find_dupe.py', 'FindDupeEnv', ['score_on_end_of_traj'], {})
    FC = cut_methods('benchmarks/fix_colour.py', 'FixColourEnv', ['score_on_end_of_traj'], {})
    mr_cases, fd_cases, fc_cases = [], [], []
    for nt in (1, 2):
        for nd in range(0, 7):
            T, D = [object() for _ in range(nt)], [object() for _ in range(nd)]
            for kt in range(nt + 1):
                for kd in range(nd + 1):
                    self = MR.__new__(MR)
                    self._MatchRegionsEnv__sensor_ref = Sensor(T[:kt] + D[:kd])
                    self._MatchRegionsEnv__ent_index = None
                    self._MatchRegionsEnv__target_shapes, self._MatchRegionsEnv__distractor_shapes = T, D
                    mr_cases.append({'n_targets': nt, 'n_distractors': nd, 'targets_in': kt, 'distractors_in': kd, 'score': hx(self.score_on_end_of_traj())})
    for nt in (1, 2, 3, 4):
        for nd in range(0, 6):
            T, D = [object() for _ in range(nt)], [object() for _ in range(nd)]
            for kt in range(nt + 1):
                for kd in range(nd + 1):
                    self = FD.__new__(FD)
                    self._FindDupeEnv__sensor_ref = Sensor(T[:kt] + D[:kd])
                    self._FindDupeEnv__block_index = None
                    self._FindDupeEnv__target_set, self._FindDupeEnv__distractor_set = set(T), set(D)
                    fd_cases.append({'n_targets': nt, 'n_distractors': nd, 'targets_in': kt, 'distractors_in': kd, 'score': hx(self.score_on_end_of_traj())})
    for nreg in (2, 3):
        blocks = list(range(nreg))           # block k belongs to region k
        for keep in itertools.product([False, True], repeat=nreg):
            # every assignment of each block to "inside region r" or "outside everything" (-1)
            for where in itertools.product(range(-1, nreg), repeat=nreg):
                self = FC.__new__(FC)
                self._sensors = [Sensor([blk for blk in blocks if where[blk] == r]) for r in range(nreg)]
                self._target_blocks = [[blocks[r]] if keep[r] else [] for r in range(nreg)]
                self._block_index = None
                fc_cases.append({'keep': list(keep), 'block_region': list(where), 'score': hx(self.score_on_end_of_traj())})
    out['match_regions'] = mr_cases
    out['find_dupe'] = fd_cases
    out['fix_colour'] = fc_cases

    with open(OUT, 'w') as f:
        json.dump(out, f, indent=None, separators=(',', ':'))
        f.write('\n')
    print(f'wrote {OUT}: {os.path.getsize(OUT)} bytes; ' + ', '.join(f'{k}' for k in out if not k.startswith('_')))


if __name__ == '__main__':
    main()
