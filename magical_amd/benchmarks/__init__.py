"""Env-name registry: the host-side mirror of magical/benchmarks/__init__.py.

Keeps the reference's public surface -- `register_envs()`, `ALL_REGISTERED_ENVS`,
`DEMO_ENVS_TO_TEST_ENVS_MAP`, `EnvName`, `update_magical_env_name`, the
`<Task>-<Demo|Test*>[-<Preproc>]-v0` grammar and the preprocessor names -- and adds
`make(name, n_envs=..., device=...)`, which returns a batched engine instead of a
gym.Env (gym is not a dependency; if `gymnasium`/`gym` is importable the names are also
registered there with n_envs=1 entry points).

Observation layouts (torch tensors on the engine's device):
  <Task>-<Variant>-v0            f32[N, n_bodies, 3]   state-only (x, y, angle); the reference's
                                 384x384 allo/ego dict is available per env via env.render()
  ...-LoRes4E-v0                 u8[N, 96, 96, 12]     4 ego frames, oldest first (benchmarks/__init__.py:252-256)
  ...-LoRes4A-v0                 u8[N, 96, 96, 12]     4 allo frames
  ...-LoResCHW4E-v0              u8[N, 12, 96, 96]     channels-first view of LoRes4E
  ...-LoRes3EA-v0                u8[N, 96, 96, 12]     [allo_t, ego_t-2, ego_t-1, ego_t] (:242-245)
  ...-LoResStack-v0              {'allo': u8[N, 96, 96, 12], 'ego': u8[N, 96, 96, 12]}  4 frames each (:257-261)
"""
import collections
import importlib
import re

__all__ = ['ALL_REGISTERED_ENVS', 'DEMO_ENVS_TO_TEST_ENVS_MAP', 'register_envs', 'make', 'EnvName',
           'update_magical_env_name', 'AVAILABLE_PREPROCESSORS']

DEFAULT_RES = (384, 384)     # benchmarks/__init__.py:23
COMMON_KWARGS = dict(res_hw=DEFAULT_RES, fps=8, phys_steps=10, phys_iter=10)   # :400-403

# (module, class, task, variant, episode length, randomisation flags that are True)  -- :405-960
_ENV_TABLE = [
    ('cluster', 'ClusterShapeEnv', 'ClusterShape', 'Demo', 240, ()),
    ('cluster', 'ClusterShapeEnv', 'ClusterShape', 'TestJitter', 240, ('rand_layout_minor',)),
    ('cluster', 'ClusterShapeEnv', 'ClusterShape', 'TestColour', 240, ('rand_shape_colour',)),
    ('cluster', 'ClusterShapeEnv', 'ClusterShape', 'TestShape', 240, ('rand_shape_type',)),
    ('cluster', 'ClusterShapeEnv', 'ClusterShape', 'TestLayout', 240, ('rand_layout_full',)),
    ('cluster', 'ClusterShapeEnv', 'ClusterShape', 'TestCountPlus', 240, ('rand_shape_colour', 'rand_shape_type', 'rand_layout_full', 'rand_shape_count')),
    ('cluster', 'ClusterShapeEnv', 'ClusterShape', 'TestDynamics', 240, ('rand_dynamics',)),
    ('cluster', 'ClusterShapeEnv', 'ClusterShape', 'TestAll', 240, ('rand_shape_colour', 'rand_shape_type', 'rand_layout_full', 'rand_shape_count', 'rand_dynamics')),
    ('cluster', 'ClusterColourEnv', 'ClusterColour', 'Demo', 240, ()),
    ('cluster', 'ClusterColourEnv', 'ClusterColour', 'TestJitter', 240, ('rand_layout_minor',)),
    ('cluster', 'ClusterColourEnv', 'ClusterColour', 'TestColour', 240, ('rand_shape_colour',)),
    ('cluster', 'ClusterColourEnv', 'ClusterColour', 'TestShape', 240, ('rand_shape_type',)),
    ('cluster', 'ClusterColourEnv', 'ClusterColour', 'TestLayout', 240, ('rand_layout_full',)),
    ('cluster', 'ClusterColourEnv', 'ClusterColour', 'TestCountPlus', 240, ('rand_shape_colour', 'rand_shape_type', 'rand_layout_full', 'rand_shape_count')),
    ('cluster', 'ClusterColourEnv', 'ClusterColour', 'TestDynamics', 240, ('rand_dynamics',)),
    ('cluster', 'ClusterColourEnv', 'ClusterColour', 'TestAll', 240, ('rand_shape_colour', 'rand_shape_type', 'rand_layout_full', 'rand_shape_count', 'rand_dynamics')),
    ('find_dupe', 'FindDupeEnv', 'FindDupe', 'Demo', 100, ()),
    ('find_dupe', 'FindDupeEnv', 'FindDupe', 'TestJitter', 100, ('rand_layout_minor',)),
    ('find_dupe', 'FindDupeEnv', 'FindDupe', 'TestColour', 100, ('rand_colours',)),
    ('find_dupe', 'FindDupeEnv', 'FindDupe', 'TestShape', 100, ('rand_shapes',)),
    ('find_dupe', 'FindDupeEnv', 'FindDupe', 'TestLayout', 100, ('rand_layout_full',)),
    ('find_dupe', 'FindDupeEnv', 'FindDupe', 'TestCountPlus', 100, ('rand_colours', 'rand_shapes', 'rand_count', 'rand_layout_full')),
    ('find_dupe', 'FindDupeEnv', 'FindDupe', 'TestDynamics', 100, ('rand_dynamics',)),
    ('find_dupe', 'FindDupeEnv', 'FindDupe', 'TestAll', 100, ('rand_colours', 'rand_shapes', 'rand_count', 'rand_layout_full', 'rand_dynamics')),
    ('fix_colour', 'FixColourEnv', 'FixColour', 'Demo', 60, ()),
    ('fix_colour', 'FixColourEnv', 'FixColour', 'TestJitter', 60, ('rand_layout_minor',)),
    ('fix_colour', 'FixColourEnv', 'FixColour', 'TestColour', 60, ('rand_colours',)),
    ('fix_colour', 'FixColourEnv', 'FixColour', 'TestShape', 60, ('rand_shapes',)),
    ('fix_colour', 'FixColourEnv', 'FixColour', 'TestLayout', 60, ('rand_layout_full',)),
    ('fix_colour', 'FixColourEnv', 'FixColour', 'TestCountPlus', 60, ('rand_colours', 'rand_shapes', 'rand_count', 'rand_layout_full')),
    ('fix_colour', 'FixColourEnv', 'FixColour', 'TestDynamics', 60, ('rand_dynamics',)),
    ('fix_colour', 'FixColourEnv', 'FixColour', 'TestAll', 60, ('rand_colours', 'rand_shapes', 'rand_count', 'rand_layout_full', 'rand_dynamics')),
    ('make_line', 'MakeLineEnv', 'MakeLine', 'Demo', 180, ()),
    ('make_line', 'MakeLineEnv', 'MakeLine', 'TestJitter', 180, ('rand_layout_minor',)),
    ('make_line', 'MakeLineEnv', 'MakeLine', 'TestColour', 180, ('rand_colours',)),
    ('make_line', 'MakeLineEnv', 'MakeLine', 'TestShape', 180, ('rand_shapes',)),
    ('make_line', 'MakeLineEnv', 'MakeLine', 'TestLayout', 180, ('rand_layout_full',)),
    ('make_line', 'MakeLineEnv', 'MakeLine', 'TestCountPlus', 180, ('rand_colours', 'rand_shapes', 'rand_count', 'rand_layout_full')),
    ('make_line', 'MakeLineEnv', 'MakeLine', 'TestDynamics', 180, ('rand_dynamics',)),
    ('make_line', 'MakeLineEnv', 'MakeLine', 'TestAll', 180, ('rand_colours', 'rand_shapes', 'rand_count', 'rand_layout_full', 'rand_dynamics')),
    ('match_regions', 'MatchRegionsEnv', 'MatchRegions', 'Demo', 120, ()),
    ('match_regions', 'MatchRegionsEnv', 'MatchRegions', 'TestJitter', 120, ('rand_layout_minor',)),
    ('match_regions', 'MatchRegionsEnv', 'MatchRegions', 'TestColour', 120, ('rand_target_colour',)),
    ('match_regions', 'MatchRegionsEnv', 'MatchRegions', 'TestShape', 120, ('rand_shape_type',)),
    ('match_regions', 'MatchRegionsEnv', 'MatchRegions', 'TestLayout', 120, ('rand_layout_full',)),
    ('match_regions', 'MatchRegionsEnv', 'MatchRegions', 'TestCountPlus', 120, ('rand_target_colour', 'rand_shape_type', 'rand_shape_count', 'rand_layout_full')),
    ('match_regions', 'MatchRegionsEnv', 'MatchRegions', 'TestDynamics', 120, ('rand_dynamics',)),
    ('match_regions', 'MatchRegionsEnv', 'MatchRegions', 'TestAll', 120, ('rand_target_colour', 'rand_shape_type', 'rand_shape_count', 'rand_layout_full', 'rand_dynamics')),
    ('move_to_corner', 'MoveToCornerEnv', 'MoveToCorner', 'Demo', 80, ()),
    ('move_to_corner', 'MoveToCornerEnv', 'MoveToCorner', 'TestColour', 80, ('rand_shape_colour',)),
    ('move_to_corner', 'MoveToCornerEnv', 'MoveToCorner', 'TestShape', 80, ('rand_shape_type',)),
    ('move_to_corner', 'MoveToCornerEnv', 'MoveToCorner', 'TestJitter', 80, ('rand_poses',)),
    ('move_to_corner', 'MoveToCornerEnv', 'MoveToCorner', 'TestDynamics', 80, ('rand_dynamics',)),
    ('move_to_corner', 'MoveToCornerEnv', 'MoveToCorner', 'TestAll', 80, ('rand_shape_colour', 'rand_shape_type', 'rand_poses', 'rand_dynamics')),
    ('move_to_region', 'MoveToRegionEnv', 'MoveToRegion', 'Demo', 40, ()),
    ('move_to_region', 'MoveToRegionEnv', 'MoveToRegion', 'TestJitter', 40, ('rand_poses_minor',)),
    ('move_to_region', 'MoveToRegionEnv', 'MoveToRegion', 'TestColour', 40, ('rand_goal_colour',)),
    ('move_to_region', 'MoveToRegionEnv', 'MoveToRegion', 'TestLayout', 40, ('rand_poses_full',)),
    ('move_to_region', 'MoveToRegionEnv', 'MoveToRegion', 'TestDynamics', 40, ('rand_dynamics',)),
    ('move_to_region', 'MoveToRegionEnv', 'MoveToRegion', 'TestAll', 40, ('rand_poses_full', 'rand_goal_colour', 'rand_dynamics')),
]

AVAILABLE_PREPROCESSORS = ['LoRes3EA', 'LoRes4E', 'LoRes4A', 'LoResStack', 'LoResCHW4E']   # :242-274
_BUILT_PREPROCESSORS = ('LoRes3EA', 'LoRes4E', 'LoRes4A', 'LoResStack', 'LoResCHW4E')

_ENV_NAME_RE = re.compile(
    r'^(?P<name_prefix>[^-]+)(?P<demo_test_spec>-(Demo|Test[^-]*))'
    r'(?P<env_name_suffix>(-[^-]+)*)(?P<version_suffix>-v\d+)$')
_REGISTERED = False
DEMO_ENVS_TO_TEST_ENVS_MAP = collections.OrderedDict()
ALL_REGISTERED_ENVS = []
_SPECS = {}


class EnvName:
    """Parser for `<name_prefix>-<demo_test_spec>[-<suffix>]-<version_suffix>` (benchmarks/__init__.py:317-391)."""

    def __init__(self, env_name):
        match = _ENV_NAME_RE.match(env_name)
        if match is None:
            raise ValueError(f"env name '{env_name}' does not match _ENV_NAME_RE spec")
        groups = match.groupdict()
        self.name_prefix = groups['name_prefix']
        self.demo_test_spec = groups['demo_test_spec']
        self.env_name_suffix = groups['env_name_suffix']
        self.version_suffix = groups['version_suffix']
        assert env_name == self.env_name

    @property
    def env_name(self):
        return self.name_prefix + self.demo_test_spec + self.env_name_suffix + self.version_suffix

    @property
    def is_test(self):
        return self.demo_test_spec.startswith('-Test')

    @property
    def demo_env_name(self):
        return self.name_prefix + '-Demo' + self.env_name_suffix + self.version_suffix

    @property
    def task(self):
        return self.name_prefix

    @property
    def variant(self):
        return self.demo_test_spec.strip('-')

    @property
    def preproc(self):
        return self.env_name_suffix.strip('-') if self.env_name_suffix else None

    @property
    def version(self):
        return self.version_suffix.strip('-')


def update_magical_env_name(env_name, *, task=None, variant=None, preproc=None, version=None):
    """benchmarks/__init__.py:285-314."""
    ename = EnvName(env_name)
    parts = [task if task is not None else ename.task, variant if variant is not None else ename.variant]
    if preproc is None:
        preproc = ename.preproc
    if preproc is not None:
        parts.append(preproc)
    parts.append(version if version is not None else ename.version)
    return '-'.join(parts)


def register_envs():
    """Register all default environment names (idempotent; returns False if already done)."""
    global _REGISTERED
    if _REGISTERED:
        return False
    _REGISTERED = True
    for module, cls, task, variant, ep_len, flags in _ENV_TABLE:
        base = f'{task}-{variant}-v0'
        spec = dict(module=module, cls=cls, ep_len=ep_len, flags=flags, debug_reward=False)
        for name, preproc in [(base, None)] + [(update_magical_env_name(base, preproc=p), p) for p in AVAILABLE_PREPROCESSORS]:
            ALL_REGISTERED_ENVS.append(name)
            _SPECS[name] = dict(spec, preproc=preproc)
    train_to_test = {}
    for name in ALL_REGISTERED_ENVS:
        parsed = EnvName(name)
        if parsed.is_test:
            train_to_test.setdefault(parsed.demo_env_name, []).append(parsed.env_name)
    DEMO_ENVS_TO_TEST_ENVS_MAP.update(sorted((k, tuple(v)) for k, v in train_to_test.items()))
    # MoveToCorner-Demo-DebugReward[-<preproc>]-v0 (benchmarks/__init__.py:1021-1047).  As in the reference, the
    # preprocessor-suffixed names point at the plain env class too (its registration loop never applies the wrapper)
    for name in ['MoveToCorner-Demo-DebugReward-v0'] + [f'MoveToCorner-Demo-DebugReward-{p}-v0' for p in AVAILABLE_PREPROCESSORS]:
        ALL_REGISTERED_ENVS.append(name)
        _SPECS[name] = dict(module='move_to_corner', cls='MoveToCornerEnv', ep_len=80, flags=(), debug_reward=True, preproc=None)
    _register_with_gym()
    return True


def _register_with_gym():
    for modname in ('gymnasium', 'gym'):
        try:
            gym = importlib.import_module(modname)
        except Exception:
            continue
        for name in ALL_REGISTERED_ENVS:
            try:
                gym.register(name, entry_point=lambda _n=name, **kw: make(_n, **kw), max_episode_steps=_SPECS[name]['ep_len'])
            except Exception:
                pass


def make(name, n_envs=1, device='cuda:0', **kwargs):
    """Batched counterpart of gym.make(name): N lockstep envs of `name` on one MI355X."""
    register_envs()
    if name not in _SPECS:
        raise KeyError(f"unknown MAGICAL env '{name}' (see magical_amd.ALL_REGISTERED_ENVS)")
    spec = _SPECS[name]
    preproc = spec['preproc']
    if preproc is not None and preproc not in _BUILT_PREPROCESSORS:
        raise NotImplementedError(f"preprocessor '{preproc}' is registered but not built yet (built: {_BUILT_PREPROCESSORS})")
    mod = importlib.import_module(f'magical_amd.benchmarks.{spec["module"]}')
    env_cls = getattr(mod, spec['cls'])
    from .preproc import wrap_preproc
    cls = wrap_preproc(env_cls, preproc)
    env_kwargs = dict(COMMON_KWARGS, max_episode_steps=spec['ep_len'], **{f: True for f in spec['flags']})
    if spec['debug_reward']:
        env_kwargs['debug_reward'] = True
    env_kwargs.update(kwargs)
    env = cls(n_envs=n_envs, device=device, **env_kwargs)
    env.spec_name = name
    return env
