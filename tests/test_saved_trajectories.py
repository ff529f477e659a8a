"""Demo files (magical_amd/saved_trajectories.py): loading the reference's pickle layout without the reference package,
preprocessing recorded full-resolution frames like the reference's wrapper stacks, replaying action tapes."""
import collections
import gzip
import os
import pickle
import sys
import types

import numpy as np
import pytest


def _fake_reference_module():
    """A stand-in for the module path the reference's demo pickles name (magical.saved_trajectories.MAGICALTrajectory)."""
    pkg, mod = types.ModuleType('magical'), types.ModuleType('magical.saved_trajectories')
    cls = collections.namedtuple('MAGICALTrajectory', ['acts', 'obs', 'rews', 'infos'])
    cls.__module__ = 'magical.saved_trajectories'
    mod.MAGICALTrajectory = cls
    pkg.saved_trajectories = mod
    return pkg, mod, cls


def _write_demo(path, T=5, seed=0, env_name='MoveToCorner-Demo-v0', res=384):
    rng = np.random.RandomState(seed)
    pkg, mod, cls = _fake_reference_module()
    sys.modules['magical'], sys.modules['magical.saved_trajectories'] = pkg, mod
    try:
        traj = cls(acts=rng.randint(0, 18, size=T), obs={k: rng.randint(0, 256, size=(T + 1, res, res, 3)).astype(np.uint8) for k in ('allo', 'ego')},
                   rews=np.zeros(T), infos=[{'eval_score': 0.0} for _ in range(T)])
        with gzip.GzipFile(path, 'wb') as fp:
            pickle.dump({'trajectory': traj, 'score': 0.75, 'env_name': env_name}, fp)
    finally:
        del sys.modules['magical'], sys.modules['magical.saved_trajectories']
    return traj


def test_load_demos_reads_reference_pickles_and_refuses_anything_else(tmp_path):
    from magical_amd import saved_trajectories as st
    paths = [str(tmp_path / f'demo-{k}.pkl.gz') for k in range(2)]
    written = [_write_demo(p, seed=k) for k, p in enumerate(paths)]
    assert 'magical' not in sys.modules
    demos = list(st.load_demos(paths))
    assert len(demos) == 2
    for d, w in zip(demos, written):
        assert isinstance(d['trajectory'], st.MAGICALTrajectory) and d['env_name'] == 'MoveToCorner-Demo-v0' and d['score'] == 0.75
        assert np.array_equal(d['trajectory'].acts, w.acts) and np.array_equal(d['trajectory'].obs['ego'], w.obs['ego'])
    evil = str(tmp_path / 'evil.pkl.gz')
    with gzip.GzipFile(evil, 'wb') as fp:
        pickle.dump({'trajectory': os.path.join}, fp)
    with pytest.raises(pickle.UnpicklingError):
        list(st.load_demos([evil]))
    # globals inside the packages a demo legitimately uses are refused too: the allow-list is by exact (module, name)
    import builtins
    import numpy.testing
    for k, obj in enumerate((numpy.testing.assert_equal, builtins.setattr, builtins.map, builtins.breakpoint, np.load, np.fromfile)):
        bad = str(tmp_path / f'evil-{k}.pkl.gz')
        with gzip.GzipFile(bad, 'wb') as fp:
            pickle.dump({'trajectory': obj}, fp)
        with pytest.raises(pickle.UnpicklingError):
            list(st.load_demos([bad]))
    assert st.splice_in_preproc_name('MoveToCorner-Demo-v0', 'LoResStack') == 'MoveToCorner-Demo-LoResStack-v0'
    with pytest.raises(AssertionError):
        st.splice_in_preproc_name('MoveToCorner-Demo-v0', 'NoSuchPreproc')


class _RecordedEnv:
    """What the reference's _MockDemoEnv is to its wrappers (saved_trajectories.py:61-78), for the oracle's wrapper
    restatements: render() returns the recorded full-resolution frame of the current step."""

    def __init__(self, obs):
        self.obs, self.t = obs, 0

    def reset(self):
        self.t = 0

    def step(self, action):
        self.t += 1
        return 0.0, False, {}

    def render(self, view):
        return self.obs[view][self.t]


@pytest.mark.parametrize('preproc', ['LoRes4E', 'LoRes4A', 'LoRes3EA', 'LoResStack', 'LoResCHW4E'])
def test_preprocess_demos_matches_the_wrapper_stacks(tmp_path, preproc):
    """preprocess_demos_with_wrapper == replaying the recorded frames through the oracle's restatement of the wrapper
    stacks (FlattenFrameStack / EagerDictFrameStack + INTER_AREA resize), step by step, bit for bit."""
    from magical_amd import saved_trajectories as st
    from oracle import env_ref
    path = str(tmp_path / 'demo.pkl.gz')
    _write_demo(path, T=6, seed=3)
    demo = next(st.load_demos([path]))
    traj = demo['trajectory']
    new, = st.preprocess_demos_with_wrapper([traj], demo['env_name'], preproc_name=preproc)
    assert np.array_equal(new.acts, traj.acts) and len(new.infos) == len(traj.infos)
    rec = _RecordedEnv(traj.obs)
    if preproc in ('LoRes4E', 'LoResCHW4E', 'LoRes4A'):
        ref = env_ref.LoRes4ERef(rec)
        if preproc == 'LoRes4A':
            rec.render = lambda view, _r=rec: _r.obs['allo'][_r.t]           # the same stack over the other view
    else:
        ref = env_ref.LoRes3EARef(rec) if preproc == 'LoRes3EA' else env_ref.LoResStackRef(rec)
    want = [ref.reset()] + [ref.step(0)[0] for _ in range(len(traj.acts))]
    for t, w in enumerate(want):
        if preproc == 'LoResStack':
            assert all(np.array_equal(new.obs[k][t], w[k]) for k in ('allo', 'ego')), t
        elif preproc == 'LoResCHW4E':
            assert np.array_equal(new.obs[t], np.moveaxis(w, 2, 0)), t
        else:
            assert np.array_equal(new.obs[t], w), t


def test_area_resize_rounds_half_to_even():
    from magical_amd.saved_trajectories import area_resize_4x
    f = np.zeros((4, 4, 1), dtype=np.uint8)
    f[0, :, 0] = [2, 2, 2, 2]            # sum 8 -> 0.5 -> 0 (even)
    assert area_resize_4x(f)[0, 0, 0] == 0
    f[1, :, 0] = [4, 4, 4, 4]            # sum 24 -> 1.5 -> 2 (even)
    assert area_resize_4x(f)[0, 0, 0] == 2


@pytest.mark.gpu
def test_replay_demos_scores_match_the_oracle(tmp_path):
    """replay_demos: action tapes of unequal length through the engine (one env per demo, fp64 build), each scored at its
    own last step -> the oracle's score for the same tape."""
    from magical_amd import saved_trajectories as st
    from oracle.env_ref import RefEnv
    paths = [str(tmp_path / f'demo-{k}.pkl.gz') for k in range(3)]
    for k, p in enumerate(paths):
        _write_demo(p, T=6 + 3 * k, seed=10 + k, env_name='MoveToCorner-Demo-v0', res=8)
    demos = list(st.load_demos(paths))
    out = st.replay_demos(demos, dtype='f64')
    assert np.all(out['recorded_scores'] == 0.75)
    for k, d in enumerate(demos):
        ref = RefEnv('MoveToCorner', max_episode_steps=len(d['trajectory'].acts))
        ref.reset()
        for a in d['trajectory'].acts:
            _, done, info = ref.step(int(a))
        assert done and abs(info['eval_score'] - out['scores'][k]) < 1e-9, (k, info['eval_score'], out['scores'][k])
