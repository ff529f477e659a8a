"""Consumers of a pymunk capture: `python -m oracle.pymunk_backend --capture` writes, wherever pymunk 5.6 / 5.7
imports, the real engine's per-substep states of fixed tapes on the repo's own world tables (masses, star parts, scores too) into
tests/golden/pymunk_capture.json.  With that ONE file committed, the tests below hold

  * oracle/magical_ref.c (CPU): masses / moments to 1e-12, the star's union, poses substep by substep -- round-off over the first
    env-step, inside the oracle's own perturbation envelope afterwards, the end-of-tape score;
  * the HIP engine (-m gpu): poses at env-steps 1 / 5 / 20 against the captured ones inside the shipped precision's envelope, and
    the score of the state reached

to pymunk's results -- on boxes where pymunk itself cannot be imported (this container, the GPU box).  Without the file they SKIP
LOUDLY, naming the command.  What always runs is the same consumer code on a capture the C oracle makes of itself (plumbing: such a
capture pins nothing and says so).  No reference file is imported, copied or shipped by any of this."""
import os

import numpy as np
import pytest

from oracle import pymunk_backend as pb
from tests.util import (EPS_F32, EPS_F64, F32_OPS_FACTOR, FLOOR_F32, FLOOR_F64, TASKS, EnvelopeTally, OracleEnvelope, comparable_mask,
                        masked_err, new_ref, ref_body_index)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, pb.CAPTURE_FIXTURE)
HOW = ('PARITY UNPINNED FOR POSES: no pymunk capture at tests/golden/pymunk_capture.json.  On any machine where `pip install '
       '"pymunk~=5.6.0"` works, run  `python -m oracle.pymunk_backend --capture`  in this repo and commit the file it writes: '
       'these tests then pin the C oracle and the HIP engine on the real Chipmunk step.')


def _fixture():
    if not os.path.exists(FIXTURE):
        pytest.skip(HOW)
    cap = pb.load_capture(FIXTURE)
    assert cap['backend'] == 'pymunk' and str(cap['pymunk_version']).startswith(pb.SUPPORTED), \
        f'{FIXTURE} was not captured from pymunk 5.6 / 5.7 (backend {cap["backend"]!r}, version {cap["pymunk_version"]!r})'
    return cap


_SELF = {}


def _self_capture(tmp_path_factory):
    """The C oracle's capture of itself, through the file format (written once per session)."""
    if 'cap' not in _SELF:
        path = str(tmp_path_factory.mktemp('capture') / 'c_capture.json')
        assert pb.main(['--capture', path, '--backend', 'c']) == 0
        _SELF['cap'] = pb.load_capture(path)
        assert _SELF['cap']['backend'] == 'c'
    return _SELF['cap']


# ---------------------------------------------------------------------------------------------------------------- consumers
def check_c_oracle_against(cap, task, exact):
    """oracle/magical_ref.c on the capture's tape, substep by substep.  exact: the capture IS the C oracle's (plumbing)."""
    rec = cap['tasks'][task]
    own = pb.capture_task(task, steps=len(rec['tape']), backend='c')
    assert list(own['tape']) == list(rec['tape']), 'the capture was made on another tape (oracle.pymunk_backend.fixed_tape changed?)'
    assert own['states'].shape == rec['states'].shape, (own['states'].shape, rec['states'].shape)
    assert np.allclose(own['mass'], rec['mass'], rtol=1e-12, atol=0), (task, 'inverse masses / moments differ from the captured ones')
    ref = new_ref(task)
    idx, mask = ref_body_index(ref), comparable_mask(ref)
    err = np.array([masked_err(a[idx][:, :3], b[idx][:, :3], mask) for a, b in zip(own['states'], rec['states'])])
    print(f'{task}: |pose(C oracle) - pose(capture)| after substeps 1 / 10 / 50 / 200: ' + ' / '.join(f'{err[min(k, len(err) - 1)]:.2e}' for k in (1, 10, 50, 200)))
    assert err[0] == 0.0, 'initial poses differ'
    if exact:
        assert err.max() == 0.0
        assert own['score'] == rec['score']
        return err
    assert err[10] < 1e-9, (task, 'first env-step beyond round-off', err[:11])
    # afterwards: no further from the real engine than the oracle's own replicas (poses perturbed by 1e-13) are from the oracle
    orc = OracleEnvelope([lambda: new_ref(task)], K=8, eps=EPS_F64, seed=1)
    tally = EnvelopeTally()
    for s, a in enumerate(rec['tape']):
        orc.step([int(a)])
        tally.check(np.array([err[10 * (s + 1)]]), orc.running, 2.0, FLOOR_F64, (task, 'env-step', s))
    tally.assert_mostly_decided(what=task)
    assert abs(own['score'] - rec['score']) < 1e-6 or err.max() > 1e-6, (task, own['score'], rec['score'])
    return err


def check_hip_engine_against(cap, task, exact_oracle):
    """The shipped fp32 engine, free running from reset on the capture's tape: poses at env-steps 1 / 5 / 20 against the captured
    ones, no further off than F32_OPS_FACTOR x the spread of the C oracle's own replicas (poses perturbed by 1e-7, velocities stored in
    fp32: tests/util.py), and the score of the state reached (the host's score functions on the engine's poses)."""
    import magical_amd
    rec = cap['tasks'][task]
    tape = np.asarray(rec['tape'], dtype=np.int32)
    env = magical_amd.make(f'{task}-Demo-v0', n_envs=4, device='cuda:0', max_episode_steps=len(tape))      # the episode ends with the tape: its
    env.auto_reset = False                                                                               # score is the captured state's
    env.reset()
    ref = new_ref(task)
    idx, mask = ref_body_index(ref), comparable_mask(ref)
    assert np.allclose(env.get_bodies()[0, 1:, :3][mask], rec['states'][0][idx][:, :3][mask], rtol=0, atol=1e-12), 'initial poses differ from the captured ones'
    orc = OracleEnvelope([lambda: new_ref(task)], K=8, eps=EPS_F32, seed=0, fp32_state=True)
    tally, shown = EnvelopeTally(), []
    for s, a in enumerate(tape):
        obs, rew, done, info = env.step(np.full(4, a, dtype=np.int32))
        orc.step([int(a)])
        if s + 1 in (1, 5, 20):
            got = env.get_bodies()[:, 1:, :3]
            want = rec['states'][10 * (s + 1)][idx][:, :3]
            errs = np.array([masked_err(got[k], want, mask) for k in range(4)])
            assert np.all(errs == errs[0]), 'identical envs on identical tapes diverged'
            tally.check(errs[:1], orc.running, F32_OPS_FACTOR, FLOOR_F32, (task, 'env-step', s + 1))
            shown.append(f'{s + 1}: {errs[0]:.2e} (envelope {F32_OPS_FACTOR * orc.running[0]:.2e})')
    print(f'{task}: |pose(HIP engine) - pose(capture)| at env-steps ' + ', '.join(shown))
    tally.assert_mostly_decided(fraction=0.3, what=task)
    assert done.all()
    score = float(info['eval_score'][0])
    last = float(np.abs(env.get_bodies()[0, 1:, :3] - rec['states'][-1][idx][:, :3])[mask].max())
    assert abs(score - rec['score']) < 1e-3 or last > 1e-4, (task, score, rec['score'], last)
    env.close()


# ---------------------------------------------------------------------------------------------------------------- the pin (needs the file)
def test_capture_fixture_is_from_pymunk():
    cap = _fixture()
    assert set(cap['tasks']) == set(TASKS)
    print(f'pymunk capture present: pymunk {cap["pymunk_version"]}, {len(cap["tasks"])} tasks')


def test_mass_table_and_star_union_equal_the_captured_ones():
    cap = _fixture()
    from oracle import geom_ref as gt
    table = cap['mass_table']
    assert abs(table['robot'][1] - gt.moment_for_circle(1.0, 0, 0.2)) < 1e-15
    assert abs(table['circle'][1] - gt.moment_for_circle(0.5, 0, 0.12)) < 1e-15
    star = gt.compute_star_verts(5, 1.3 * 0.12, 0.65 * 0.12)
    area = lambda p: 0.5 * abs(sum(p[i][0] * p[(i + 1) % len(p)][1] - p[(i + 1) % len(p)][0] * p[i][1] for i in range(len(p))))
    closed = [p[:-1] if p[0] == p[-1] else p for p in cap['star_parts']]
    assert abs(sum(area(p) for p in closed) - area(star)) < 1e-12          # pymunk's parts cover the outline ...
    assert abs(sum(area(p) for p in gt.star_convex_parts(star)) - area(star)) < 1e-12      # ... and so do the oracle's


@pytest.mark.parametrize('task', TASKS)
def test_c_oracle_tracks_the_captured_pymunk_states(task):
    check_c_oracle_against(_fixture(), task, exact=False)


@pytest.mark.gpu
@pytest.mark.parametrize('task', TASKS)
def test_hip_engine_tracks_the_captured_pymunk_states(task):
    check_hip_engine_against(_fixture(), task, exact_oracle=False)


# ---------------------------------------------------------------------------------------------------------------- plumbing (always runs)
@pytest.mark.parametrize('task', ['MoveToCorner', 'ClusterShape'])
def test_consumer_on_the_c_oracles_own_capture(task, tmp_path_factory):
    """The file format and the CPU consumer, end to end, on a capture the C oracle makes of itself: zero error by construction."""
    check_c_oracle_against(_self_capture(tmp_path_factory), task, exact=True)
    check_c_oracle_against(_self_capture(tmp_path_factory), task, exact=False)       # (the gates a pymunk capture goes through)


def test_a_self_capture_is_not_accepted_as_the_fixture(tmp_path_factory):
    cap = _self_capture(tmp_path_factory)
    assert cap['backend'] == 'c' and cap['pymunk_version'] is None and cap['mass_table'] is None


def test_skip_message_names_the_command():
    assert 'python -m oracle.pymunk_backend --capture' in HOW and 'tests/golden/pymunk_capture.json' in HOW


@pytest.mark.gpu
@pytest.mark.parametrize('task', ['MoveToCorner', 'MatchRegions', 'ClusterColour'])
def test_hip_consumer_on_the_c_oracles_own_capture(task, tmp_path_factory):
    """The GPU consumer on the C oracle's capture of itself (what it would do with a pymunk capture, with the oracle in pymunk's place)."""
    check_hip_engine_against(_self_capture(tmp_path_factory), task, exact_oracle=True)
