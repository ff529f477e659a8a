"""CPU checks of the KERNEL LOGIC: magical_amd/csrc/mgx_sim.h + mgx_raster.h compiled for the host
(tests/emu, a test harness that the product never loads) against the oracle.  These run without a GPU and
catch algorithmic mismatches (SAT narrowphase vs GJK/EPA, contact cache, joint solver, rasteriser
classification) before GPU time is spent; the real parity tests through the C ABI are in test_gpu_parity.py.
"""
import zlib

import ctypes as C

import numpy as np
import pytest

from tests.emu.emu import EmuBatch
from tests.util import TASKS, comparable_mask, new_ref, ref_body_index, ref_entities_as_tuples


def _pair(task, mode, n=1):
    ref = new_ref(task)
    em = EmuBatch(ref_entities_as_tuples(ref), ref.max_episode_steps, n, mode=mode)
    em.reset()
    return ref, em


@pytest.mark.parametrize('task', TASKS)
def test_reset_state_matches_oracle(task):
    ref, em = _pair(task, 'f64')
    idx, mask = ref_body_index(ref), comparable_mask(ref)
    got = em.bodies()[0, 1:, :3]
    want = ref.bodies()[idx][:, :3]
    assert np.array_equal(got[mask], want[mask])          # bit-identical initial poses (incl. finger roots)
    assert np.all(em.bodies()[0, :, 3:] == 0)


@pytest.mark.parametrize('task', TASKS)
@pytest.mark.parametrize('nl', [16, 32, 64])
def test_f64_phases_one_step_equivalence(task, nl):
    """FULL-STATE teacher forcing: every env-step starts from the oracle's body state and from a cold solver state on both sides
    (accumulated joint / contact impulses zeroed, arbiters and cache entries kept: ref_clear_warm / the impulse rows of the motion
    blob); the fp64 phases then reproduce the oracle's next state to round-off on EVERY step, for any lane count (the phases are
    lane-count independent).  (Round 3 forced the bodies only and had to allow a couple of steps per tape up to 5e-2: in a pressed
    contact the warm-start impulses are sloppy variables that stay apart once an amplified step has separated them, and surface at
    the next stick / slip change -- tests/test_gpu_parity.py::test_f64_engine_one_step_equivalence_and_contact_coverage.)"""
    ref, em = _pair(task, 'f64')
    idx, mask = ref_body_index(ref), comparable_mask(ref)
    rng = np.random.RandomState(zlib.crc32(task.encode()) % 1000)      # (hash() of a str changes from process to process)
    warm_row0 = max((m >> 12) for m in em.rows if (m & 15) >= 3) + 1 + 5      # motion blob: velocity rows, 5 force limits, then the impulses
    errs = []
    for t in range(40):
        a = rng.randint(18) if t % 3 else 1       # bias towards driving forward into things
        eb = em.bodies()
        eb[0, 1:, :] = ref.bodies()[idx]
        em.set_bodies(eb)
        ref.L.ref_clear_warm(ref.h)
        em.sf[warm_row0:] = 0
        snap = ref.L.ref_clone(ref.h)
        ref.step(a)
        em.run([a], nl=nl)
        want = ref.bodies()[idx][:, :3]
        e = np.abs(em.bodies()[0, 1:, :3] - want)[mask].max()
        if e >= 1e-9:
            # a step the reference dynamics amplify (zero-length pins at an action change)?  Then the oracle's own
            # clones, poses perturbed by 1e-13, part as far on this very step; an algorithmic mismatch has no such excuse
            # (perturbations from one ulp up: the unperturbed oracle has EXACT coincidences -- a finger root's pin of length exactly
            # zero exerts nothing, one of length 1e-17 pulls in a round-off direction -- that a 1e-13 clone never has)
            spread, rs = 0.0, np.random.RandomState(t)
            for q in range(48):
                eps = (1e-16, 1e-15, 1e-14, 1e-13)[q % 4]
                h = ref.L.ref_clone(snap)
                buf = np.zeros((ref.L.ref_nbodies(h), 9))
                ref.L.ref_get_bodies(h, buf.ctypes.data_as(C.POINTER(C.c_double)))
                buf[idx, :3] += rs.uniform(-eps, eps, (len(idx), 3))
                ref.L.ref_set_bodies(h, buf.ctypes.data_as(C.POINTER(C.c_double)))
                ref.L.ref_step(h, int(a), 8.0)
                ref.L.ref_get_bodies(h, buf.ctypes.data_as(C.POINTER(C.c_double)))
                spread = max(spread, np.abs(buf[idx][:, :3] - want)[mask].max())
                ref.L.ref_free(h)
            assert e <= 100 * spread and e < 5e-2, (task, t, e, spread)
            e = 0.0
        ref.L.ref_free(snap)
        errs.append(e)
    errs = np.asarray(errs)
    assert np.median(errs) < 1e-13 and errs.max() < 1e-9, errs
    assert em.si[2, 0] == 0      # no contact-cache / overlap-list overflow


def test_f64_free_running_with_contacts():
    """Free run while the robot shoves the block into the wall: contacts, warm starting and the cache are exercised
    continuously; agreement stays at round-off for the first env-steps (the dynamics then amplify it)."""
    ref, em = _pair('MoveToCorner', 'f64')
    idx, mask = ref_body_index(ref), comparable_mask(ref)
    # face the block and push
    seen_contacts = 0
    for t, a in enumerate([3, 3, 3, 1, 1, 1, 1, 10, 10, 10, 10, 10]):
        ref.step(a)
        em.run([a], nl=16)
        seen_contacts += len(ref.contacts())
        d = np.abs(em.bodies()[0, 1:, :3] - ref.bodies()[idx][:, :3])[mask].max()
        assert d < (1e-12 if t < 4 else 1e-6), (t, d)
    assert seen_contacts > 0


def test_contacts_match_oracle_when_pushing():
    """Contact normals / points / accumulated impulses of the SAT narrowphase + cache equal the oracle's GJK/EPA ones."""
    n_checked = 0
    # free-running comparison of the contact set on a deterministic push
    ref, em = _pair('MoveToCorner', 'f64')
    for a in [2] * 6 + [7] * 8 + [1] * 25:
        ref.set_action(a)
        for _ in range(10):
            ref.substep()
            em.run([a], n_sub=1, nl=16, count_step=False)
            rc, ec = ref.contacts(), em.contacts()
            npts = int(sum(r[4] for r in rc))
            assert npts == len(ec)
            if npts:
                pts_ref = np.array([r[5 + 7 * k: 5 + 7 * k + 2] for r in rc for k in range(int(r[4]))])
                assert np.abs(pts_ref - ec[:, 4:6]).max() < 1e-4
                n_checked += npts
    assert n_checked > 20


@pytest.mark.parametrize('mode,p99_max,abs_max', [('mixed', 1e-5, 5e-3), ('f32', 5e-3, 5e-2)])
def test_reduced_precision_one_step_error(mode, p99_max, abs_max):
    """The shipped arithmetic (fp32 motion + fp64 poses) vs an all-fp32 build, teacher-forced one-step pose error."""
    ref, em = _pair('MoveToCorner', mode)
    idx, mask = ref_body_index(ref), comparable_mask(ref)
    rng = np.random.RandomState(0)
    errs = []
    for t in range(80):
        a = rng.randint(18)
        eb = em.bodies()
        eb[0, 1:, :] = ref.bodies()[idx]
        em.set_bodies(eb)
        ref.step(a)
        em.run([a], nl=16)
        errs.append(np.abs(em.bodies()[0, 1:, :3] - ref.bodies()[idx][:, :3])[mask].max())
    errs = np.array(errs)
    assert np.percentile(errs, 99) < p99_max and errs.max() < abs_max
    if mode == 'mixed':
        assert np.median(errs) < 1e-7


@pytest.mark.parametrize('task', TASKS)
def test_raster_logic_bit_exact(task):
    """Tile classification + pixel classification + coverage masks + line blending == the oracle's brute-force
    384x384 painter + 4x4 box filter, bit for bit, ego and allo."""
    ref, em = _pair(task, 'f64')
    idx = ref_body_index(ref)
    rng = np.random.RandomState(7)
    for rep in range(3):
        for _ in range(7):
            ref.step(rng.randint(18))
        eb = em.bodies()
        eb[0, 1:, :] = ref.bodies()[idx]
        em.set_bodies(eb)
        for view in ('ego', 'allo'):
            assert np.array_equal(em.render(0, view), ref.render_lores(view)), (task, rep, view)
    assert np.array_equal(em.render(0, 'ego', native=True), ref.render('ego'))


def test_episode_counter_and_done_flags():
    ref, em = _pair('MoveToRegion', 'mixed', n=3)
    for t in range(40):
        done = em.run([0, 1, 2], nl=16)
        assert done.all() == (t == 39) and done.any() == (t == 39)
    assert list(em.si[0]) == [40, 40, 40]
    em.reset(mask=[1, 0, 1])
    assert list(em.si[0]) == [0, 40, 0]


@pytest.mark.parametrize('task', TASKS)
def test_shipped_precision_drift_within_perturbation_envelope(task):
    """The CPU twin of tests/test_gpu_parity.py::test_f32_drift_within_perturbation_envelope: the kernel phases compiled for the
    host in the shipped precision (fp32 velocities / impulses, fp64 poses), free running against the oracle, next to the oracle's
    own spread under a 1e-7 pose perturbation (two replicas per env).  Gate: at env-steps 1, 5, 20, 80 the drift quantiles
    (median, p90 over the action tapes) stay within 2x the replicas'; at step 1 the median stays within 10x that of a replica
    that only stores its velocities in fp32."""
    from tests.util import EPS_F32, OracleEnvelope, masked_err, quantiles, velround_step
    n, steps = 32, (1, 5, 20, 80)
    tape = np.random.RandomState(7).randint(0, 18, size=(steps[-1], n)).astype(np.int32)
    orc = OracleEnvelope([lambda: new_ref(task)] * n, K=2, eps=EPS_F32, seed=2)
    em = EmuBatch(ref_entities_as_tuples(orc.base[0]), 1000, n, mode='mixed')
    em.reset()
    vround = [new_ref(task) for _ in range(n)]
    for s in range(steps[-1]):
        em.run(tape[s], nl=16)
        got = em.bodies()[:, 1:, :3]
        want, _ = orc.step(tape[s])
        if s + 1 in steps:
            (m, p), (em_, ep) = quantiles(np.array([masked_err(got[k], want[k], orc.mask) for k in range(n)])), quantiles(orc.all)
            assert m <= 2 * em_ and p <= 2 * ep, (task, s + 1, m, p, em_, ep)
        if s == 0:
            for k, r in enumerate(vround):
                velround_step(r, tape[0, k])
            v1 = np.median([masked_err(r.bodies()[orc.idx][:, :3], want[k], orc.mask) for k, r in enumerate(vround)])
            assert m <= 10 * v1, (task, m, v1)
    assert int(em.si[2].sum()) == 0


@pytest.mark.parametrize('mode', ['mixed', 'f64'])
@pytest.mark.parametrize('task', ['MoveToCorner', 'MatchRegions', 'ClusterColour'])
def test_contact_split_and_register_narrowphase_change_no_bit(task, mode):
    """k_step's two round-4 restructurings are claimed to evaluate the expressions they replace, in their order: the contact point's
    velocity and bias chains on two lanes (MGX_CONTACT_SPLIT) and the SAT loops on a shape's vertices in registers (MGX_NARROW_REGS).
    With contraction off (the emulation's build) the phase code with and without them must then produce the SAME BITS: 16 envs, 60
    env-steps of driving into things, every persistent state row compared."""
    ref = new_ref(task)
    ents = ref_entities_as_tuples(ref)
    rng = np.random.RandomState(zlib.crc32(task.encode()) % 1000)
    tape = rng.randint(0, 18, size=(60, 16)).astype(np.int32)
    tape[::3] = 1                                           # forward: contacts with blocks and walls
    runs = []
    for defines in ((), ('MGX_CONTACT_SPLIT=0',), ('MGX_NARROW_REGS=0',), ('MGX_CONTACT_SPLIT=0', 'MGX_NARROW_REGS=0')):
        em = EmuBatch(ents, ref.max_episode_steps, 16, mode=mode, defines=defines)
        em.reset()
        touched = 0
        for t in range(60):
            em.run(tape[t])
            touched += len(em.contacts())
        runs.append((em.sp.copy(), em.sf.copy(), em.si.copy(), touched))
    assert runs[0][3] > 50, 'the tape must produce contacts'
    for other in runs[1:]:
        assert other[3] == runs[0][3]
        for a, b in zip(runs[0][:3], other[:3]):
            assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize('mode', ['mixed', 'f64'])
@pytest.mark.parametrize('task', ['MoveToCorner', 'MatchRegions', 'ClusterColour', 'FindDupe'])
def test_sort_and_sweep_broadphase_changes_no_bit(task, mode):
    """The sort-and-sweep broadphase (MGX_BROAD_SAP builds, mgx_sim.h: rank the AABB min-x keys, sweep with early exit, emit in ascending
    candidate-pair order) must leave the list the candidate-pair form leaves, entry for entry -- the arbiter order, and with it every bit
    of the state, depends on it: 16 envs, 60 env-steps of driving into things, every persistent state row compared (also with the manifold
    slots of the crowded worlds handed out in every world: MGX_MANIFOLD_SLOTS_ALL)."""
    ref = new_ref(task)
    ents = ref_entities_as_tuples(ref)
    rng = np.random.RandomState(zlib.crc32(task.encode()) % 1000)
    tape = rng.randint(0, 18, size=(60, 16)).astype(np.int32)
    tape[::3] = 1
    runs = []
    for defines in ((), ('MGX_BROAD_SAP=1',), ('MGX_MANIFOLD_SLOTS_ALL=1',), ('MGX_BROAD_SAP=1', 'MGX_MANIFOLD_SLOTS_ALL=1')):
        em = EmuBatch(ents, ref.max_episode_steps, 16, mode=mode, defines=defines)
        em.reset()
        touched = 0
        for t in range(60):
            em.run(tape[t])
            touched += len(em.contacts())
        runs.append((em.sp.copy(), em.sf.copy(), em.si.copy(), touched))
    assert runs[0][3] > 15, 'the tape must produce contacts'
    for other in runs[1:]:
        assert other[3] == runs[0][3]
        for a, b in zip(runs[0][:3], other[:3]):
            assert a.tobytes() == b.tobytes()
