"""CPU probe (round 4): why teacher forcing of the BODIES alone left "offenders" in the fp64 one-step comparison.
The host build of the kernel phases (tests/emu, fp64) against the oracle on the GPU test's script (chase a block, every fourth env
squeezing one against a wall), every env-step started from the oracle's body state; next to every sample the oracle's own clones:
poses perturbed by 1e-13, and warm-start impulses scaled by 1 +- max(1e-12, the env's previous error) (ref_perturb_warm).

    python tools/dev/warm_start_probe.py [Task] [--cold]

--cold: FULL-state forcing (accumulated impulses zeroed on both sides at every env-step boundary: ref_clear_warm / the impulse rows of
the motion blob) -- the form tests/test_gpu_parity.py::test_f64_engine_one_step_equivalence_and_contact_coverage now uses.  Without it
the probe lists the offenders (runs inside one env, same action, clones calm); with it none is left."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.emu.emu import EmuBatch
from tests.util import new_ref, ref_entities_as_tuples, ref_body_index, comparable_mask, masked_err, EPS_F64
from oracle.env_ref import FPS
src = open(os.path.join(ROOT, 'tests', 'test_gpu_parity.py')).read()
ns = {}
exec('import numpy as np\n' + src[src.index('def _chase_action'):src.index('def _live_arbiters')], ns)
exec(src[src.index('def _pin_block_against_wall'):src.index("@pytest.mark.parametrize('task', TASKS)\ndef test_f64_engine_one_step_equivalence")], ns)
args = [a for a in sys.argv[1:] if not a.startswith('--')]
task = args[0] if args else 'MoveToCorner'
COLD = '--cold' in sys.argv
n, t = 32, 60
refs = [new_ref(task) for _ in range(n)]
em = EmuBatch(ref_entities_as_tuples(refs[0]), 1000, n, 'f64'); em.reset()
idx, mask = ref_body_index(refs[0]), comparable_mask(refs[0])
L = refs[0].L
pinned = [k % 4 == 3 and ns['_pin_block_against_wall'](r, k // 4) for k, r in enumerate(refs)]
rs = np.random.RandomState(23)
def clones(r, action, pose_eps, warm_rel, K=3):
    out = []
    for q in range(K):
        h = L.ref_clone(r.h)
        buf = np.zeros((L.ref_nbodies(h), 9)); L.ref_get_bodies(h, buf.ctypes.data_as(C.POINTER(C.c_double)))
        buf[idx, :3] += rs.uniform(-pose_eps, pose_eps, (len(idx), 3))
        L.ref_set_bodies(h, buf.ctypes.data_as(C.POINTER(C.c_double)))
        if warm_rel: L.ref_perturb_warm(h, warm_rel, int(rs.randint(1 << 30)))
        L.ref_step(h, int(action), float(FPS))
        buf2 = np.zeros_like(buf); L.ref_get_bodies(h, buf2.ctypes.data_as(C.POINTER(C.c_double))); L.ref_free(h)
        out.append(buf2[idx][:, :3])
    return out
rows = []
prev_err = np.zeros(n)
for s in range(t):
    eb = em.bodies()
    for k, r in enumerate(refs):
        eb[k, 1:, :] = r.bodies()[idx]
        if COLD: L.ref_clear_warm(r.h)
    em.set_bodies(eb)
    nv = max((m >> 12) for m in em.rows if (m & 15) >= 3) + 1
    if COLD: em.sf[nv + 5:] = 0
    acts = np.array([(9 if (s // 5) % 2 else 0) + 1 if pinned[k] else ns['_chase_action'](r, k, s) for k, r in enumerate(refs)], dtype=np.int32)
    cl_pose = [clones(r, acts[k], EPS_F64, 0.0) for k, r in enumerate(refs)]
    cl_warm = [clones(r, acts[k], EPS_F64, max(1e-12, prev_err[k])) for k, r in enumerate(refs)]
    em.run(acts)
    got = em.bodies()[:, 1:, :3]
    for k, r in enumerate(refs):
        r.step(acts[k]); want = r.bodies()[idx][:, :3]
        e = masked_err(got[k], want, mask)
        sp = max(masked_err(c, want, mask) for c in cl_pose[k]); sw = max(masked_err(c, want, mask) for c in cl_warm[k])
        rows.append((s, k, e, sp, sw, len(r.contacts()))); prev_err[k] = e
rows = np.array(rows)
e, sp, sw = rows[:, 2], rows[:, 3], rows[:, 4]
print(task, 'median', np.median(e), 'p99', np.percentile(e, 99), 'max', e.max())
off = rows[e >= 1e-9]
print('offenders', len(off))
for r in off: print('  step %d env %d err %.1e pose-spread %.1e warm-spread %.1e arb %d' % (r[0], r[1], r[2], r[3], r[4], r[5]))
bad = rows[(e > 1e-9) & (e > 100 * np.maximum(sp, sw))]
for r in bad: print('  UNEXPLAINED step %d env %d err %.1e pose-spread %.1e warm-spread %.1e arb %d' % (r[0], r[1], r[2], r[3], r[4], r[5]))
print('unexplained by pose spread (x1e3):', int(((e > 1e-9) & (e > 1e3 * sp)).sum()), ' by pose+warm (x1e3):', int(((e > 1e-9) & (e > 1e3 * np.maximum(sp, sw))).sum()))
