"""Shared helpers for the parity tests: oracle <-> product body index mapping."""
import numpy as np

from oracle.entities_ref import Robot as RefRobot, Shape as RefShape
from oracle.env_ref import RefEnv

TASKS = ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']


def ref_body_index(ref_env):
    """Oracle body indices of the non-static bodies, in creation order == product bodies 1..n."""
    idx = []
    for e in ref_env.world.entities:
        if isinstance(e, (RefRobot, RefShape)):
            idx += list(e.bodies)
    return idx


def comparable_mask(ref_env):
    """bool[n_dyn, 3] of pose components that are persistent in the product (control body and eye positions are not)."""
    idx = ref_body_index(ref_env)
    mask = np.ones((len(idx), 3), dtype=bool)
    rb = ref_env.task.robot.bodies          # robot, control, eye, eye, finger, finger
    mask[idx.index(rb[1])] = False
    for eye in rb[2:4]:
        mask[idx.index(eye), :2] = False
    return mask


def ref_entities_as_tuples(ref_env):
    """Entity list of an oracle env in the emulator's tuple format."""
    from oracle.entities_ref import GoalRegion
    st = {'triangle': 0, 'square': 1, 'pentagon': 2, 'hexagon': 3, 'octagon': 4, 'circle': 5, 'star': 6}
    co = {'red': 0, 'green': 1, 'blue': 2, 'yellow': 3}
    out = []
    for e in ref_env.world.entities:
        if isinstance(e, RefRobot):
            out.append(('robot', e.init_pos[0], e.init_pos[1], e.init_angle))
        elif isinstance(e, RefShape):
            out.append(('shape', st[e.shape_type], co[e.colour_name], e.init_pos[0], e.init_pos[1], e.init_angle))
        elif isinstance(e, GoalRegion):
            out.append(('goal', e.x, e.y, e.h, e.w, co[e.colour_name]))
    return out


def new_ref(task):
    e = RefEnv(task)
    e.reset()
    return e


# ---------------------------------------------------------------------------------------------------------------------
# Perturbation envelopes: how far the ORACLE moves from itself under a perturbation the size of the engine's rounding.
# The reference dynamics are chaotic by construction (zero-length PinJoints take their direction from a round-off-level
# vector), so "pose error vs the oracle" only means something next to the oracle's own spread.
EPS_F32 = 1e-7      # one fp32 rounding of a unit-scale quantity (2^-24 = 6e-8), the shipped engine's resolution
EPS_F64 = 1e-13     # a few hundred fp64 roundings: what an all-fp64 engine differs from the oracle by (libm, FMA contraction)
# Per-env bounds for the shipped fp32 build use replicas that are perturbed by EPS_F32 AND store their velocities in fp32
# (one rounding per substep); the engine rounds each of the ~100 operations a velocity goes through per substep (10
# iterations x ~10 constraint rows), a random walk ~sqrt(100) = 10 times wider: that factor, not a tuned constant.
F32_OPS_FACTOR = 10.0


# An envelope is a gate only between a floor and a cap (round-2 advisor).  Below FLOOR the replicas did not move apart at all
# (nothing touches, steady motion) and a last-bit difference of a correct engine must still pass; above ENVELOPE_CAP the
# oracle disagrees with itself by a fortieth of the arena and "within the envelope" would accept anything: such samples are
# left UNDECIDED -- not passed -- and every test that uses the gate asserts that most of its samples were decided.
ENVELOPE_CAP = 5e-2
FLOOR_F64, FLOOR_F32 = 1e-12, 1e-6


class EnvelopeTally:
    def __init__(self):
        self.decided, self.total = 0, 0

    def check(self, errs, running, factor, floor, what=()):
        errs, raw = np.asarray(errs, dtype=np.float64), factor * np.asarray(running, dtype=np.float64)
        decided = raw <= ENVELOPE_CAP
        bound = np.maximum(raw, floor)
        self.decided += int(decided.sum()); self.total += int(decided.size)
        bad = decided & ~(errs <= bound)
        assert not bad.any(), (what, 'pose error above the oracle\'s own spread', errs, bound, np.nonzero(bad)[0])

    def assert_mostly_decided(self, fraction=0.5, what=()):
        assert self.total > 0 and self.decided >= fraction * self.total, (what, f'only {self.decided} of {self.total} samples had a meaningful envelope')


def perturb_bodies(ref_env, eps, rs):
    """x, y, angle of every non-static body of an oracle env += U(-eps, eps), independently."""
    b = ref_env.bodies()
    for k in ref_body_index(ref_env):
        b[k, :3] += rs.uniform(-eps, eps, 3)
    ref_env.set_bodies(b)


def masked_err(a, b, mask):
    return float(np.abs(a - b)[mask].max())


class OracleEnvelope:
    """n oracle envs + K perturbed replicas of each.  step(actions) advances them and returns
    (poses[n, n_dyn, 3] of the unperturbed envs, spread[n]) with spread[k] = the largest pose deviation of env k's replicas
    from env k after this step; `running` keeps the maximum over the steps of the episode (a bound that only widens).
    `factories` build an env ready to step (already reset); `base` = the caller's own unperturbed envs (then the caller
    steps them, and step() only reads their poses).  fp32_state: the replicas also keep their velocity state in fp32
    (velround_step): the shipped engine's storage format, i.e. rounding noise that keeps coming, not just an initial offset."""

    def __init__(self, factories, K, eps, seed=0, base=None, fp32_state=False):
        self.own_base, self.fp32_state = base is None, fp32_state
        self.base = [f() for f in factories] if base is None else list(base)
        self.rs, self.eps = np.random.RandomState(seed), eps
        self.reps = [[f() for _ in range(K)] for f in factories]
        for row in self.reps:
            for q in row:
                perturb_bodies(q, eps, self.rs)
        self.idx, self.mask = ref_body_index(self.base[0]), comparable_mask(self.base[0])
        self.running = np.zeros(len(self.base))

    def reset(self):
        """Next episode: every replica resets (drawing what its base env draws) and is perturbed afresh."""
        for row in self.reps:
            for q in row:
                q.reset()
                perturb_bodies(q, self.eps, self.rs)
        self.idx, self.mask = ref_body_index(self.base[0]), comparable_mask(self.base[0])
        self.running[:] = 0

    def step(self, actions):
        want, now, self.all = [], np.zeros(len(self.base)), []
        for k, r in enumerate(self.base):
            if self.own_base:
                r.step(actions[k])
            idx = ref_body_index(r)
            w = r.bodies()[idx][:, :3]
            want.append(w)
            for q in self.reps[k]:
                if self.fp32_state:
                    velround_step(q, actions[k])
                else:
                    q.step(actions[k])
                self.all.append(masked_err(q.bodies()[idx][:, :3], w, comparable_mask(r)))
                now[k] = max(now[k], self.all[-1])
        self.running = np.maximum(self.running, now)
        self.all = np.array(self.all)           # every replica's deviation after this step (pooled quantiles)
        return want, now


def velround_step(ref_env, action):
    """One env-step of the oracle with its velocity state rounded to fp32 after every substep: the reference dynamics under
    the shipped engine's STORAGE precision alone (the engine also rounds every operation in between)."""
    ref_env.set_action(action)
    for _ in range(10):
        ref_env.substep()
        b = ref_env.bodies()
        b[:, 3:] = b[:, 3:].astype(np.float32)
        ref_env.set_bodies(b)


def quantiles(x):
    return float(np.median(x)), float(np.percentile(x, 90))
