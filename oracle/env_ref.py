"""Single-env restatement of BaseEnv (magical/base_env.py:60-343) + the LoRes4E
preprocessor (magical/benchmarks/__init__.py:80-136,219-256).

TEST INFRASTRUCTURE: used by tests/, smoke() and bench.py's cpu_baseline leg as
the checker.  Never imported by the product package.
"""
import collections
import ctypes as C

import numpy as np

from ._lib import lib
from .entities_ref import PhysVars, ArenaBoundaries, RefWorld
from .tasks_ref import TASKS

DEFAULT_RES = 384      # benchmarks/__init__.py:23
FPS = 8                # benchmarks/__init__.py:401-404
PHYS_ITER = 10


class RefEnv:
    def __init__(self, task, max_episode_steps=None, gjk_warm=True, rand_dynamics=False, seed=None, backend=None, **task_flags):
        self.task_cls = TASKS[task]
        self.max_episode_steps = max_episode_steps or self.task_cls.ep_len
        self.backend = backend                            # None: the C restatement; 'pymunk': oracle/pymunk_backend.py
        self.world = None
        self.task = None
        self.gjk_warm = gjk_warm
        self.rand_dynamics = rand_dynamics
        self.task_flags = task_flags                      # rand_* flags of the task constructor
        self.rng = np.random.RandomState(seed=seed)       # base_env.py:133-140

    # base_env.py:177-234
    def reset(self):
        # base_env.py:198-203: the physics variables are drawn before anything else touches the rng
        pv = PhysVars.sample(self.rng) if self.rand_dynamics else None
        self.world = RefWorld(phys_vars=pv, phys_iter=PHYS_ITER, backend=self.backend)
        self.L.ref_set_gjk_warm(self.world.h, 1 if self.gjk_warm else 0)
        self.arena = self.world.add(ArenaBoundaries())
        self.task = self.task_cls(self.world, rng=self.rng, **self.task_flags)
        if 'poses' in self.task.choices:
            # poses were drawn on that world; the episode runs in a fresh one built at them (placement_ref.py)
            choices = self.task.choices
            self.world = RefWorld(phys_vars=pv, phys_iter=PHYS_ITER, backend=self.backend)
            self.L.ref_set_gjk_warm(self.world.h, 1 if self.gjk_warm else 0)
            self.arena = self.world.add(ArenaBoundaries())
            self.task = self.task_cls(self.world, rng=None, replay=choices, **self.task_flags)
        self._episode_steps = 0
        return None

    @property
    def h(self):
        return self.world.h

    @property
    def L(self):
        return self.world.L if self.world is not None else lib()

    # base_env.py:255-292 (without the render)
    def step(self, action):
        self.L.ref_step(self.h, int(action), float(FPS))
        self._episode_steps += 1
        done = self._episode_steps >= self.max_episode_steps
        eval_score = 0.0
        if done:
            eval_score = float(self.task.score_on_end_of_traj())
            assert 0 <= eval_score <= 1
        return 0.0, done, {'eval_score': eval_score}

    def substep(self, dt=1.0 / FPS / 10):
        self.L.ref_substep(self.h, dt)

    def set_action(self, action):
        self.L.ref_set_action(self.h, int(action))

    def bodies(self):
        n = self.L.ref_nbodies(self.h)
        buf = np.zeros((n, 9), dtype=np.float64)
        self.L.ref_get_bodies(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)))
        return buf

    def set_bodies(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self.L.ref_set_bodies(self.h, arr.ctypes.data_as(C.POINTER(C.c_double)))

    def body_mass(self):
        n = self.L.ref_nbodies(self.h)
        buf = np.zeros((n, 2), dtype=np.float64)
        self.L.ref_get_body_mass(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)))
        return buf

    def joint_acc(self):
        n = self.L.ref_njoints(self.h)
        buf = np.zeros((n, 2), dtype=np.float64)
        self.L.ref_get_joint_acc(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)))
        return buf

    def contacts(self, max_rows=128):
        buf = np.zeros((max_rows, 19), dtype=np.float64)
        n = self.L.ref_get_contacts(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)), max_rows)
        return buf[:n]

    # base_env.py:309-338 -- one view at native resolution
    def render(self, view='ego', res=DEFAULT_RES):
        out = np.zeros((res, res, 3), dtype=np.uint8)
        self.L.ref_render(self.h, 0 if view == 'ego' else 1, res, out.ctypes.data)
        return out

    def render_lores(self, view='ego', small=96):
        """gym ResizeObservation: cv2.resize(frame, (96,96), INTER_AREA)."""
        full = self.render(view, DEFAULT_RES)
        return area_downsample(full, DEFAULT_RES // small)


def area_downsample(img, factor):
    img = np.ascontiguousarray(img)
    h, w, c = img.shape
    out = np.zeros((h // factor, w // factor, c), dtype=np.uint8)
    lib().ref_area_downsample(img.ctypes.data, h, factor, c, out.ctypes.data)
    return out


class LoRes4ERef:
    """FlattenFrameStack(allo=0, ego=4) then ResizeObservation(96)
    (benchmarks/__init__.py:219-256): obs = u8[96,96,12], oldest frame first."""

    def __init__(self, env):
        self.env = env
        self.frames = collections.deque(maxlen=4)

    def reset(self):
        self.env.reset()
        frame = self.env.render('ego')
        for _ in range(4):
            self.frames.append(frame)
        return self._obs()

    def step(self, action):
        rew, done, info = self.env.step(action)
        self.frames.append(self.env.render('ego'))
        return self._obs(), rew, done, info

    def _obs(self):
        stacked = np.concatenate(list(self.frames), axis=-1)
        return area_downsample(stacked, DEFAULT_RES // 96)


class LoRes3EARef:
    """FlattenFrameStack(OrderedDict(allo=1, ego=3)) then ResizeObservation(96)
    (benchmarks/__init__.py:219-251, :80-136): channels = [allo_t, ego_t-2, ego_t-1, ego_t]."""

    def __init__(self, env):
        self.env = env
        self.allo = collections.deque(maxlen=1)
        self.ego = collections.deque(maxlen=3)

    def reset(self):
        self.env.reset()
        self.allo.append(self.env.render('allo'))
        ego = self.env.render('ego')
        for _ in range(3):
            self.ego.append(ego)
        return self._obs()

    def step(self, action):
        rew, done, info = self.env.step(action)
        self.allo.append(self.env.render('allo'))
        self.ego.append(self.env.render('ego'))
        return self._obs(), rew, done, info

    def _obs(self):
        stacked = np.concatenate(list(self.allo) + list(self.ego), axis=-1)
        return area_downsample(stacked, DEFAULT_RES // 96)


class LoResStackRef:
    """ResizeDictObservation(96) then EagerDictFrameStack(4) (benchmarks/__init__.py:46-77,139-169,204-215):
    obs = {'allo': u8[96,96,12], 'ego': u8[96,96,12]}, oldest frame first in each."""

    def __init__(self, env):
        self.env = env
        self.frames = collections.deque(maxlen=4)

    def _frame(self):
        f = DEFAULT_RES // 96
        return {'allo': area_downsample(self.env.render('allo'), f), 'ego': area_downsample(self.env.render('ego'), f)}

    def reset(self):
        self.env.reset()
        frame = self._frame()
        for _ in range(4):
            self.frames.append(frame)
        return self._obs()

    def step(self, action):
        rew, done, info = self.env.step(action)
        self.frames.append(self._frame())
        return self._obs(), rew, done, info

    def _obs(self):
        return {k: np.concatenate([fr[k] for fr in self.frames], axis=-1) for k in ('allo', 'ego')}
