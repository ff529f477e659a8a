"""Batched mirror of magical/base_env.py: one object steps N independent envs on one MI355X.

`BaseEnv` keeps the reference's constructor keywords, constants, `reset()/step()/render()/
seed()/close()` protocol, `action_to_flags/flags_to_action`, the `on_reset()` /
`score_on_end_of_traj()` task hooks and `add_entities()`; what used to be one pymunk Space +
one pyglet Viewer per env is one native engine handle (include/mgx.h) per GPU.

step() returns (obs, reward, done, info) like the reference (base_env.py:255-292), batched:
  obs    : torch tensor on the engine's device (layout depends on the preprocessor, see
           magical_amd/benchmarks/__init__.py)
  reward : torch.float32[N], always 0 (base_env.py:266-267)
  done   : numpy bool[N]
  info   : {'eval_score': numpy float64[N]}  (0.0 until done, base_env.py:285-288); with terminal_observation=True, in steps in
           which episodes end, also 'terminal_observation' (the finished envs' last observation) and 'terminal_env_idx'
With auto_reset=True (default, SB3 VecEnv semantics) finished envs are reset inside step()
and their returned obs is the first observation of the next episode (the terminal observation
is overwritten; a caller that needs it steps with auto_reset=False and resets itself).

OWNERSHIP OF OBSERVATIONS.  The reference returns a fresh numpy array per step; here the frame
stack is ONE persistent device tensor that the raster kernel shifts in place, and step() /
reset() return that same tensor every time (LoResCHW4E: a permuted view of it).  A collector
that keeps references (`buf.append(obs)`) ends up with T aliases of the newest stack: copy
what you keep (`obs.clone()`), or construct the env with `copy_obs=True`, which hands out a
clone per call (one extra 110 KB/env device copy per step).
"""
import abc
import ctypes as C
import os

import numpy as np

from . import _native as nat
from . import entities as en
from . import spaces


class PhysicsVariables:
    """base_env.py:49-57 (defaults + the uniform ranges used by rand_dynamics)."""
    robot_pos_joint_max_force = (3, (2.2, 3.5))
    robot_rot_joint_max_force = (1, (0.7, 1.5))
    robot_finger_max_force = (4, (2.5, 4.5))
    shape_trans_joint_max_force = (1.5, (1.0, 1.8))
    shape_rot_joint_max_force = (0.1, (0.07, 0.15))
    NAMES = ('robot_pos_joint_max_force', 'robot_rot_joint_max_force', 'robot_finger_max_force',
             'shape_trans_joint_max_force', 'shape_rot_joint_max_force')

    @classmethod
    def defaults(cls):
        return [float(getattr(cls, n)[0]) for n in cls.NAMES]

    @classmethod
    def sample_batch(cls, brng):
        """sample() for every env of a BatchRng: float64[m, 5], the same arithmetic on the same draws."""
        u = brng.random_sample(len(cls.NAMES))
        return np.stack([lo + (hi - lo) * u[:, i] for i, (lo, hi) in enumerate(getattr(cls, n)[1] for n in cls.NAMES)], axis=1)

    @classmethod
    def sample(cls, rng):
        # rng.uniform(lo, hi) per variable, in order = lo + (hi - lo) * random_sample(): one call for the five doubles
        u = rng.random_sample(len(cls.NAMES))
        return [float(lo + (hi - lo) * x) for x, (lo, hi) in zip(u, (getattr(cls, n)[1] for n in cls.NAMES))]


class BaseEnv(abc.ABC):
    # constants for all envs (base_env.py:61-76)
    ROBOT_RAD = 0.2
    ROBOT_MASS = 1.0
    SHAPE_RAD = ROBOT_RAD * 0.6
    ARENA_BOUNDS_LRBT = [-1, 1, -1, 1]
    ARENA_SIZE_MAX = max(ARENA_BOUNDS_LRBT)
    # minor-jitter bounds of the Test*Jitter variants (base_env.py:68-76)
    RAND_GOAL_MIN_SIZE = 0.5
    RAND_GOAL_MAX_SIZE = 0.8
    RAND_GOAL_SIZE_RANGE = RAND_GOAL_MAX_SIZE - RAND_GOAL_MIN_SIZE
    JITTER_PCT = 0.05
    JITTER_POS_BOUND = ARENA_SIZE_MAX * JITTER_PCT / 2.0
    JITTER_ROT_BOUND = JITTER_PCT * np.pi
    JITTER_TARGET_BOUND = JITTER_PCT * RAND_GOAL_SIZE_RANGE / 2

    # True in tasks whose episodes differ in the blocks' shape types or in the number of entities (Test*Shape /
    # TestCountPlus / TestAll): on_reset() then lists every entity an episode can have, sample_variation() says which
    # are present ('enabled') and of what type ('shape_types'), and the engine keeps one world per env
    variable_worlds = False
    # False in the tasks whose score is a function of the goal regions' overlap sets alone (MoveToRegion, MatchRegions, FindDupe,
    # FixColour): step() then hands score_on_end_of_traj() no poses -- the sets come from the device (k_score,
    # mgx_engine_score_overlaps), one byte per region x entity x env instead of the pose rows
    score_needs_poses = True

    def __init__(self, *, n_envs=1, device='cuda:0', res_hw=(384, 384), fps=8, phys_steps=10, phys_iter=10,
                 max_episode_steps=None, rand_dynamics=False, ego_view=True, allo_view=True,
                 dtype='f32', lanes_per_env=0, auto_reset=True, copy_obs=False, strict_capacity=False, overlap=True, batch_draws=True,
                 obs_ring=None, terminal_observation=False, device_scores=True):
        import torch
        if fps != 8 or phys_steps != 10 or phys_iter != 10:
            raise NotImplementedError('the engine is built for the registered rates: fps=8, 10 substeps, 10 iterations '
                                      '(benchmarks/__init__.py:401-404)')
        assert ego_view or allo_view, 'must use egocentric view or allocentric view (or both)'
        self.n_envs, self.fps, self.phys_steps, self.phys_iter = int(n_envs), fps, phys_steps, phys_iter
        self.res_hw, self.max_episode_steps = tuple(res_hw), max_episode_steps
        self.ego_view, self.allo_view, self.rand_dynamics = ego_view, allo_view, rand_dynamics
        self.auto_reset, self.copy_obs, self.strict_capacity = auto_reset, bool(copy_obs), bool(strict_capacity)
        # the point tasks' episode-end scores on the device (mgx_engine_score_points) where that is bit-exact; False: always on the host
        self.device_scores = bool(device_scores) and not os.environ.get('MGX_HOST_SCORES')
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise nat.MgxError('magical_amd runs on an MI355X (torch device "cuda:N"); there is no CPU fallback')
        if self.device.index is None:      # 'cuda' = torch's current device, pinned now (the engine lives on one GPU)
            self.device = torch.device('cuda', torch.cuda.current_device())
        # LoResCHW4E only: keep the last `obs_ring` frames as channel planes u8[N, R, 3, 96, 96] and hand out windows of that ring
        # (benchmarks/preproc.py); None reads MGX_OBS_RING, 0 = the in-place 12-channel stack
        self.obs_ring = int(os.environ.get('MGX_OBS_RING', '0')) if obs_ring is None else int(obs_ring)
        if self.obs_ring and self.obs_ring < 5:
            raise ValueError('obs_ring: at least 5 frames (4 in the window + the one being written)')
        # auto-reset replaces a finished env's observation by the first one of its next episode; with terminal_observation=True
        # step() also hands out the observation the episode ENDED with (what the reference's step() returns together with done,
        # base_env.py:255-292; every MAGICAL episode ends by its time limit, so value bootstrapping wants it):
        # info['terminal_observation'] (rows in the order of info['terminal_env_idx']), at the cost of one more rasteriser pass
        # in the steps in which episodes end
        self.terminal_observation = bool(terminal_observation)
        if self.terminal_observation and self.obs_ring:
            raise NotImplementedError('terminal_observation with obs_ring: not built')
        self.batch_draws = bool(batch_draws)   # per-episode draws of all envs of a reset per native call (batch_rng.py) instead of env by env
        self.overlap = bool(overlap)       # step(): physics + observation as a producer / consumer kernel pair (mgx_engine_step_render)
        self._obs_ready = False
        self._fill_idx = None              # host-side index list of the device mask handed to _observe() (step() only)
        self.capacity_overflows = 0        # contacts / overlapping pairs the fixed-size working set dropped (see step())
        self.dtype_name = dtype
        self._dtype = {'f32': nat.MGX_F32, 'f64': nat.MGX_F64, 'f32_pure': nat.MGX_F32_PURE}[dtype]
        self._lanes = lanes_per_env
        self.action_space_n = len(en.ACTION_NUMS_FLAGS_NAMES)
        # per-env spaces, as a gym VecEnv reports them (base_env.py:97-109); `num_envs` envs run in lockstep
        self.num_envs = self.n_envs
        self.action_space = spaces.Discrete(self.action_space_n)
        self._lib = nat.lib()
        self._world = None
        self._engine = None
        self._entities = None
        self._robot = None
        self.seed()
        self._build()
        self.observation_space = self._observation_space()

    # ------------------------------------------------------------------ reference protocol
    def action_to_flags(self, int_action):
        return en.ACTION_ID_TO_FLAGS[int(int_action)]

    def flags_to_action(self, flags):
        return en.FLAGS_TO_ACTION_ID[tuple(flags)]

    def seed(self, seed=None):
        """base_env.py:133-140.  Every env of the batch has its own np.random.RandomState: env k is seeded with
        `seed + k` (what gym's VectorEnv.seed(int) does), so env 0 draws exactly what the reference env seeded with
        `seed` draws.  `self.rng` is env 0's."""
        if seed is None:
            seed = np.random.randint(0, (1 << 31) - 1)
        self.rngs = [np.random.RandomState(seed=(seed + k) % (1 << 32)) for k in range(self.n_envs)]
        self.rng = self.rngs[0]
        self._rng_addr_cache = None
        return [seed]

    def _rng_addrs(self, env_idx):
        """uint64[len(env_idx)]: where the MT19937 states of the envs' generators live (batch_rng.state_addresses), looked up once
        per generator: a RandomState keeps its state where it is for life (set_state copies into it).  The cache holds the
        generators it was made from, so an env whose generator was replaced (env.rngs[k] = ...) is seen and looked up again."""
        from .batch_rng import state_addresses
        c = getattr(self, '_rng_addr_cache', None)
        if c is None or len(c[0]) != len(self.rngs) or not all(a is b for a, b in zip(c[0], self.rngs)):
            c = self._rng_addr_cache = (tuple(self.rngs), state_addresses(self.rngs))
        return np.ascontiguousarray(c[1][np.asarray(env_idx, dtype=np.int64)])

    def _make_robot(self, init_pos, init_angle):
        return en.Robot(radius=self.ROBOT_RAD, init_pos=init_pos, init_angle=init_angle, mass=self.ROBOT_MASS)

    def _make_shape(self, **kwargs):
        return en.Shape(shape_size=self.SHAPE_RAD, **kwargs)

    @abc.abstractmethod
    def on_reset(self):
        """Create the task's entities and pass them to add_entities() in draw/joint order."""

    @abc.abstractmethod
    def score_on_end_of_traj(self, poses):
        """poses: float64[M, n_bodies, 3] of the M finished envs -> float64[M] in [0, 1]."""

    def add_entities(self, entities):
        L, w = self._lib, self._world
        for ent in entities:
            if isinstance(ent, en.Robot):
                self._robot = ent
                ent.ent_id = nat.check(L.mgx_world_add_robot(w, ent.init_pos[0], ent.init_pos[1], ent.init_angle))
            elif isinstance(ent, en.Shape):
                ent.ent_id = nat.check(L.mgx_world_add_shape(w, en.SHAPE_TYPE_ID[ent.shape_type], en.COLOUR_ID[ent.colour_name],
                                                             ent.init_pos[0], ent.init_pos[1], ent.init_angle))
            elif isinstance(ent, en.GoalRegion):
                ent.ent_id = nat.check(L.mgx_world_add_goal(w, ent.x, ent.y, ent.h, ent.w, en.COLOUR_ID[ent.colour_name]))
            else:
                raise TypeError(f"don't know how to handle entity {ent!r}")
            self._entities.append(ent)

    # ------------------------------------------------------------------ engine construction
    _warm = False

    def _build(self):
        import torch
        L = self._lib
        w = C.c_void_p()
        nat.check(L.mgx_world_create(C.byref(w)))
        self._world = w
        self._entities = []
        pv = (C.c_double * 5)(*PhysicsVariables.defaults())
        nat.check(L.mgx_world_set_phys_vars(w, pv))
        self.on_reset()      # arena is added by the library itself, first (base_env.py:213)
        assert isinstance(self._robot, en.Robot)
        nat.check(L.mgx_world_finalize(w, int(self.max_episode_steps or (1 << 30))))
        out = C.c_int()
        self.n_bodies = self._info('n_bodies')
        for ent in self._entities:
            kind, body = C.c_int(), C.c_int()
            nat.check(L.mgx_world_entity(w, ent.ent_id, C.byref(kind), C.byref(body), None, None))
            ent.body = body.value if body.value >= 0 else None
            if isinstance(ent, en.GoalRegion):
                bb = (C.c_double * 4)()
                nat.check(L.mgx_world_goal_bb(w, ent.ent_id, bb))
                ent.bb = tuple(bb)       # l b r t
        eng = C.c_void_p()
        # lanes per env: the caller's, else the engine's choice -- told (-1) when this env's steps are rendered by the fused env-step, where
        # the crowded worlds do better with narrow groups (include/mgx.h)
        lanes = self._lanes
        if lanes == 0 and self.overlap and type(self)._fused_target is not BaseEnv._fused_target:
            lanes = -1
        nat.check(L.mgx_engine_create(w, self.n_envs, self.device.index, self._dtype, lanes, C.byref(eng)))
        self._engine = eng
        ne = len(self._entities)
        self._default_shape_types = np.array([en.SHAPE_TYPE_ID[e.shape_type] if isinstance(e, en.Shape) else -1 for e in self._entities], dtype=np.int32)
        self.entity_shape_types = np.tile(self._default_shape_types, (self.n_envs, 1))          # per env
        self.entity_enabled = np.ones((self.n_envs, ne), dtype=bool)                              # per env
        if self.variable_worlds:
            # the largest world an episode can have sizes the engine's per-env tables: everything present, every block a star
            cap_types = np.where(self._default_shape_types >= 0, en.SHAPE_TYPE_ID[en.ShapeType.STAR], -1).astype(np.int32)
            cap = C.c_void_p()
            nat.check(L.mgx_world_variant(w, None, cap_types.ctypes.data_as(C.POINTER(C.c_int)), C.byref(cap)))
            try:
                nat.check(L.mgx_engine_enable_env_worlds(eng, cap))
            finally:
                L.mgx_world_destroy(cap)
        rp, rf, ri, szp, szf = (C.c_int() for _ in range(5))
        nat.check(L.mgx_engine_state_shape(eng, C.byref(rp), C.byref(rf), C.byref(ri), C.byref(szp), C.byref(szf)))
        tp = torch.float64 if szp.value == 8 else torch.float32
        tf = torch.float64 if szf.value == 8 else torch.float32
        self.state_p = torch.zeros((rp.value, self.n_envs), dtype=tp, device=self.device)
        self.state_f = torch.zeros((rf.value, self.n_envs), dtype=tf, device=self.device)
        self.state_i = torch.zeros((ri.value, self.n_envs), dtype=torch.int32, device=self.device)
        self._done_dev = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)
        self._reward = torch.zeros(self.n_envs, dtype=torch.float32, device=self.device)
        self._steps = np.zeros(self.n_envs, dtype=np.int64)
        self.phys_vars = np.tile(np.asarray(PhysicsVariables.defaults(), dtype=np.float64), (self.n_envs, 1))   # per env
        self._ent_colour = None      # device int32[n_entities, N], allocated when an env first deviates from the template
        self._default_colours = np.array([en.COLOUR_ID[e.colour_name] if hasattr(e, 'colour_name') else -1 for e in self._entities], dtype=np.int64)
        self.entity_colours = np.tile(self._default_colours, (self.n_envs, 1))          # per env
        # initial (x, y, angle) of every entity's main body, per env
        self._default_poses = np.array([[e.init_pos[0], e.init_pos[1], e.init_angle] if hasattr(e, 'init_pos')
                                        else [e.x + e.w / 2, e.y - e.h / 2, 0.0]          # GoalRegion body: box centre (entities.py:794-797)
                                        for e in self._entities], dtype=np.float64)
        self.entity_poses = np.tile(self._default_poses, (self.n_envs, 1, 1))
        self._ent_pose = None        # device [n_entities * 3, N], allocated when an env first deviates from the template
        # goal regions (sensors: rendering + scoring only): GoalRegion(x, y, h, w) per env, in entity order
        self._goal_ent_idx = [i for i, e in enumerate(self._entities) if isinstance(e, en.GoalRegion)]
        self._goal_xyhw0 = np.array([[self._entities[i].x, self._entities[i].y, self._entities[i].h, self._entities[i].w]
                                     for i in self._goal_ent_idx], dtype=np.float64).reshape(-1, 4)
        self.goal_xyhw = np.tile(self._goal_xyhw0, (self.n_envs, 1, 1))
        self._goal_rect = None       # device float64 [n_goals * 4, N], allocated when an env first deviates
        # pose-blob row of (x, y, angle) per body, -1 where the body is not persistent
        n = nat.check(L.mgx_world_n_state_entries(w))
        self._pose_rows = -np.ones((self.n_bodies, 3), dtype=np.int64)
        self._pose_sel = None       # device copy of the row table (get_poses_tensor)
        self._motion_rows = -np.ones((self.n_bodies, 9), dtype=np.int64)
        for k in range(n):
            b, c, r = C.c_int(), C.c_int(), C.c_int()
            nat.check(L.mgx_world_state_entry(w, k, C.byref(b), C.byref(c), C.byref(r)))
            if c.value < 3:
                self._pose_rows[b.value, c.value] = r.value
            else:
                self._motion_rows[b.value, c.value] = r.value
        del out

    @property
    def lanes_per_env(self):
        """Lanes of a wavefront that step one env (with per-env worlds the engine re-picks it as worlds come and go)."""
        return self._lib.mgx_engine_lanes_per_env(self._engine)

    def _info(self, key):
        out = C.c_int()
        nat.check(self._lib.mgx_world_info(self._world, nat.INFO[key], C.byref(out)))
        return out.value

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ reset / step
    def reset(self):
        """base_env.py:177-234 for every env.  Returns the first observation."""
        self._reset_envs(np.arange(self.n_envs), None)
        if not self._warm:
            # the first pose read-back loads torch's gather / copy kernels (tens of ms, once per process); pay for it
            # here rather than in the middle of the first rollout
            idx = np.arange(self.n_envs)
            self._scoring_envs = idx
            self.score_on_end_of_traj(self.get_poses(idx))      # likewise the scoring path (numpy / BLAS / shape-table set-up)
            if not self.score_needs_poses:
                self._overlap = self.region_overlaps(None)
                self.score_on_end_of_traj(None)
            self._check_capacity(None, 0)
            if self.auto_reset:
                self._pinned_snapshots()        # pinning host memory costs milliseconds: not at the first episode end
            self._warm = True
        obs = self._observe(fill_all=True)
        if self.copy_obs:
            obs = {k: v.clone() for k, v in obs.items()} if isinstance(obs, dict) else obs.clone()
        return obs

    def step(self, actions):
        import torch
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), device=self.device)
        actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        assert actions.shape == (self.n_envs,)
        # a step in which no episode ends (nothing is reset between the physics and the rendering) and whose observation is one
        # rasteriser pass goes out as ONE fused call: the raster kernel consumes envs as the step kernel finishes them
        ends = self.max_episode_steps is not None and bool((self._steps + 1 >= self.max_episode_steps).any())
        target = None if (ends or not self.overlap) else self._fused_target()
        if target is None:
            nat.check(self._lib.mgx_engine_step(self._engine, self.state_p.data_ptr(), self.state_f.data_ptr(),
                                                self.state_i.data_ptr(), actions.data_ptr(), self._done_dev.data_ptr(),
                                                self._stream()))
        else:
            # (a preprocessor with two rasteriser passes: the first one is the consumer of the step kernel, the second follows)
            first, rest = (target[0], target[1:]) if isinstance(target, list) else (target, ())
            out, view, layout = first
            nat.check(self._lib.mgx_engine_step_render(self._engine, self.state_p.data_ptr(), self.state_f.data_ptr(),
                                                       self.state_i.data_ptr(), actions.data_ptr(), self._done_dev.data_ptr(),
                                                       out.data_ptr(), out.stride(0), view, layout, self._stream()))
            for out, view, layout in rest:
                nat.check(self._lib.mgx_engine_render(self._engine, self.state_p.data_ptr(), out.data_ptr(), out.stride(0), view, layout,
                                                      None, self._stream()))
            self._obs_ready = True
        self._steps += 1
        done = np.zeros(self.n_envs, dtype=bool)
        eval_score = np.zeros(self.n_envs, dtype=np.float64)
        if self.max_episode_steps is not None:
            done = self._steps >= self.max_episode_steps
        fill = None
        obs = None
        if done.any():
            idx = np.nonzero(done)[0]
            self._scoring_envs = idx          # which envs the poses belong to (per-env task state of Test* variants)
            sel = None if len(idx) == self.n_envs else idx      # lockstep batches finish together: no gather then
            # Where the next episode draws nothing on the host (the Demo variants), the reset and the rasterisation of the new
            # episode's first frame are ENQUEUED before the host scores the old one: what the scores read (pose rows or the
            # device's overlap sets, the overflow counters) is snapshotted on the stream first, and the host then works on the
            # snapshots while the GPU resets and renders.  Tasks with per-episode draws keep per-env tables that the draws
            # overwrite, so they score first, as before.
            early = (self.auto_reset and not self.rand_dynamics and not self.variable_worlds and not self.sample_variation_is_active())
            snap_p = snap_i = snap_o = snap_s = None
            dev_scores = self._device_point_scores()
            term = term_token = None
            want_term = self.terminal_observation and self.auto_reset
            if want_term and not early:
                term, term_token = self._terminal_begin(idx)
            if early:
                # asynchronous copies into pinned host memory, an event after them, THEN the reset and the rasterisation: the host
                # waits for the event only, and scores while the GPU is still busy
                pin = self._pinned_snapshots()
                snap_i = pin['i']; snap_i.copy_(self.state_i[2], non_blocking=True)
                if dev_scores:
                    self._enqueue_point_scores(self._done_dev)
                    snap_s = pin['s']; snap_s.copy_(self._score_dev, non_blocking=True)
                elif self.score_needs_poses:
                    snap_p = pin['p']; snap_p.copy_(self.state_p, non_blocking=True)
                else:
                    self._enqueue_region_overlaps(self._done_dev)
                    snap_o = pin['o']; snap_o.copy_(self._overlap_dev, non_blocking=True)
                pin['ev'].record(torch.cuda.current_stream(self.device))
                if want_term:
                    term, term_token = self._terminal_begin(idx)
                self._reset_envs(idx, self._done_dev)
                fill = self._done_dev
                self._fill_idx = idx
                obs = self._observe(fill_mask=fill)
                pin['ev'].synchronize()
            if dev_scores:
                # the point tasks' scores come off the device whole (mgx_engine_score_points): no pose rows leave the GPU
                if not early:
                    self._enqueue_point_scores(self._done_dev)
                    snap_s = self._score_dev.cpu()
                eval_score[idx] = snap_s.numpy()[idx]
            elif self.score_needs_poses:
                eval_score[idx] = self.score_on_end_of_traj(self.get_poses(sel, source=snap_p))
            elif early:
                self._overlap = snap_o.numpy() if sel is None else snap_o.numpy()[:, :, sel]
                eval_score[idx] = self.score_on_end_of_traj(None)
            else:
                self._overlap = self.region_overlaps(sel, mask_dev=self._done_dev)
                eval_score[idx] = self.score_on_end_of_traj(None)
            assert np.all((eval_score >= 0) & (eval_score <= 1)), 'eval score out of range'
            self._check_capacity(sel, len(idx), source=snap_i)
            if self.auto_reset and not early:
                # the device-side done flags written by the step kernel double as reset + frame-fill masks
                self._reset_envs(idx, self._done_dev)
                fill = self._done_dev
                self._fill_idx = idx
        if obs is None:
            obs = self._observe(fill_mask=fill)
        self._fill_idx = None
        info = {'eval_score': eval_score}
        if done.any() and self.terminal_observation and self.auto_reset:
            self._terminal_end(term_token, self._done_dev)
            info['terminal_observation'], info['terminal_env_idx'] = term, idx
        if self.copy_obs:
            obs = {k: v.clone() for k, v in obs.items()} if isinstance(obs, dict) else obs.clone()
        return obs, self._reward, done, info

    def _check_capacity(self, idx, n_finished, source=None):
        """Chipmunk never drops a contact (base_env.py:243); this engine's per-env working set is sized from the world
        (every shape pair that can touch) and counts what did not fit in state_i[2].  The counter of the envs whose episode
        just ended is read here -- the stream is already drained by the pose download, so no extra synchronisation -- and a
        non-zero count is reported: a warning by default, an MgxError with strict_capacity=True."""
        import torch
        import warnings
        if source is None:
            row = self.state_i[2]
            n = int((row if idx is None else row[torch.as_tensor(idx, device=row.device)]).sum())
        else:
            row = source.numpy()
            n = int((row if idx is None else row[np.asarray(idx)]).sum())
        if n:
            self.capacity_overflows += n
            msg = (f'{n} contact(s) / overlapping pair(s) exceeded the per-env working set in {n_finished} finished episode(s) and were '
                   'dropped (the reference never drops contacts): results for those envs deviate from the reference')
            if self.strict_capacity:
                raise nat.MgxError(msg)
            warnings.warn(msg, nat.MgxCapacityWarning, stacklevel=3)

    # ------------------------------------------------------------------ per-env variation (Test* variants)
    def sample_variation(self, rng, k):
        """Task hook: draw this episode's random choices from `rng` with the same calls, in the same order, as the
        reference's on_reset() (after the physics variables, base_env.py:198-214), for env `k` (tasks keep what their
        score needs per env).  Return None (Demo) or a dict; supported keys: 'colours' = {entity: colour name},
        'poses' = {entity: (x, y, angle)}, 'goal_hw' = {goal region: (h, w)},
        'randomise_poses' = (entities, kwargs of geom.pm_randomise_all_poses), or a list of such stages and callables
        stage(poses[M, n_entities, 3], ent_hw[M, n_entities, 2] or None, place(entities, **kwargs)) -- the
        pose draws must come last in the reference's on_reset (they do in every task), because they are made after this
        hook returns, for all envs in one native call."""
        return None

    def sample_variation_batch(self, brng, env_idx):
        """Batched form of sample_variation(): the same draws for all envs `env_idx` of a reset at once (brng: batch_rng.BatchRng
        over their streams, in env_idx order).  Returns None or a dict of arrays over the m envs: 'colours' int64[m, n_entities]
        (whole rows of native colour ids), 'shape_types' int32[m, n_entities] / 'enabled' bool[m, n_entities] (whole rows),
        'goal_hw' {goal entity id: (h[m], w[m])}, 'randomise_poses' as in sample_variation().  Tasks that do not override it
        are drawn env by env."""
        return None

    def sample_variation_is_active(self):
        """Does sample_variation() draw anything for this task variant?  (The reference's on_reset branches on constructor
        flags only, so the answer is the same for every env and episode; asked on a scratch stream.)"""
        if getattr(self, '_variation_active', None) is None:
            # the hook writes env 0's row of the task's per-env tables (FixColour._keep_env, FindDupe._is_target_env,
            # Cluster._class_env): the probe must leave them as they were -- its first call may come from step() at an episode
            # end of an env restored with set_state() and never reset(), right before that table is scored
            saved = {a: (None if getattr(self, a, None) is None else np.array(getattr(self, a), copy=True)) for a in self.TASK_STATE_ATTRS}
            try:
                self._variation_active = self.sample_variation(np.random.RandomState(0), 0) is not None
            finally:
                for a, v in saved.items():
                    setattr(self, a, v)
        return self._variation_active

    def default_entity_poses(self):
        """float64[n_entities, 3] copy of the Demo layout, indexed like self._entities (= ent_id order)."""
        return self._default_poses.copy()

    def _reset_envs(self, env_idx, mask_dev):
        """BaseEnv.reset() for the envs `env_idx` (device mask `mask_dev`, None = all): per-env draws first (env k from
        its own stream self.rngs[k], in the reference's order: physics variables, then the task's on_reset choices), then
        the reset kernel -- with the drawn entity poses if any -- then the drawn force limits and colours."""
        import torch
        pvs, colour_rows, pose_rows, pose_spec, hw_rows, world_rows = [], [], [], None, {}, []
        # a task without per-episode draws (the Demo variants) returns None for every env and touches no stream: ask once
        draws = self.rand_dynamics or self.variable_worlds or (len(env_idx) > 0 and self.sample_variation_is_active())
        # (the batched form only where it belongs to the same class as the per-env form: a subclass that overrides
        # sample_variation() alone is drawn env by env)
        owner = lambda name: next(c for c in type(self).__mro__ if name in c.__dict__)
        batched = (draws and self.batch_draws and owner('sample_variation_batch') is not BaseEnv
                   and issubclass(owner('sample_variation_batch'), owner('sample_variation')))
        hw_batch, rng_addrs = None, None
        if batched:
            # every kind of draw for all envs of this reset at once, in the reference's order per env: physics variables first
            from .batch_rng import BatchRng
            brng = BatchRng([self.rngs[k] for k in env_idx], self._lib, addrs=self._rng_addrs(env_idx))
            rng_addrs = brng.addrs          # where the envs' MT19937 states live: the placement below draws from the same streams
            if self.rand_dynamics:
                pvs = PhysicsVariables.sample_batch(brng)
            var = self.sample_variation_batch(brng, np.asarray(env_idx)) or {}
            if 'colours' in var:
                colour_rows = var['colours']
            if self.variable_worlds:
                n_ent = len(self._entities)
                types = var.get('shape_types')
                enabled = var.get('enabled')
                types = np.tile(self._default_shape_types, (len(env_idx), 1)) if types is None else types
                enabled = np.ones((len(env_idx), n_ent), dtype=bool) if enabled is None else enabled
                world_rows = (np.ascontiguousarray(types, dtype=np.int32), np.ascontiguousarray(enabled, dtype=bool))
            pose_spec = var.get('randomise_poses')
            hw_batch = var.get('goal_hw')
        for k in (env_idx if (draws and not batched) else ()):
            rng = self.rngs[k]
            if self.rand_dynamics:
                pvs.append(PhysicsVariables.sample(rng))
            var = self.sample_variation(rng, int(k))
            if var is not None and 'colours' in var:
                row = self._default_colours.copy()
                for ent, name in var['colours'].items():
                    row[ent.ent_id] = en.COLOUR_ID[name]
                colour_rows.append(row)
            if self.variable_worlds:
                trow, erow = self._default_shape_types.copy(), np.ones(len(self._entities), dtype=bool)
                for ent, t in (var or {}).get('shape_types', {}).items():
                    trow[ent.ent_id] = en.SHAPE_TYPE_ID[t]          # str-Enum keys hash and compare as their values
                for ent, on in (var or {}).get('enabled', {}).items():
                    erow[ent.ent_id] = bool(on)
                world_rows.append((trow, erow))
            if var is not None and 'poses' in var:
                row = self._default_poses.copy()
                for ent, pose in var['poses'].items():
                    row[ent.ent_id] = pose
                pose_rows.append(row)
            if var is not None and 'randomise_poses' in var:
                pose_spec = var['randomise_poses']       # the same for every env of a task: one native call below
            if var is not None and 'goal_hw' in var:
                hw_rows[int(k)] = {g.ent_id: hw for g, hw in var['goal_hw'].items()}
        if len(world_rows):
            # this episode's world of every env being reset, before anything is placed in it
            idx32 = np.ascontiguousarray(env_idx, dtype=np.int32)
            if isinstance(world_rows, tuple):
                types, enabled = world_rows[0], np.ascontiguousarray(world_rows[1], dtype=np.uint8)
            else:
                types = np.ascontiguousarray(np.stack([t for t, _ in world_rows]), dtype=np.int32)
                enabled = np.ascontiguousarray(np.stack([e for _, e in world_rows]), dtype=np.uint8)
            self.entity_shape_types[env_idx], self.entity_enabled[env_idx] = types, enabled.astype(bool)
            nat.check(self._lib.mgx_engine_set_env_variants(self._engine, len(idx32), idx32.ctypes.data_as(C.POINTER(C.c_int)),
                                                            enabled.ctypes.data_as(C.POINTER(C.c_uint8)), types.ctypes.data_as(C.POINTER(C.c_int)),
                                                            self._stream()))
        ent_hw = None
        if hw_batch:
            # {goal entity id: (h[m], w[m])}
            ent_hw = np.zeros((len(env_idx), len(self._entities), 2), dtype=np.float64)
            ent_hw[:, self._goal_ent_idx] = self._goal_xyhw0[:, 2:]
            for e, (h, w) in hw_batch.items():
                ent_hw[:, e, 0], ent_hw[:, e, 1] = h, w
        if hw_rows:
            # resized goal regions keep their top-left corner (GoalRegion(x, y, h, w), entities.py:769-797): new centre
            ent_hw = np.zeros((len(env_idx), len(self._entities), 2), dtype=np.float64)
            ent_hw[:, self._goal_ent_idx] = self._goal_xyhw0[:, 2:]
            for i, k in enumerate(env_idx):
                for e, (h, w) in hw_rows.get(int(k), {}).items():
                    ent_hw[i, e] = (h, w)
        if pose_spec is not None or hw_rows or hw_batch:
            batch = np.ascontiguousarray(np.tile(self._default_poses, (len(env_idx), 1, 1)))
            if ent_hw is not None:
                for g, e in enumerate(self._goal_ent_idx):
                    batch[:, e, 0] = self._goal_xyhw0[g, 0] + ent_hw[:, e, 1] / 2
                    batch[:, e, 1] = self._goal_xyhw0[g, 1] - ent_hw[:, e, 0] / 2
            if pose_spec is not None:
                # geom.py pm_randomise_all_poses for all envs being reset, each on its own stream, natively
                from . import geom
                rngs = [self.rngs[k] for k in env_idx]
                rng_addrs = self._rng_addrs(env_idx) if rng_addrs is None else rng_addrs
                stages = pose_spec if isinstance(pose_spec, list) else [pose_spec]
                for stage in stages:
                    if callable(stage):
                        # task-specific step between placements (e.g. move a block onto its region), or a placement whose
                        # limits depend on the env; gets the poses so far, the envs' goal sizes and a runner for placements
                        stage(batch, ent_hw, lambda ents, **kw: geom.pm_randomise_all_poses_batch(
                            self, batch, ents, self.ARENA_BOUNDS_LRBT, rngs, ent_hw=ent_hw, env_idx=env_idx, addrs=rng_addrs, **kw))
                    else:
                        ents, kwargs = stage
                        geom.pm_randomise_all_poses_batch(self, batch, ents, self.ARENA_BOUNDS_LRBT, rngs, ent_hw=ent_hw, env_idx=env_idx, addrs=rng_addrs, **kwargs)
            pose_rows = batch             # (float64[M, n_entities, 3] as it is: a list of M rows would be re-assembled below)
            if len(self._goal_ent_idx):
                # the goal regions' rectangles of these envs, back in GoalRegion(x, y, h, w) form: x, y = top-left corner
                hw = ent_hw[:, self._goal_ent_idx] if ent_hw is not None else np.tile(self._goal_xyhw0[:, 2:], (len(env_idx), 1, 1))
                c = batch[:, self._goal_ent_idx, :2]
                self.set_goal_rects(np.stack([c[..., 0] - hw[..., 1] / 2, c[..., 1] + hw[..., 0] / 2, hw[..., 0], hw[..., 1]], axis=-1), env_idx)
        sp, sf, si = self.state_p.data_ptr(), self.state_f.data_ptr(), self.state_i.data_ptr()
        mask = None if mask_dev is None else mask_dev.data_ptr()
        if len(pose_rows):
            self.entity_poses[env_idx] = np.asarray(pose_rows)
            if self._ent_pose is None:
                self._ent_pose = torch.as_tensor(np.ascontiguousarray(self.entity_poses.reshape(self.n_envs, -1).T),
                                                 device=self.device).to(self.state_p.dtype).contiguous()       # [n_ent * 3, N]
            else:
                self._ent_pose[:, torch.as_tensor(env_idx, device=self.device)] = torch.as_tensor(
                    np.ascontiguousarray(self.entity_poses[env_idx].reshape(len(env_idx), -1).T), device=self.device).to(self.state_p.dtype)
        if self._ent_pose is not None:
            nat.check(self._lib.mgx_engine_reset_poses(self._engine, sp, sf, si, mask, self._ent_pose.data_ptr(), self._stream()))
        else:
            nat.check(self._lib.mgx_engine_reset(self._engine, sp, sf, si, mask, self._stream()))
        self._steps[env_idx] = 0
        if len(pvs):
            self.set_phys_vars(np.asarray(pvs, dtype=np.float64), env_idx)
        if len(colour_rows):
            self.set_entity_colours(np.asarray(colour_rows), env_idx)

    def set_entity_colours(self, colour_ids, env_idx=None):
        """colour_ids: int[M, n_entities] (entities.COLOUR_ID values; ignored for the robot) for the envs `env_idx`.
        Recolours the envs' primitives (fill / darkened outline / lightened goal interior, entities.py:712-757,807-819)."""
        import torch
        idx = np.arange(self.n_envs) if env_idx is None else np.asarray(env_idx)
        colour_ids = np.asarray(colour_ids, dtype=np.int64).reshape(len(idx), len(self._entities))
        self.entity_colours[idx] = colour_ids
        if self._ent_colour is None:
            self._ent_colour = torch.as_tensor(np.maximum(self.entity_colours, 0).T.astype(np.int32), device=self.device).contiguous()   # [n_ent, N]
            nat.check(self._lib.mgx_engine_set_entity_colours(self._engine, self._ent_colour.data_ptr()))
        else:
            self._ent_colour[:, torch.as_tensor(idx, device=self.device)] = torch.as_tensor(np.maximum(colour_ids, 0).T.astype(np.int32), device=self.device)

    def set_goal_rects(self, xyhw, env_idx=None):
        """xyhw: float64[M, n_goals, 4] = GoalRegion(x, y, h, w) of every goal region for the envs `env_idx`: what the
        rasteriser draws and what score_on_end_of_traj() sees as the sensors."""
        import torch
        idx = np.arange(self.n_envs) if env_idx is None else np.asarray(env_idx)
        xyhw = np.asarray(xyhw, dtype=np.float64).reshape(len(idx), len(self._goal_ent_idx), 4)
        self.goal_xyhw[idx] = xyhw
        if self._goal_rect is None:
            if self.state_p.dtype != torch.float64:
                raise NotImplementedError('per-env goal rectangles need fp64 poses (dtype f32 or f64)')
            self._goal_rect = torch.as_tensor(np.ascontiguousarray(self.goal_xyhw.reshape(self.n_envs, -1).T), device=self.device).contiguous()
            nat.check(self._lib.mgx_engine_set_goal_rects(self._engine, self._goal_rect.data_ptr()))
        else:
            self._goal_rect[:, torch.as_tensor(idx, device=self.device)] = torch.as_tensor(
                np.ascontiguousarray(xyhw.reshape(len(idx), -1).T), device=self.device)

    def _pinned_snapshots(self):
        """Pinned host buffers + the event of the episode-end snapshots (allocated on first use)."""
        import torch
        pin = getattr(self, '_pin', None)
        if pin is None:
            pin = {'i': torch.empty(self.n_envs, dtype=torch.int32).pin_memory(), 'ev': torch.cuda.Event()}
            if self._device_point_scores():
                pin['s'] = torch.empty(self.n_envs, dtype=torch.float64).pin_memory()
            elif self.score_needs_poses:
                pin['p'] = torch.empty(tuple(self.state_p.shape), dtype=self.state_p.dtype).pin_memory()
            else:
                pin['o'] = torch.empty((len(self._goal_ent_idx), len(self._entities), self.n_envs), dtype=torch.uint8).pin_memory()
            self._pin = pin
        return pin

    def device_score_spec(self):
        """Task hook: what mgx_engine_score_points needs to compute score_on_end_of_traj() on the device (None: the task scores on
        the host, from downloaded poses or overlap sets)."""
        return None

    def _device_point_scores(self):
        """Can the episode-end scores of this task be computed on the device, bit for bit?  (a point task, and a numpy whose two
        library primitives round in one of the ways the kernel knows: _scoring.numpy_dot_modes)"""
        use = getattr(self, '_use_dev_scores', None)
        if use is None:
            from .benchmarks._scoring import numpy_dot_modes
            use = self._use_dev_scores = (self.score_needs_poses and self.device_scores and self.device_score_spec() is not None
                                          and numpy_dot_modes() is not None)
        return use

    def _enqueue_point_scores(self, mask_dev=None):
        """score_on_end_of_traj() of the envs in the mask -> self._score_dev (float64[N] on the device), nothing synchronised."""
        import torch
        from .benchmarks._scoring import numpy_dot_modes
        spec = self.device_score_spec()
        dot_mode, mm_mode = numpy_dot_modes()
        if getattr(self, '_score_dev', None) is None:
            self._score_dev = torch.zeros(self.n_envs, dtype=torch.float64, device=self.device)
        ents = np.ascontiguousarray(spec['ents'], dtype=np.int32)
        cls_default = np.ascontiguousarray(spec.get('cls_default', [0] * len(ents)), dtype=np.int32)
        params = np.ascontiguousarray(spec['params'], dtype=np.float64)
        cls_env = spec.get('cls_env')
        cls_ptr = None
        if cls_env is not None:     # per-env classes (variants that redraw colours / types): the table as it stands now
            self._cls_dev = torch.as_tensor(np.ascontiguousarray(cls_env, dtype=np.int8), device=self.device)
            assert tuple(self._cls_dev.shape) == (self.n_envs, len(ents))
            cls_ptr = self._cls_dev.data_ptr()
        nat.check(self._lib.mgx_engine_score_points(
            self._engine, self.state_p.data_ptr(), spec['task'], len(ents), ents.ctypes.data_as(C.POINTER(C.c_int)),
            cls_default.ctypes.data_as(C.POINTER(C.c_int)), int(spec.get('n_classes', 1)), cls_ptr,
            params.ctypes.data_as(C.POINTER(C.c_double)), dot_mode, mm_mode, None if mask_dev is None else mask_dev.data_ptr(),
            self._score_dev.data_ptr(), self._stream()))

    def _enqueue_region_overlaps(self, mask_dev=None):
        import torch
        ng, ne = len(self._goal_ent_idx), len(self._entities)
        if getattr(self, '_overlap_dev', None) is None:
            self._overlap_dev = torch.zeros((ng, ne, self.n_envs), dtype=torch.uint8, device=self.device)
        nat.check(self._lib.mgx_engine_score_overlaps(self._engine, self.state_p.data_ptr(), None if mask_dev is None else mask_dev.data_ptr(),
                                                      self._overlap_dev.data_ptr(), self._stream()))

    def region_overlaps(self, env_idx=None, mask_dev=None):
        """GoalRegion.get_overlapping_ents(com_overlap=True) (entities.py:821-881) on the device for the envs `env_idx`
        (default all): numpy u8 [n_goals, n_entities, M], bit 0 = body position inside the region's box, bit 1 = every shape of
        the block overlaps the region (goals in entity order; mgx_engine_score_overlaps).  Synchronises (one small copy)."""
        import torch
        self._enqueue_region_overlaps(mask_dev)
        out = self._overlap_dev if env_idx is None else self._overlap_dev[:, :, torch.as_tensor(env_idx, device=self.device)]
        return out.cpu().numpy()

    def goal_bb(self, goal):
        """Sensor box (l, b, r, t) of a goal region for the envs being scored: scalars while every env has the
        template's rectangle, float64[M] arrays otherwise (same arithmetic as mgx_world_goal_bb, entities.py:794-797)."""
        if self._goal_rect is None:
            return goal.bb
        g = self._goal_ent_idx.index(self._entities.index(goal))
        x, y, h, w = (self.goal_xyhw[self._scoring_envs, g, c] for c in range(4))
        cx, cy, hw, hh = x + w / 2, y - h / 2, w / 2, h / 2
        return cx - hw, cy - hh, cx + hw, cy + hh

    def set_phys_vars(self, values, env_idx=None):
        """values: float64[M, 5] joint max forces (robot_pos, robot_rot, robot_finger, shape_trans, shape_rot) of the
        envs `env_idx` (default: all).  Stored as max impulse per substep, like the template's own limits."""
        import torch
        values = np.asarray(values, dtype=np.float64).reshape(-1, 5)
        idx = np.arange(self.n_envs) if env_idx is None else np.asarray(env_idx)
        assert values.shape[0] == len(idx)
        self.phys_vars[idx] = values
        dt = 1.0 / self.fps / self.phys_steps
        row = self._info('physvar_row')
        imp = torch.as_tensor((values * dt).T.copy(), device=self.device).to(self.state_f.dtype)        # [5, M]
        self.state_f[row:row + 5, torch.as_tensor(idx, device=self.device)] = imp

    def close(self):
        if self._engine is not None:
            self._lib.mgx_engine_destroy(self._engine)
            self._engine = None
        if self._world is not None:
            self._lib.mgx_world_destroy(self._world)
            self._world = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ observations
    def _observation_space(self):
        """Space of ONE env's observation.  Without a preprocessor that is this engine's state-only observation; the
        reference's Dict{'allo','ego': Box(0,255,(384,384,3),u8)} (base_env.py:97-107) is what `render()` returns."""
        return spaces.Box(-np.inf, np.inf, (self.n_bodies, 3), np.float32)

    def _terminal_begin(self, idx):
        """After the physics of a step in which the envs `idx` finish, before they are reset: (their terminal observations,
        a token for _terminal_end).  Default (observations made afresh on every call): just this step's observation."""
        import torch
        it = torch.as_tensor(idx, device=self.device)
        obs = self._observe()
        return ({k: v[it].clone() for k, v in obs.items()} if isinstance(obs, dict) else obs[it].clone()), None

    def _terminal_end(self, token, done_dev):
        """After the observation of the step has been made: undo what _terminal_begin did to the envs that did not finish."""

    def _fused_target(self):
        """(tensor, view, layout) of the rasteriser pass that makes this env's observation -- or a list of them, in launch order, for
        a preprocessor that renders two views -- or None (state-only observation)."""
        return None

    def handoff_stats(self):
        """(deferred, timeouts) of the fused step: consumer workgroups that did not wait and were served by the clean-up launch."""
        d, t = C.c_uint(), C.c_uint()
        nat.check(self._lib.mgx_engine_handoff_stats(self._engine, C.byref(d), C.byref(t)))
        return d.value, t.value

    def _observe(self, fill_all=False, fill_mask=None):
        """Default (no preprocessor): state-only observation f32[N, n_bodies, 3] = (x, y, angle)."""
        return self.get_poses_tensor()

    def get_poses_tensor(self):
        import torch
        sel = getattr(self, '_pose_sel', None)
        if sel is None:     # the row table is fixed once the world is built: upload it once, not per step
            rows = torch.as_tensor(np.where(self._pose_rows >= 0, self._pose_rows, 0).reshape(-1), device=self.device)
            valid = torch.as_tensor((self._pose_rows >= 0).reshape(-1), device=self.device)[:, None].to(self.state_p.dtype)
            sel = self._pose_sel = (rows, valid)
        p = self.state_p.index_select(0, sel[0]) * sel[1]
        return p.reshape(self.n_bodies, 3, self.n_envs).permute(2, 0, 1).to(torch.float32)

    def get_poses(self, env_idx=None, source=None):
        """float64[M, n_bodies, 3] poses of the selected envs on the host (synchronises).  source: a snapshot of the pose blob."""
        import torch
        if source is None:
            sp = self.state_p if env_idx is None else self.state_p[:, torch.as_tensor(env_idx, device=self.device)]
            sp = sp.to(torch.float64).cpu().numpy()
        else:
            # a host snapshot: gathered with numpy (torch's CPU indexing wakes its whole thread pool, whose workers then spin
            # for a while -- measured as 50 ms hiccups of the steps after a partial episode end)
            sp = source.numpy()
            if env_idx is not None:
                sp = sp[:, np.asarray(env_idx)]
            sp = sp.astype(np.float64, copy=False)
        out = np.zeros((sp.shape[1], self.n_bodies, 3), dtype=np.float64)
        for b in range(self.n_bodies):
            for c in range(3):
                r = self._pose_rows[b, c]
                if r >= 0:
                    out[:, b, c] = sp[r]
        return out

    # ------------------------------------------------------------------ checkpoint / resume
    # per-env task bookkeeping (numpy arrays or None) that belongs to an episode in flight; tasks list theirs
    TASK_STATE_ATTRS = ()

    def get_state(self):
        """Snapshot of everything the envs' futures depend on: the three device state blobs, the per-env episode choices
        (force limits, colours, poses, goal rectangles, worlds), the tasks' per-env bookkeeping, the episode counters and
        every env's random stream.  The reference can only pickle constructor arguments (base_env.py:20-46); this is the
        batched engine's counterpart of rebuilding a world from reconstruct_signature() dumps (entities.py:46-84)."""
        d = {'state_p': self.state_p.clone(), 'state_f': self.state_f.clone(), 'state_i': self.state_i.clone(),
             'steps': self._steps.copy(), 'phys_vars': self.phys_vars.copy(), 'entity_colours': self.entity_colours.copy(),
             'entity_poses': self.entity_poses.copy(), 'goal_xyhw': self.goal_xyhw.copy(),
             'entity_shape_types': self.entity_shape_types.copy(), 'entity_enabled': self.entity_enabled.copy(),
             'rng': [r.get_state() for r in self.rngs],
             'task': {a: (None if getattr(self, a) is None else getattr(self, a).copy()) for a in self.TASK_STATE_ATTRS},
             'have': {'colours': self._ent_colour is not None, 'poses': self._ent_pose is not None, 'goals': self._goal_rect is not None}}
        return d

    def set_episode_steps(self, steps):
        """Set every env's episode step counter (the TimeLimit clock of benchmarks/__init__.py:979-999): host mirror and the
        device row the step kernel increments.  Envs of one batch can then be at different points of their episodes, as they
        are when a long-running job resumes from get_state() snapshots taken at different times."""
        import torch
        steps = np.asarray(steps, dtype=np.int64).reshape(self.n_envs)
        assert (steps >= 0).all() and (self.max_episode_steps is None or (steps < self.max_episode_steps).all())
        self._steps[:] = steps
        self.state_i[0].copy_(torch.as_tensor(steps.astype(np.int32), device=self.device))

    def set_state(self, d):
        """Inverse of get_state() on an env of the same task / variant / size (any seed, any history)."""
        import torch
        assert d['state_p'].shape == self.state_p.shape and d['state_f'].shape == self.state_f.shape and d['state_i'].shape == self.state_i.shape
        idx = np.arange(self.n_envs)
        if self.variable_worlds:
            self.entity_shape_types[:], self.entity_enabled[:] = d['entity_shape_types'], d['entity_enabled']
            idx32 = np.ascontiguousarray(idx, dtype=np.int32)
            types = np.ascontiguousarray(self.entity_shape_types, dtype=np.int32)
            enabled = np.ascontiguousarray(self.entity_enabled, dtype=np.uint8)
            nat.check(self._lib.mgx_engine_set_env_variants(self._engine, len(idx32), idx32.ctypes.data_as(C.POINTER(C.c_int)),
                                                            enabled.ctypes.data_as(C.POINTER(C.c_uint8)), types.ctypes.data_as(C.POINTER(C.c_int)),
                                                            self._stream()))
        self.state_p.copy_(d['state_p']); self.state_f.copy_(d['state_f']); self.state_i.copy_(d['state_i'])
        self._steps[:] = d['steps']
        self.phys_vars[:] = d['phys_vars']
        if d['have']['colours'] or self._ent_colour is not None:
            self.set_entity_colours(d['entity_colours'])
        else:
            self.entity_colours[:] = d['entity_colours']
        if d['have']['goals'] or self._goal_rect is not None:
            self.set_goal_rects(d['goal_xyhw'])
        else:
            self.goal_xyhw[:] = d['goal_xyhw']
        self.entity_poses[:] = d['entity_poses']
        if d['have']['poses'] or self._ent_pose is not None:      # only the reset kernel reads these
            self._ent_pose = torch.as_tensor(np.ascontiguousarray(self.entity_poses.reshape(self.n_envs, -1).T),
                                             device=self.device).to(self.state_p.dtype).contiguous()
        for r, st in zip(self.rngs, d['rng']):
            r.set_state(st)
        for a, v in d['task'].items():
            setattr(self, a, None if v is None else v.copy())

    def get_bodies(self):
        """float64[N, n_bodies, 9] (x y a vx vy w vbx vby wb); non-persistent components are 0."""
        sp = self.state_p.to('cpu').numpy().astype(np.float64)
        sf = self.state_f.to('cpu').numpy().astype(np.float64)
        out = np.zeros((self.n_envs, self.n_bodies, 9), dtype=np.float64)
        for b in range(self.n_bodies):
            for c in range(9):
                r = self._pose_rows[b, c] if c < 3 else self._motion_rows[b, c]
                if r >= 0:
                    out[:, b, c] = (sp if c < 3 else sf)[r]
        return out

    def set_bodies(self, bodies):
        """Inverse of get_bodies (parity tests / checkpoint restore)."""
        import torch
        sp = self.state_p.to('cpu').numpy().copy()
        sf = self.state_f.to('cpu').numpy().copy()
        for b in range(self.n_bodies):
            for c in range(9):
                r = self._pose_rows[b, c] if c < 3 else self._motion_rows[b, c]
                if r >= 0:
                    (sp if c < 3 else sf)[r] = bodies[:, b, c]
        self.state_p.copy_(torch.as_tensor(sp))
        self.state_f.copy_(torch.as_tensor(sf))

    def substeps(self, actions, n):
        """n physics substeps under `actions` without advancing the episode counter (parity tests)."""
        import torch
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions))
        actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        nat.check(self._lib.mgx_engine_substeps(self._engine, self.state_p.data_ptr(), self.state_f.data_ptr(),
                                                self.state_i.data_ptr(), actions.data_ptr(), int(n), self._stream()))

    def render_frames(self, out, view='ego', layout='frame', fill_mask=None):
        """Rasterise every env into `out` (torch.uint8 on the engine device): [N,96,96,3] or [N,96,96,12]."""
        lay = {'frame': nat.OBS_FRAME, 'stack4': nat.OBS_STACK4, 'stack3_hi': nat.OBS_STACK3_HI, 'slot_lo': nat.OBS_SLOT_LO,
               'planar': nat.OBS_PLANAR}[layout]
        # one env's pixels are contiguous; envs may be strided (a slot of a ring of frames)
        assert out[0].is_contiguous() and out.device == self.device and out.shape[0] == self.n_envs
        nat.check(self._lib.mgx_engine_render(self._engine, self.state_p.data_ptr(), out.data_ptr(), out.stride(0),
                                              nat.VIEW_EGO if view == 'ego' else nat.VIEW_ALLO, lay,
                                              None if fill_mask is None else fill_mask.data_ptr(), self._stream()))
        return out

    def render(self, mode='rgb_array', env=0):
        """base_env.py:309-338 for ONE env at the native 384x384: {'allo': u8[384,384,3], 'ego': ...} (numpy)."""
        import torch
        assert mode == 'rgb_array'
        views = {}
        buf = torch.empty((384, 384, 3), dtype=torch.uint8, device=self.device)
        for name, flag, vid in (('allo', self.allo_view, nat.VIEW_ALLO), ('ego', self.ego_view, nat.VIEW_EGO)):
            if flag:
                nat.check(self._lib.mgx_engine_render_native(self._engine, self.state_p.data_ptr(), int(env),
                                                             buf.data_ptr(), vid, self._stream()))
                views[name] = buf.cpu().numpy().copy()
        return views

    # ------------------------------------------------------------------ kernel timing (bench.py)
    def set_timing(self, every):
        """Bracket every `every`-th kernel launch with HIP events (0 = off)."""
        nat.check(self._lib.mgx_engine_set_timing(self._engine, int(every)))

    def read_timing(self, which, max_n=4096):
        buf = (C.c_float * max_n)()
        n = nat.check(self._lib.mgx_engine_timing_read(self._engine, 0 if which == 'step' else 1, buf, max_n))
        return np.array(buf[:n], dtype=np.float64)
