"""Demonstration files: the host-side mirror of magical/saved_trajectories.py.

The reference's demonstrations (`magical-data`, `.pkl.gz`) are gzip'ed pickles of
`{'trajectory': MAGICALTrajectory(acts, obs, rews, infos), 'score': float, 'env_name': str}` with FULL-RESOLUTION
observations `obs = {'allo': u8[T + 1, 384, 384, 3], 'ego': ...}` (saved_trajectories.py:14-47).  This module

  * reads them (`load_demos`, :36-47) without the `magical` / `imitation` packages: the trajectory class is found under
    every module path the reference's own unpickler accepts (:24-33) plus the reference's own; apart from those, only an
    explicit allow-list of (module, name) pairs -- numpy array / scalar / dtype reconstruction, builtin containers -- may unpickle;
  * applies a preprocessor to the recorded frames (`preprocess_demos_with_wrapper`, :81-149) -- what the reference
    does by replaying the frames through its gym wrappers: FlattenFrameStack / EagerDictFrameStack + a 384 -> 96
    INTER_AREA resize (benchmarks/__init__.py:80-136,139-169,208-274), i.e. for every step the exact 4x4 box mean of
    the most recent frames, cvRound'ed (ties to even), oldest frame first;
  * replays a demo's action tape through the engine (`replay_demos`): the recorded scores are the corpus SURVEY.md
    section 8f-4 names for end-to-end parity once the data is at hand.

Data preparation runs on the host in numpy: it is a one-off pass over files, not the step path.
"""
import gzip
import io
import pickle
from typing import List, NamedTuple, Optional

import numpy as np

from .benchmarks import AVAILABLE_PREPROCESSORS, update_magical_env_name


class MAGICALTrajectory(NamedTuple):
    """saved_trajectories.py:14-21 (field-compatible with imitation's Trajectory)."""
    acts: np.ndarray
    obs: dict
    rews: np.ndarray
    infos: Optional[List[dict]]


_TRAJ_CLASSES = {('magical.saved_trajectories', 'MAGICALTrajectory'), ('imitation.util.rollout', 'Trajectory'),
                 ('milbench.baselines.saved_trajectories', 'MILBenchTrajectory'), (__name__, 'MAGICALTrajectory')}
# everything else a demo file may reference, by exact (module, name): what numpy arrays / scalars / dtypes and builtin
# containers pickle to (numpy >= 1.x `numpy.core`, numpy 2 `numpy._core`; `_codecs.encode` rebuilds the byte strings of
# protocol-2 array pickles).  No prefixes: `numpy.testing`, `numpy.distutils`, `builtins.eval` ... are all refused.
_ALLOWED_GLOBALS = {(m, n) for m in ('numpy.core.multiarray', 'numpy._core.multiarray') for n in ('_reconstruct', 'scalar')} | {
    ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer'), ('numpy', 'ndarray'), ('numpy', 'dtype'),
    ('collections', 'OrderedDict'), ('_codecs', 'encode'), ('copyreg', '_reconstructor'),
} | {('builtins', n) for n in ('dict', 'list', 'tuple', 'set', 'frozenset', 'int', 'float', 'complex', 'str', 'bytes', 'bytearray',
                               'bool', 'slice', 'range', 'object')}


class _TrajRewriteUnpickler(pickle.Unpickler):
    """saved_trajectories.py:24-33, restricted: trajectory classes map to MAGICALTrajectory; the globals on the explicit
    allow-list above (numpy array / scalar / dtype reconstruction, builtin containers) load as they are; anything else is
    refused (a demo file holds nothing else)."""

    def find_class(self, module, name):
        if (module, name) in _TRAJ_CLASSES:
            return MAGICALTrajectory
        if (module, name) in _ALLOWED_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'demo files may not reference {module}.{name}')


def load_demos(demo_paths, rewrite_traj_cls=True, verbose=False):
    """saved_trajectories.py:36-47: generator over the demo dictionaries of `demo_paths`."""
    n_demos = len(demo_paths)
    for d_num, d_path in enumerate(demo_paths, start=1):
        if verbose:
            print(f"Loading '{d_path}' ({d_num}/{n_demos})")
        with gzip.GzipFile(d_path, 'rb') as fp:
            yield _TrajRewriteUnpickler(io.BytesIO(fp.read())).load()


def splice_in_preproc_name(base_env_name, preproc_name):
    """saved_trajectories.py:50-58: "MoveToCorner-Demo-v0" + "LoResStack" -> "MoveToCorner-Demo-LoResStack-v0"."""
    assert preproc_name in AVAILABLE_PREPROCESSORS, \
        f"no preprocessor named '{preproc_name}', options are {', '.join(AVAILABLE_PREPROCESSORS)}"
    return update_magical_env_name(base_env_name, preproc=preproc_name)


def area_resize_4x(frames):
    """cv2.resize(..., (96, 96), interpolation=INTER_AREA) of u8[..., 384, 384, C] frames: the integer-factor path is the
    exact 4x4 box mean, saturate_cast<uchar>(sum / 16) = round half to even."""
    frames = np.asarray(frames)
    assert frames.dtype == np.uint8 and frames.shape[-3] % 4 == 0 and frames.shape[-2] % 4 == 0
    h, w, c = frames.shape[-3:]
    s = frames.reshape(frames.shape[:-3] + (h // 4, 4, w // 4, 4, c)).astype(np.int32).sum(axis=(-4, -2))
    return ((s + 7 + ((s >> 4) & 1)) >> 4).astype(np.uint8)


def _stacked(frames, depth):
    """FlattenFrameStack / EagerDictFrameStack of one view: out[t] = frames[t - depth + 1 .. t] on the channel axis,
    oldest first, the first frame repeated before the start (reset() fills the deque with copies, :130-136)."""
    T = frames.shape[0]
    idx = np.clip(np.arange(T)[:, None] + np.arange(-depth + 1, 1)[None, :], 0, None)        # [T, depth]
    out = frames[idx]                                                                          # [T, depth, H, W, C]
    return np.ascontiguousarray(np.moveaxis(out, 1, 3).reshape(T, frames.shape[1], frames.shape[2], depth * frames.shape[3]))


def preprocess_obs(obs, preproc_name):
    """The observation sequence a preprocessor's wrapper stack produces from recorded full-resolution frames
    obs = {'allo': u8[T, 384, 384, 3], 'ego': ...} (benchmarks/__init__.py:242-274).  Resizing commutes with stacking on
    the channel axis, so every frame is resized once."""
    if preproc_name not in AVAILABLE_PREPROCESSORS:
        raise KeyError(preproc_name)
    small = {k: area_resize_4x(np.asarray(obs[k])) for k in (('allo', 'ego') if preproc_name in ('LoRes3EA', 'LoResStack')
                                                               else (('allo',) if preproc_name == 'LoRes4A' else ('ego',)))}
    if preproc_name == 'LoRes4A':
        return _stacked(small['allo'], 4)
    if preproc_name == 'LoRes4E':
        return _stacked(small['ego'], 4)
    if preproc_name == 'LoResCHW4E':
        return np.ascontiguousarray(np.moveaxis(_stacked(small['ego'], 4), 3, 1))
    if preproc_name == 'LoRes3EA':
        return np.concatenate([_stacked(small['allo'], 1), _stacked(small['ego'], 3)], axis=3)
    return {'allo': _stacked(small['allo'], 4), 'ego': _stacked(small['ego'], 4)}            # LoResStack


def preprocess_demos_with_wrapper(trajectories, orig_env_name=None, preproc_name=None, wrapper=None):
    """saved_trajectories.py:81-149 for the built-in preprocessors: every trajectory with its observations replaced by
    what the named preprocessor would have shown (acts / rews / infos unchanged).  `orig_env_name` is accepted for
    signature compatibility (the reference instantiates that env only to borrow its observation space)."""
    if wrapper is not None:
        raise NotImplementedError('custom gym wrappers cannot be applied without gym; pass preproc_name')
    assert preproc_name is not None
    out = []
    for traj in trajectories:
        assert len(traj.obs['ego' if 'ego' in traj.obs else 'allo']) == len(traj.acts) + 1, 'a trajectory has T + 1 observations'
        new_obs = preprocess_obs(traj.obs, preproc_name)
        out.append(type(traj)(acts=np.asarray(traj.acts), obs=new_obs, rews=np.asarray(traj.rews), infos=list(traj.infos) if traj.infos is not None else None))
    return out


def replay_demos(demo_dicts, device='cuda:0', preproc=None, **env_kwargs):
    """Replay the action tapes of demos of ONE env name through the engine, all demos in lockstep (one env per demo;
    shorter demos idle on action 0 after their end and are scored at their own last step).  Returns
    {'scores': f64[n], 'recorded_scores': f64[n], 'final_obs': [...]}: with the reference's data the two score columns are
    the end-to-end parity check of SURVEY.md section 8f-4 (demonstrations solve their task, so both should be ~1)."""
    import torch
    from . import make
    demo_dicts = list(demo_dicts)
    names = {d['env_name'] for d in demo_dicts}
    assert len(names) == 1, f'one env name per replay, got {sorted(names)}'
    name = names.pop()
    if preproc is not None:
        name = splice_in_preproc_name(name, preproc)
    tapes = [np.asarray(d['trajectory'].acts, dtype=np.int32).reshape(-1) for d in demo_dicts]
    n, T = len(tapes), max(len(t) for t in tapes)
    env = make(name, n_envs=n, device=device, auto_reset=False, max_episode_steps=None, **env_kwargs)
    try:
        obs = env.reset()
        scores = np.zeros(n, dtype=np.float64)
        for t in range(T):
            acts = np.array([tape[t] if t < len(tape) else 0 for tape in tapes], dtype=np.int32)
            obs, _, _, _ = env.step(torch.as_tensor(acts, device=env.device))
            ending = [k for k, tape in enumerate(tapes) if len(tape) == t + 1]
            if ending:
                idx = np.asarray(ending)
                env._scoring_envs = idx
                scores[idx] = env.score_on_end_of_traj(env.get_poses(idx))
        return {'scores': scores, 'recorded_scores': np.array([float(d.get('score', np.nan)) for d in demo_dicts]), 'final_obs': obs}
    finally:
        env.close()
