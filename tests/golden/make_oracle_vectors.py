"""Regenerates tests/golden/oracle_vectors.json with the CPU oracle (oracle/, test infrastructure):

    python tests/golden/make_oracle_vectors.py

For every Demo task: the action tape, the body poses the oracle reaches after it, and SHA-256 digests of the
observations (96x96 ego / allo frames, the LoRes4E stack) it renders at reset and at those poses.  Uses:
  * tests/test_oracle_*.py (CPU)  -- the oracle still produces exactly these vectors (a change of the oracle, of the
    compiler flags or of libm that alters them is noticed);
  * tests/test_gpu_parity.py (GPU) -- the HIP rasteriser, given the golden poses through the C ABI, produces byte-identical
    frames without the oracle being rebuilt or imported on the GPU box.
The vectors are NOT from the reference (pymunk / pyglet cannot run here): they record the
restatement, whose own pins are tests/golden/static_frames_48.npz and the constants of SURVEY.md Appendix D.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TASKS = ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']
T = 6


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    from oracle.env_ref import LoRes4ERef, RefEnv
    out = {}
    for k, task in enumerate(TASKS):
        env = LoRes4ERef(RefEnv(task))
        obs0 = env.reset()
        tape = np.random.RandomState(100 + k).randint(0, 18, size=T).tolist()
        rec = {'tape': tape, 'reset': {'ego': sha(env.env.render_lores('ego')), 'allo': sha(env.env.render_lores('allo')),
                                       'lores4e': sha(obs0), 'bodies': env.env.bodies()[:, :3].tolist()}}
        for a in tape:
            obs, _, _, _ = env.step(a)
        rec['final'] = {'ego': sha(env.env.render_lores('ego')), 'allo': sha(env.env.render_lores('allo')), 'lores4e': sha(obs),
                        'bodies': env.env.bodies().tolist()}
        out[task] = rec
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'oracle_vectors.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote oracle_vectors.json for', len(out), 'tasks')


if __name__ == '__main__':
    main()
