"""Regenerates tests/golden/variant_vectors.json with the CPU oracle (oracle/, test infrastructure):

    python tests/golden/make_variant_vectors.py

For Test* variants whose episodes differ in shape types / entity counts / colours / layout (one TestAll or TestCountPlus
per task): what a seeded oracle env draws at its first two resets -- which of the task's entity slots exist, the blocks'
shape types, every entity's pose -- plus SHA-256 of the first LoRes4E observation and the score after a short action tape.
Uses: tests/test_oracle_render.py (CPU: the oracle's restatement of the reference's on_reset draws still consumes the
random stream exactly like this) and tests/test_gpu_parity.py (GPU: the product, seeded alike, draws the same worlds and
renders byte-identical first observations, without consulting the oracle).
Like oracle_vectors.json these record the restatement, not the reference.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
_C = {'rand_layout_full': True}
CASES = [
    ('MoveToCorner', 'TestAll', {'rand_shape_colour': True, 'rand_shape_type': True, 'rand_poses': True, 'rand_dynamics': True}),
    ('MatchRegions', 'TestCountPlus', dict(_C, rand_target_colour=True, rand_shape_type=True, rand_shape_count=True)),
    ('MakeLine', 'TestCountPlus', dict(_C, rand_colours=True, rand_shapes=True, rand_count=True)),
    ('FindDupe', 'TestCountPlus', dict(_C, rand_colours=True, rand_shapes=True, rand_count=True)),
    ('FixColour', 'TestCountPlus', dict(_C, rand_colours=True, rand_shapes=True, rand_count=True)),
    ('ClusterColour', 'TestCountPlus', dict(_C, rand_shape_colour=True, rand_shape_type=True, rand_shape_count=True)),
    ('ClusterShape', 'TestShape', {'rand_shape_type': True}),
]
SEED, EP = 9001, 4


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def record(task, flags, seed):
    from oracle.entities_ref import GoalRegion
    from oracle.env_ref import LoRes4ERef, RefEnv
    env = LoRes4ERef(RefEnv(task, max_episode_steps=EP, seed=seed, **flags))
    out = []
    for episode in range(2):
        obs = env.reset()
        t = env.env.task
        slots = t.slots
        rec = {'enabled': [s is not None for s in slots],
               'shape_types': [str(s.shape_type) if s is not None and hasattr(s, 'shape_type') else None for s in slots],
               'poses': [list(t.main_pose(s)) if s is not None and not isinstance(s, GoalRegion) else None for s in slots],
               'lores4e': sha(obs)}
        tape = np.random.RandomState(seed + episode).randint(0, 18, size=EP).tolist()
        for a in tape:
            _, _, done, info = env.step(a)
        rec['tape'], rec['score'] = tape, info['eval_score']
        out.append(rec)
    return out


def main():
    out = {f'{task}-{variant}': {'flags': flags, 'seed': SEED, 'episode_steps': EP, 'episodes': record(task, flags, SEED)}
           for task, variant, flags in CASES}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variant_vectors.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote variant_vectors.json for', len(out), 'variants')


if __name__ == '__main__':
    main()
