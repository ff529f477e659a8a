#!/usr/bin/env python
"""Free-running pose drift of the shipped (fp32) engine against the oracle, next to the oracle's own spread, up to env-step
200 (BASELINE.json: "pose drift < 1e-3 over 200 steps"), for all 8 Demo tasks -- the long form of
tests/test_gpu_parity.py::test_f32_drift_within_perturbation_envelope.

    python tools/drift_table.py [--envs 48] [--steps 200] > profiles/rNN_pose_drift_vs_oracle_envelope.txt

Per task and env-step: median / p90 over the action tapes of
  engine        max |pose difference| engine vs oracle, same tape
  replica 1e-7  oracle vs a copy of itself whose poses started U(-1e-7, 1e-7) off (one fp32 rounding at unit scale)
  fp32 state    oracle vs a copy that stores its velocities in fp32 after every substep (the engine's storage format)
(poses: x, y, angle of every body whose pose is persistent state; arena = [-1, 1]^2, so 1e-3 is 0.05 % of the arena.)
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=48)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--dtype', default='f32')
    args = ap.parse_args()
    import magical_amd
    from tests.util import EPS_F32, TASKS, OracleEnvelope, masked_err, new_ref, quantiles, velround_step
    n, T = args.envs, args.steps
    marks = [s for s in (1, 2, 5, 10, 20, 40, 80, 120, 160, 200) if s <= T]
    print(__doc__.split('\n\n')[2])
    for task in TASKS:
        tape = np.random.RandomState(7).randint(0, 18, size=(T, n)).astype(np.int32)
        env = magical_amd.make(f'{task}-Demo-v0', n_envs=n, device='cuda:0', max_episode_steps=10 ** 6, dtype=args.dtype)
        env.reset()
        orc = OracleEnvelope([lambda: new_ref(task)] * n, K=1, eps=EPS_F32, seed=2)
        vr = [new_ref(task) for _ in range(n)]
        print(f'\n{task}  ({n} tapes, dtype {args.dtype}; env-step = 10 substeps)')
        print('  env-step |      engine med / p90 | replica 1e-7 med / p90 |   fp32 state med / p90')
        for s in range(T):
            env.step(tape[s])
            got = env.get_bodies()[:, 1:, :3]
            want, now = orc.step(tape[s])
            for k, r in enumerate(vr):
                velround_step(r, tape[s, k])
            if s + 1 in marks:
                d = np.array([masked_err(got[k], want[k], orc.mask) for k in range(n)])
                v = np.array([masked_err(r.bodies()[orc.idx][:, :3], want[k], orc.mask) for k, r in enumerate(vr)])
                (a, b), (c, e), (f, g) = quantiles(d), quantiles(now), quantiles(v)
                print(f'  {s + 1:8d} | {a:9.2e} / {b:9.2e} | {c:9.2e} / {e:9.2e} | {f:9.2e} / {g:9.2e}')
        env.close()


if __name__ == '__main__':
    main()
