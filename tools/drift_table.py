#!/usr/bin/env python
"""Free-running pose drift of the engine against the oracle up to env-step 200 (BASELINE.json: "pose drift < 1e-3 over 200
steps"), for all 8 Demo tasks: the shipped fp32 build AND the all-fp64 (reference-precision) build on the same action tapes
and the same oracle run, next to the oracle's own spread -- the long form of
tests/test_gpu_parity.py::test_f32_drift_within_perturbation_envelope / test_f64_drift_meets_the_substep_target.

    python tools/drift_table.py [--envs 48] [--steps 200] [--norm linf|l2] > profiles/rNN_pose_drift_f64_vs_oracle.txt

Per task and env-step: median / p90 over the action tapes of
  engine f32     pose difference, shipped build (dtype f32: fp32 velocities / impulses / contacts, fp64 poses) vs oracle, same tape
  engine f64     ... the all-fp64 build (dtype f64: the reference's own precision)
  replica 1e-7   oracle vs a copy of itself whose poses started U(-1e-7, 1e-7) off (one fp32 rounding at unit scale)
  replica 1e-13  ... U(-1e-13, 1e-13) off (a few hundred fp64 roundings: what a second fp64 engine differs by)
  fp32 state     oracle vs a copy that stores its velocities in fp32 after every substep (the fp32 build's storage format)
(poses: x, y, angle of every body whose pose is persistent state; arena = [-1, 1]^2, so 1e-3 is 0.05 % of the arena.
 norm linf = the largest single component per env -- what the tests gate --, l2 = the Euclidean norm over them -- BASELINE's word.)
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=48)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--norm', default='l2', choices=('linf', 'l2'))
    ap.add_argument('--tasks', nargs='*')
    args = ap.parse_args()
    import magical_amd
    from tests.util import EPS_F32, EPS_F64, TASKS, comparable_mask, new_ref, perturb_bodies, quantiles, ref_body_index, velround_step
    n, T = args.envs, args.steps
    marks = [s for s in (1, 2, 5, 10, 20, 40, 80, 120, 160, 200) if s <= T]
    print(__doc__.split('\n\n')[2])
    print(f'norm: {args.norm}')

    def err(a, b, mask):
        d = np.abs(np.asarray(a) - np.asarray(b))[mask]
        return float(d.max()) if args.norm == 'linf' else float(np.sqrt((d * d).sum()))
    summary = []
    for task in (args.tasks or TASKS):
        tape = np.random.RandomState(7).randint(0, 18, size=(T, n)).astype(np.int32)
        envs = {d: magical_amd.make(f'{task}-Demo-v0', n_envs=n, device='cuda:0', max_episode_steps=10 ** 6, dtype=d) for d in ('f32', 'f64')}
        for e in envs.values():
            e.reset()
        rs = np.random.RandomState(2)
        base = [new_ref(task) for _ in range(n)]
        rep7, rep13, vr = [new_ref(task) for _ in range(n)], [new_ref(task) for _ in range(n)], [new_ref(task) for _ in range(n)]
        for r in rep7:
            perturb_bodies(r, EPS_F32, rs)
        for r in rep13:
            perturb_bodies(r, EPS_F64, rs)
        idx, mask = ref_body_index(base[0]), comparable_mask(base[0])
        print(f'\n{task}  ({n} tapes; env-step = 10 substeps)')
        print('  env-step |  engine f32 med / p90 |  engine f64 med / p90 | replica 1e-7 med / p90 | replica 1e-13 med / p90 |  fp32 state med / p90')
        for s in range(T):
            for e in envs.values():
                e.step(tape[s])
            for k in range(n):
                base[k].step(tape[s, k]); rep7[k].step(tape[s, k]); rep13[k].step(tape[s, k])
                velround_step(vr[k], tape[s, k])
            if s + 1 in marks:
                want = [b.bodies()[idx][:, :3] for b in base]
                cols = []
                for d in ('f32', 'f64'):
                    got = envs[d].get_bodies()[:, 1:, :3]
                    cols.append(quantiles([err(got[k], want[k], mask) for k in range(n)]))
                for other in (rep7, rep13, vr):
                    cols.append(quantiles([err(r.bodies()[idx][:, :3], want[k], mask) for k, r in enumerate(other)]))
                print(f'  {s + 1:8d} | ' + ' | '.join(f'{a:9.2e} / {b:9.2e}' for a, b in cols))
                if s + 1 in (20, 200):
                    summary.append((task, s + 1, cols[0][0], cols[1][0], cols[2][0], cols[3][0]))
        for e in envs.values():
            e.close()
    print(f'\nBASELINE.json target "< 1e-3 over 200 steps" ({args.norm}, median over {n} tapes; substep 200 = env-step 20):')
    print('  task            env-step |   f32 build |   f64 build | replica 1e-7 | replica 1e-13 | f32 < 1e-3 | f64 < 1e-3')
    for task, s, a, b, c, d in summary:
        print(f'  {task:15s} {s:8d} | {a:11.2e} | {b:11.2e} | {c:12.2e} | {d:13.2e} | {"yes" if a < 1e-3 else "no":>10s} | {"yes" if b < 1e-3 else "no":>10s}')


if __name__ == '__main__':
    main()
