"""Host timeline of a 20-step window with a partial episode end inside: per-step host time and GPU completion lag."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import magical_amd
N, K, W = 4096, 20, 5
env = magical_amd.make('MoveToCorner-Demo-LoRes4E-v0', n_envs=N, device='cuda:0')
ep = env.max_episode_steps
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(400, N)).astype(np.int32), device='cuda:0')
env.reset()
for s in range(ep): env.step(tape[s])
for n_end in (0, 1024):
    acc = np.zeros(K); tot = []
    for rep in range(30):
        clocks = np.full(N, 30, dtype=np.int64); clocks[:n_end] = ep - 11 - W
        env.set_episode_steps(clocks)
        for s in range(W): env.step(tape[s])
        torch.cuda.synchronize()
        ts = [time.perf_counter()]
        evs = []
        for s in range(K):
            env.step(tape[W + s]); ts.append(time.perf_counter())
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        torch.cuda.synchronize(); t_end = time.perf_counter()
        acc += np.diff(ts) * 1e3; tot.append((t_end - ts[0]) * 1e3)
        if rep == 29:
            gpu = [evs[0].elapsed_time(e) for e in evs]
            print('  GPU completion time of each step relative to step 0 (ms):', ' '.join('%.2f' % g for g in gpu))
            print('  per-step GPU deltas:', ' '.join('%.2f' % d for d in np.diff(gpu)))
    print('%d envs end: window median %.2f ms; host ms per step call: %s' % (n_end, np.median(tot), ' '.join('%.2f' % x for x in acc / 30)))
