"""GPU probe: wall time of 20-step windows with no / a partial / a full-batch episode end inside (development tool)."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
N, K, W = 4096, 20, 5
env = magical_amd.make('MoveToCorner-Demo-LoRes4E-v0', n_envs=N, device='cuda:0', overlap=os.environ.get('OVERLAP', '1') == '1')
ep = env.max_episode_steps
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(400, N)).astype(np.int32), device='cuda:0')
env.reset()
for s in range(ep): env.step(tape[s])
def window(n_end):
    clocks = np.full(N, 30, dtype=np.int64); clocks[:n_end] = ep - 11 - W
    env.set_episode_steps(clocks)
    for s in range(W): env.step(tape[s])
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for s in range(K):
        env.step(tape[W + s]); ts.append(time.perf_counter())
    torch.cuda.synchronize(); ts.append(time.perf_counter())
    w = (ts[-1] - ts[0]) * 1e3
    if w > 18: print('   slow window %.1f ms: handoff %s; host ms per step %s; final sync %.2f' % (w, env.handoff_stats(), ' '.join('%.2f' % x for x in np.diff(ts[:-1]) * 1e3), (ts[-1] - ts[-2]) * 1e3))
    return w
for n_end in (0, 1024, N):
    r = [window(n_end) for _ in range(40)]
    print('%4d envs end in the window: median %.2f ms, max %.2f, windows over 18 ms: %d of %d' % (n_end, np.median(r), max(r), sum(x > 18 for x in r), len(r)))
