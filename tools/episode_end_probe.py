"""Host-side cost of an episode-end step (scoring + auto-reset), GPU queue drained first (development tool).
Usage: episode_end_probe.py [task] [n_finishing]   (n_finishing < 4096: only that many envs end, the partial-mask path)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cProfile, pstats
import magical_amd
task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-LoRes4E-v0'
part = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0')
env.reset()
T = env.max_episode_steps
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(2 * T + 2, N)).astype(np.int32), device='cuda:0')
if part:
    clocks = np.zeros(N, dtype=np.int64); clocks[:part] = T // 2
    env.set_episode_steps(clocks)
    T = T // 2
for rep in range(2):
    for s in range(T - 1): env.step(tape[s])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); _, _, done, _ = env.step(tape[T - 1]); torch.cuda.synchronize(); t1 = time.perf_counter()
    assert done.sum() == (part or N), done.sum()
    print('episode-end step (%d envs finish), queue drained: %.2f ms wall' % (done.sum(), (t1 - t0) * 1e3))
    t0 = time.perf_counter(); env.step(tape[T]); torch.cuda.synchronize(); t1 = time.perf_counter()
    print('ordinary step, queue drained: %.2f ms wall' % ((t1 - t0) * 1e3))
    if part:
        clocks = env._steps.copy(); clocks[:part] = env.max_episode_steps - T; env.set_episode_steps(clocks)
if part:
    for s in range(T - 1): env.step(tape[s])
else:
    for s in range(T + 1, 2 * T - 1): env.step(tape[s])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
_, _, done, _ = env.step(tape[2 * T - 1]); torch.cuda.synchronize()
pr.disable()
print('profiled step: %d envs finished' % done.sum())
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
