"""Host-side cost of an episode-end step (scoring + auto-reset), GPU queue drained first (development tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cProfile, pstats
import magical_amd
task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-LoRes4E-v0'
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0')
env.reset()
T = env.max_episode_steps
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(2 * T + 2, N)).astype(np.int32), device='cuda:0')
for s in range(T - 1): env.step(tape[s])
torch.cuda.synchronize()
t0 = time.perf_counter(); _, _, done, _ = env.step(tape[T - 1]); torch.cuda.synchronize(); t1 = time.perf_counter()
assert done.all()
print('episode-end step, queue drained: %.2f ms wall' % ((t1 - t0) * 1e3))
for s in range(T, 2 * T - 1): env.step(tape[s])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
env.step(tape[2 * T - 1]); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
