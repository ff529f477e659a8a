"""cProfile of a WARM episode-end step (scoring + per-env-world reset + first observation) at 4096 envs (development tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, cProfile, pstats
import magical_amd
task = sys.argv[1]
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0')
env.reset()
T = env.max_episode_steps
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(T, N)).astype(np.int32), device='cuda:0')
for ep in range(3):
    for s in range(T - 1): env.step(tape[s])
    torch.cuda.synchronize()
    if ep == 2:
        pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter(); _, _, done, _ = env.step(tape[T - 1]); torch.cuda.synchronize(); t1 = time.perf_counter()
    if ep == 2:
        pr.disable()
    assert done.all()
    print(task, 'episode', ep, 'end step, queue drained: %.2f ms wall' % ((t1 - t0) * 1e3), flush=True)
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
