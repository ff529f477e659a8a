"""Where does the wall time of env.step() go? (development tool)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cProfile, pstats
import magical_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = magical_amd.make('MoveToCorner-Demo-LoRes4E-v0', n_envs=N, device='cuda:0')
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(400, N)).astype(np.int32), device='cuda:0')
for s in range(20): env.step(tape[s])
torch.cuda.synchronize()
# host time per call without waiting for the GPU
t_host = []; t_done = []
for s in range(20, 260):
    t0 = time.perf_counter(); _, _, done, _ = env.step(tape[s]); dt = time.perf_counter() - t0
    (t_done if done.any() else t_host).append(dt)
torch.cuda.synchronize()
print('host time per step() call: normal %.1f us (n=%d), episode-end steps %.1f ms (n=%d)' % (np.mean(t_host) * 1e6, len(t_host), np.mean(t_done) * 1e3, len(t_done)))
t0 = time.perf_counter()
for s in range(260, 339): env.step(tape[s])      # no episode end inside (ends at multiples of 80: 320?) 
torch.cuda.synchronize(); print('wall per step over 79 steps incl. possibly one episode end: %.3f ms' % ((time.perf_counter() - t0) / 79 * 1e3))
pr = cProfile.Profile(); pr.enable()
for s in range(340, 400): env.step(tape[s])
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
