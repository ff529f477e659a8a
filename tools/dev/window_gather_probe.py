"""Where the one-rank RCCL gather's ≈ 0.3 ms go in a 20-step window: host stamps after the step loop, after the score upload, after the
collective call returns, after the final synchronize (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.distributed as dist
import magical_amd
from magical_amd.distributed import gather_rollout_results, init_from_env
if 'nogroup' not in sys.argv: init_from_env(backend='nccl', single_process_group=True)
N, K, W = 4096, 20, 5
env = magical_amd.make('MoveToCorner-Demo-LoRes4E-v0', n_envs=N, device='cuda:0')
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(400, N)).astype(np.int32), device='cuda:0')
env.reset()
for s in range(60): env.step(tape[s])
last = torch.zeros(N, dtype=torch.float64, device='cuda:0')
pin = torch.zeros(N, dtype=torch.float64).pin_memory()
if 'nogroup' not in sys.argv: gather_rollout_results(last, N)
torch.cuda.synchronize()
ep = env.max_episode_steps
mode = sys.argv[1] if len(sys.argv) > 1 else 'gather'
rows = []
for rep in range(30):
    if 'ends' in sys.argv:
        clocks = np.full(N, 30, dtype=np.int64); clocks[:1024] = ep - 11 - W
        env.set_episode_steps(clocks)
    for s in range(W): env.step(tape[s])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        obs, rew, done, info = env.step(tape[W + s])
        if done.any(): pin.numpy()[done] = info['eval_score'][done]
    t1 = time.perf_counter()
    last.copy_(pin, non_blocking=True)
    t2 = time.perf_counter()
    if mode == 'gather':
        out = gather_rollout_results(last, N)
    elif mode == 'gather_ev':
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = gather_rollout_results(last, N); e1.record()
    elif mode == 'copy':
        out = last.clone()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    rows.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3])
print('first three windows:', ' | '.join(' '.join('%.2f' % x for x in r) for r in rows[:3]))
r = np.median(np.array(rows), axis=0)
print('%s: host step loop %.2f ms, upload call %.3f, collective call %.3f, final synchronize %.2f, window %.2f ms' % (mode, *r))
