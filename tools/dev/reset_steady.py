"""Steady-state reset time of per-env-world variants (repeated resets of one env batch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import magical_amd
N = 4096
for name in ('ClusterColour-TestAll-LoRes4E-v0', 'MatchRegions-TestCountPlus-LoRes4E-v0'):
    env = magical_amd.make(name, n_envs=N, device='cuda:0')
    env.seed(3)
    ts = []
    for r in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.reset(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        a = torch.zeros(N, dtype=torch.int32, device='cuda:0')
        for _ in range(25):          # (a rollout in between, as in use: the retired worlds are freed meanwhile)
            env.step(a)
    print(name, 'resets of %d envs (ms):' % N, ' '.join('%.1f' % t for t in ts), ' episode: %d steps' % env.max_episode_steps)
    env.close()
