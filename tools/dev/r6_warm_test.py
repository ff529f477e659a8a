import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch, magical_amd
for name in ['ClusterColour-TestAll-LoRes4E-v0', 'ClusterColour-TestAll-LoRes4E-v0', 'MatchRegions-TestAll-LoRes4E-v0', 'MatchRegions-TestAll-LoRes4E-v0']:
    env = magical_amd.make(name, n_envs=4096, device='cuda:0')
    T = env.max_episode_steps
    tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(T, 4096)).astype(np.int32), device='cuda:0')
    env.seed(0); env.reset(); torch.cuda.synchronize()
    t0 = time.perf_counter(); marks = []
    for s in range(T):
        _, _, done, info = env.step(tape[s])
        if s in (0, 1, 4, 19, T - 2, T - 1):
            torch.cuda.synchronize(); marks.append((s, round((time.perf_counter() - t0) * 1e3, 1)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, 'episode', round(dt * 1e3, 1), 'ms =', round(4096 * T / dt / 1e6, 2), 'M; cumulative ms at steps', marks, flush=True)
    env.close()
