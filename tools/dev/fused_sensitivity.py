"""How much of the fused env-step is the physics?  Times step+render at 4096 envs with the solver's iteration count cut down
(mgx_engine_debug_iterations: fewer iterations = a shorter k_step, same rasteriser work)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import magical_amd
task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-LoRes4E-v0'
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0', max_episode_steps=100000)
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(400, N)).astype(np.int32), device='cuda:0')
for it in (10, 5, 2, 0):
    env._lib.mgx_engine_debug_iterations(env._engine, it)
    for fused in (True, False):
        env.overlap = fused
        for s in range(30):
            env.step(tape[s])
        env.set_timing(4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(30, 330):
            env.step(tape[s])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 300 * 1e3
        print(f'iterations {it:2d} fused {fused}: {ms:.3f} ms/step, k_step {env.read_timing("step").mean():.3f} k_raster {env.read_timing("render").mean():.3f}')
        env.set_timing(0)
