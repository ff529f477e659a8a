"""GPU probe: k_raster's workgroups-per-CU variant (3 / 4 / 5) forced per task vs the engine's own choice (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
N, T = 4096, 120
for task in sys.argv[1:] or ['MakeLine', 'FindDupe', 'MatchRegions', 'ClusterColour']:
    for waves in (0, 3, 4, 5):
        env = magical_amd.make(f'{task if "-" in task else task + "-Demo"}-LoRes4E-v0', n_envs=N, device='cuda:0')
        if waves:
            env._lib.mgx_engine_debug_raster_waves(env._engine, waves)
        env.reset()
        tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(T + 10, N)).astype(np.int32), device='cuda:0')
        for s in range(10): env.step(tape[s])
        env.set_timing(4); torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(10, 10 + T): env.step(tape[s])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        r = env.read_timing('render')
        print(f'{task} waves {waves or "auto"}: {N * T / dt / 1e6:.2f} M env-steps/s, k_raster {float(np.mean(r)):.3f} ms', flush=True)
        env.close()
