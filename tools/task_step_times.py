"""GPU probe: ms per env-step of every Demo task at 4096 envs (LoRes4E, no episode end inside), with the kernels' LDS footprints (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
T = 300
for task in ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']:
    name = f'{task}-Demo-LoRes4E-v0'
    a = magical_amd.make(name, n_envs=4096, device='cuda:0', max_episode_steps=100000)
    a.reset()
    tape = torch.as_tensor(np.random.RandomState(2).randint(0, 18, size=(T, 4096)).astype(np.int32), device='cuda:0')
    for s in range(40): a.step(tape[s])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(40, T): a.step(tape[s])
    torch.cuda.synchronize(); ta = time.perf_counter() - t0
    d, to = a.handoff_stats()
    print('%-30s %.3f ms/step  %5.2f M env-steps/s  LDS k_step %6d B  k_raster %6d B  gave up %d, timeouts %d' % (
        name, ta / (T - 40) * 1e3, 4096 * (T - 40) / ta / 1e6, a._lib.mgx_engine_lds_bytes(a._engine, 0), a._lib.mgx_engine_lds_bytes(a._engine, 1), d, to), flush=True)
    a.close()
