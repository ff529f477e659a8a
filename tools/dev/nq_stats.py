"""Undecided pixels per env-frame (the rasteriser's LDS queue load) for the 8 Demo tasks over a random rollout.
Needs a -DMGX_RASTER_CLOCKS build of the library (the shipped one carries no phase clocks):
  python -c "from magical_amd import _native as n; n.build(force=True, defines=['MGX_RASTER_CLOCKS'], out=n.LIB_PATH.replace('.so', '_clocks.so'))"
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python <this tool>
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import magical_amd
N = 2048
for task in ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']:
    for variant in ('Demo', 'TestAll'):
        try:
            env = magical_amd.make(f'{task}-{variant}-LoRes4E-v0', n_envs=N, device='cuda:0')
        except Exception as ex:
            continue
        env.reset()
        env.overlap = False
        tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(60, N)).astype(np.int32), device='cuda:0')
        clk = torch.zeros((N, 16), dtype=torch.int64, device='cuda:0')
        env._lib.mgx_engine_debug_raster_clocks(env._engine, C.c_void_p(clk.data_ptr()))
        allq = []
        for s in range(60):
            env.step(tape[s])
            if s % 6 == 5:
                torch.cuda.synchronize(); allq.append(clk[:, 5].cpu().numpy().copy())
        env._lib.mgx_engine_debug_raster_clocks(env._engine, None)
        q = np.concatenate(allq)
        print(f'{task}-{variant}: n_prims {env._info("n_prims")}  queued pixels p50 {np.percentile(q,50):.0f} p90 {np.percentile(q,90):.0f} p99 {np.percentile(q,99):.0f} p99.9 {np.percentile(q,99.9):.0f} max {q.max()}  raster LDS {env._lib.mgx_engine_lds_bytes(env._engine, 1)}')
        env.close()
