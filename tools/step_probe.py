"""GPU probe: k_step launch time vs solver iterations / substeps / lanes (development tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import magical_amd
from magical_amd import _native as nat

task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-v0'
N = 4096
for L in ((16, 32) if 'Cluster' in task else (4, 16)):
    env = magical_amd.make(task, n_envs=N, device='cuda:0', lanes_per_env=L)
    env.reset()
    tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(40, N)).astype(np.int32), device='cuda:0')
    for s in range(30):
        env.step(tape[s])
    def t_sub(nsub, n=10):
        a = tape[35]
        for _ in range(2): env.substeps(a, nsub)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): env.substeps(a, nsub)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(f'{task} L={L} lds={env._lib.mgx_engine_lds_bytes(env._engine, 0)}B')
    for it in (10, 5, 0):
        env._lib.mgx_engine_debug_iterations(env._engine, it)
        print('  iterations=%2d: 10 substeps %.3f ms, 1 substep %.3f ms, 0 substeps (load/store only) %.3f ms' % (it, t_sub(10), t_sub(1), t_sub(0)))
    env._lib.mgx_engine_debug_iterations(env._engine, -1)
    env.close()
