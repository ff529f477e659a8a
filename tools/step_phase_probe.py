"""GPU probe: shader cycles per k_step phase (needs a -DMGX_STEP_PROBE build of libmagical_hip.so; development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import magical_amd

task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-v0'
L = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0', lanes_per_env=L)
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(80, N)).astype(np.int32), device='cuda:0')
for s in range(40):
    env.step(tape[s])
blocks = N * env.lanes_per_env // 64
clk = torch.zeros((blocks, 32), dtype=torch.int64, device='cuda:0')
env._lib.mgx_engine_debug_step_clocks(env._engine, C.c_void_p(clk.data_ptr()))
torch.cuda.synchronize(); t0 = time.perf_counter()
env.substeps(tape[41], 10); torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3
env._lib.mgx_engine_debug_step_clocks(env._engine, None)
c = clk.cpu().numpy().astype(np.float64)
names = ["ph_init_work", "ph_load_state", "ph_integrate", "ph_shapes", "ph_broad", "ph_broad_(unused)", "ph_narrow",
         "ph_arbiters_joints", "solve_begin", "solve_warm_contacts", "solve_warm_pg", "solve_iter_publish",
         "solve_iter_contacts", "solve_iter_pg", "solve_end", "ph_cache_commit", "solve_warm_chain", "solve_iter_chain"]
tot = c[:, :20].sum(axis=1).mean()
print('%s L=%d lds=%dB: launch %.3f ms; instrumented cycles per workgroup %.0f' % (task, env.lanes_per_env, env._lib.mgx_engine_lds_bytes(env._engine, 0), ms, tot))
tt = c[:, :20].sum(axis=1)
print('  per-workgroup total: p50 %.0f p90 %.0f p99 %.0f max %.0f' % tuple(np.percentile(tt, [50, 90, 99, 100])))
w = int(np.argmax(tt))
print('  slowest workgroup:', ', '.join('%s %.0f' % (n, c[w, i]) for i, n in enumerate(names) if c[w, i] > 0.03 * tt[w]))
for i, n in enumerate(names + ['', 'other']):
    if n and c[:, i].mean() > 0:
        print('  %-22s %8.0f cyc  %5.1f %%   (max %8.0f)' % (n, c[:, i].mean(), 100 * c[:, i].mean() / tot, c[:, i].max()))
