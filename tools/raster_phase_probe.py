"""GPU probe: run k_raster truncated after each phase (S, C, T, Q) so that `rocprofv3 --pmc` attributes instruction
counts and durations to phases by differencing (development tool).  Dispatch order: for stop in 1,2,3,4: 4 launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd

task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-v0'
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0')
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(60, N)).astype(np.int32), device='cuda:0')
for s in range(60):
    env.step(tape[s])
stack = torch.zeros((N, 96, 96, 12), dtype=torch.uint8, device='cuda:0')
for stop in (1, 2, 3, 4, 9):
    env._lib.mgx_engine_debug_raster_stop(env._engine, stop)
    env.render_frames(stack, view='ego', layout='stack4'); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        env.render_frames(stack, view='ego', layout='stack4')
    torch.cuda.synchronize()
    ts = (time.perf_counter() - t0) / 4 * 1e3
    frame = torch.zeros((N, 96, 96, 3), dtype=torch.uint8, device='cuda:0')
    env.render_frames(frame, view='ego', layout='frame'); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        env.render_frames(frame, view='ego', layout='frame')
    torch.cuda.synchronize()
    print('stop after phase %d: stack4 %.3f ms   frame %.3f ms' % (stop, ts, (time.perf_counter() - t0) / 4 * 1e3))
env._lib.mgx_engine_debug_raster_stop(env._engine, 0)
