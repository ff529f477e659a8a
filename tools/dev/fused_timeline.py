"""Timeline of one fused env-step: when raster workgroups start, how long they wait for their env, how long they work.
Needs a -DMGX_RASTER_CLOCKS build of the library (the shipped one carries no phase clocks):
  python -c "from magical_amd import _native as n; n.build(force=True, defines=['MGX_RASTER_CLOCKS'], out=n.LIB_PATH.replace('.so', '_clocks.so'))"
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python <this tool>
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import magical_amd
task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-LoRes4E-v0'
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0', max_episode_steps=100000)
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(100, N)).astype(np.int32), device='cuda:0')
for s in range(60):
    env.step(tape[s])
clk = torch.zeros((N, 16), dtype=torch.int64, device='cuda:0')
for fused in (True, False):
    env.overlap = fused
    for s in range(60, 64):
        env.step(tape[s])
    env._lib.mgx_engine_debug_raster_clocks(env._engine, C.c_void_p(clk.data_ptr()))
    torch.cuda.synchronize()
    env.step(tape[65]); torch.cuda.synchronize()
    env._lib.mgx_engine_debug_raster_clocks(env._engine, None)
    c = clk.cpu().numpy().astype(np.float64)
    t0 = c[:, 9].min()
    start = (c[:, 9] - t0) / 100.0            # us (100 MHz constant clock)
    got = start + c[:, 0] / 100.0             # after the hand-off wait + setup start
    end = start + c[:, 4] / 100.0
    work = end - got
    print(f'fused={fused}: raster WG start  p0 {start.min():.0f} p25 {np.percentile(start,25):.0f} p50 {np.percentile(start,50):.0f} p75 {np.percentile(start,75):.0f} max {start.max():.0f} us')
    print(f'   wait (start -> CLK0) p50 {np.percentile(got-start,50):.0f} p90 {np.percentile(got-start,90):.0f} max {(got-start).max():.0f} us;  work p10 {np.percentile(work,10):.0f} p50 {np.percentile(work,50):.0f} p90 {np.percentile(work,90):.0f} us;  last end {end.max():.0f} us')
    # how many raster workgroups are working (between got and end) over time
    for tt in range(0, int(end.max()) + 50, 50):
        active = int(((got <= tt) & (end > tt)).sum()); waiting = int(((start <= tt) & (got > tt)).sum()); done = int((end <= tt).sum())
        print(f'   t={tt:4d} us: working {active:5d}  waiting {waiting:5d}  done {done:5d}')
