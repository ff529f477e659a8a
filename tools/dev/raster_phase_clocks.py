"""Phase clocks of a rasteriser workgroup: wall-clock ticks (100 MHz) from the workgroup's start
to the end of phases S, C, T and Q + E, all workgroups of one launch at full occupancy (development tool).
Needs a -DMGX_RASTER_CLOCKS build of the library (the shipped one carries no phase clocks):
  python -c "from magical_amd import _native as n; n.build(force=True, defines=['MGX_RASTER_CLOCKS'], out=n.LIB_PATH.replace('.so', '_clocks.so'))"
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python <this tool>
"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
clk = torch.zeros((N, 16), dtype=torch.int64, device='cuda:0')
env.render_frames(stack, view='ego', layout='stack4'); torch.cuda.synchronize()
env._lib.mgx_engine_debug_raster_clocks(env._engine, C.c_void_p(clk.data_ptr()))
env.render_frames(stack, view='ego', layout='stack4'); torch.cuda.synchronize()
env._lib.mgx_engine_debug_raster_clocks(env._engine, C.c_void_p(0))
c = clk.cpu().numpy().astype(np.float64)
names = (('stage', 0), ('S.bodies', 12), ('S.prims', 13), ('S', 1), ('C', 2), ('T', 3), ('Q+E', 4)) if c[:, 12].any() else (('stage', 0), ('S', 1), ('C', 2), ('T', 3), ('Q+E', 4))
prev = 0
for n, k in names:
    v = np.percentile(c[:, k], 50)
    print('%-9s ends at p50 %6.1f us  (+%5.1f)   p90 %6.1f' % (n, v / 100, (v - prev) / 100, np.percentile(c[:, k], 90) / 100))
    prev = v
print('queued pixels p50 %d p99 %d' % (np.percentile(c[:, 5], 50), np.percentile(c[:, 5], 99)))
