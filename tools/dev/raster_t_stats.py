"""Phase T of k_raster, wave 0 of every workgroup (MGX_RASTER_PROBE build): mixed tiles walked, items classified, cycles in the
gather and in the classification (development tool)."""
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
q = lambda a: ' '.join('%s %.0f' % (n, np.percentile(a, p)) for n, p in (('p10', 10), ('p50', 50), ('p90', 90), ('max', 100)))
print(task, 'wave 0 of each workgroup, phase T:')
print('  mixed tiles walked (of 36):', q(c[:, 9]))
print('  items classified          :', q(c[:, 10]), ' per mixed tile %.1f' % (c[:, 10].sum() / max(c[:, 9].sum(), 1)))
print('  cycles (s_memtime, 100 MHz): gather', q(c[:, 6]), '| classify', q(c[:, 7]), '| phase T', q(c[:, 8]))
