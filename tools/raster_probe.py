"""GPU probe: k_raster launch time vs state / layout / view (development tool).
Needs a -DMGX_RASTER_CLOCKS build of the library (the shipped one carries no phase clocks):
  python -c "from magical_amd import _native as n; n.build(force=True, defines=['MGX_RASTER_CLOCKS'], out=n.LIB_PATH.replace('.so', '_clocks.so'))"
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python <this tool>
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import magical_amd

def timeit(env, out, view, layout, fill=None, n=20):
    for _ in range(3):
        env.render_frames(out, view=view, layout=layout, fill_mask=fill)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        env.render_frames(out, view=view, layout=layout, fill_mask=fill)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-v0'
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0')
env.reset()
if len(sys.argv) > 2:
    env._lib.mgx_engine_debug_raster_waves(env._engine, int(sys.argv[2]))
print('LDS per workgroup: k_raster %d B, k_step %d B' % (env._lib.mgx_engine_lds_bytes(env._engine, 1), env._lib.mgx_engine_lds_bytes(env._engine, 0)))
frame = torch.zeros((N, 96, 96, 3), dtype=torch.uint8, device='cuda:0')
stack = torch.zeros((N, 96, 96, 12), dtype=torch.uint8, device='cuda:0')
ones = torch.ones(N, dtype=torch.uint8, device='cuda:0')
print('reset state: ego frame %.3f ms, ego stack4 %.3f ms, ego stack4(fill) %.3f, allo frame %.3f ms' % (
    timeit(env, frame, 'ego', 'frame'), timeit(env, stack, 'ego', 'stack4'), timeit(env, stack, 'ego', 'stack4', ones), timeit(env, frame, 'allo', 'frame')))
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(60, N)).astype(np.int32), device='cuda:0')
for s in range(60):
    env.step(tape[s])
    if s in (0, 4, 19, 59):
        print('after %d steps: ego frame %.3f ms, ego stack4 %.3f ms, allo frame %.3f ms' % (
            s + 1, timeit(env, frame, 'ego', 'frame'), timeit(env, stack, 'ego', 'stack4'), timeit(env, frame, 'allo', 'frame')))
clk = torch.zeros((N, 16), dtype=torch.int64, device='cuda:0')
env._lib.mgx_engine_debug_raster_clocks(env._engine, C.c_void_p(clk.data_ptr()))
env.render_frames(stack, view='ego', layout='stack4'); torch.cuda.synchronize()
env._lib.mgx_engine_debug_raster_clocks(env._engine, None)
c = clk.cpu().numpy().astype(np.float64)
names = ['stage tmpl', 'setup (S)', 'tile classify (C)', 'pixel pass (T)', 'queue resolve (Q)']
prev = 0
for i, nme in enumerate(names):
    print('  %-20s cumulative %.1f us (100 MHz wall clock)  delta mean %.1f us  max %.1f us' % (nme, c[:, i].mean() / 100, (c[:, i] - (c[:, i - 1] if i else 0)).mean() / 100, (c[:, i] - (c[:, i - 1] if i else 0)).max() / 100))
print('  queued pixels per env: mean %.0f max %.0f' % (c[:, 5].mean(), c[:, 5].max()))
if c[:, 8].max() > 0:   # MGX_RASTER_PROBE build: wave 0 of each block, shader cycles
    print('  wave 0 phase T: total %.0f cyc; gather %.0f cyc, classify %.0f cyc, mixed tiles %.1f, items %.0f' % (
        c[:, 8].mean(), c[:, 6].mean(), c[:, 7].mean(), c[:, 9].mean(), c[:, 10].mean()))
    print('  phase Q, shader cycles per wavefront (first round): waves 0..3 mean %s, max %s; phase E (wave 0 incl. barrier wait) mean %.0f max %.0f' % (
        np.round(c[:, 11:15].mean(0)), c[:, 11:15].max(0), c[:, 15].mean(), c[:, 15].max()))
# copy bandwidth reference
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): stack.copy_(stack + 0)
torch.cuda.synchronize(); print('torch read+write of the stack tensor x2: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
