"""GPU-box probe: where does a reset of N envs of a Test* variant spend its host time?  (cProfile, top entries)"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
name = sys.argv[1] if len(sys.argv) > 1 else 'MatchRegions-TestCountPlus-LoRes4E-v0'
N = 4096
GAP = int(sys.argv[2]) if len(sys.argv) > 2 else 10        # env-steps between two resets (an episode is 40 - 240)
env = magical_amd.make(name, n_envs=N, device='cuda:0')
env.seed(3); env.reset(); torch.cuda.synchronize()
for _ in range(3):          # steady state: the first resets of a process also warm the allocator and the staging buffers up
    for _ in range(10): env.step(torch.zeros(N, dtype=torch.int32, device='cuda:0'))
    env.reset(); torch.cuda.synchronize()
idx = np.arange(N)
# time the native calls of the reset separately (cProfile does not see inside ctypes)
native_ms = {}
class _Timed:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith('mgx_engine_set_env_variants') and 'randomise' not in name and 'reset' not in name:
            return fn
        def call(*a):
            t = time.perf_counter(); r = fn(*a); native_ms[name] = native_ms.get(name, 0.0) + (time.perf_counter() - t) * 1e3
            return r
        return call
env._lib = _Timed(env._lib)
# steady state, without the profiler: seven resets (ten env-steps between them), wall time and native calls of each
walls = []
for rep in range(7):
    for _ in range(GAP): env.step(torch.zeros(N, dtype=torch.int32, device='cuda:0'))
    torch.cuda.synchronize()
    native_ms.clear()
    t0 = time.perf_counter()
    env._reset_envs(idx, None); torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) * 1e3)
    print('  reset %d: %.1f ms' % (rep, walls[-1]), 'native calls (ms):', {k: round(v, 1) for k, v in native_ms.items()})
print(name, 'steady-state reset of %d envs, %d env-steps apart: median %.1f ms, min %.1f, max %.1f over %d resets' % (N, GAP, np.median(walls), min(walls), max(walls), len(walls)))
native_ms.clear()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
env._reset_envs(idx, None); torch.cuda.synchronize()
pr.disable()
print(name, 'reset of %d envs under cProfile: %.1f ms' % (N, (time.perf_counter() - t0) * 1e3), 'native calls (ms):', {k: round(v, 1) for k, v in native_ms.items()})
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
for _ in range(3):
    t0 = time.perf_counter(); env.step(torch.zeros(N, dtype=torch.int32, device='cuda:0')); torch.cuda.synchronize()
print('step %.2f ms' % ((time.perf_counter() - t0) * 1e3))
env.set_timing(1)
for _ in range(20): env.step(torch.zeros(N, dtype=torch.int32, device='cuda:0'))
torch.cuda.synchronize()
print('k_step %.3f ms  k_raster %.3f ms' % (env.read_timing('step').mean(), env.read_timing('raster').mean()))
