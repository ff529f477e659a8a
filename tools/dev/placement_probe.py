"""Where the fused env-step's workgroups run: HW_ID / XCC_ID of every step workgroup and of every rasteriser wavefront
(-DMGX_RASTER_CLOCKS build, see fused_timeline.py).  Answers: do a CU's four step wavefronts sit on four SIMDs, and how many
rasteriser workgroups are resident beside them?
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/placement_probe.py [task]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import magical_amd
task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-LoRes4E-v0'
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0', max_episode_steps=100000)
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(100, N)).astype(np.int32), device='cuda:0')
for s in range(64):
    env.step(tape[s])
clk = torch.zeros((N, 16), dtype=torch.int64, device='cuda:0')
sclk = torch.zeros((N, 4), dtype=torch.int64, device='cuda:0')      # (more rows than step workgroups)
env._lib.mgx_engine_debug_raster_clocks(env._engine, C.c_void_p(clk.data_ptr()))
env._lib.mgx_engine_debug_step_clocks(env._engine, C.c_void_p(sclk.data_ptr()))
torch.cuda.synchronize()
env.step(tape[65]); torch.cuda.synchronize()
env._lib.mgx_engine_debug_raster_clocks(env._engine, None)
env._lib.mgx_engine_debug_step_clocks(env._engine, None)
s = sclk.cpu().numpy(); s = s[s[:, 2] != 0]
c = clk.cpu().numpy()
def fields(hw, xcc):
    return dict(wave=hw & 15, simd=(hw >> 4) & 3, cu=(hw >> 8) & 15, sh=(hw >> 12) & 1, se=(hw >> 13) & 7, xcc=xcc & 15)
def cu_key(f): return (int(f['xcc']), int(f['se']), int(f['sh']), int(f['cu']))
t0 = min(s[:, 2].min(), c[:, 9].min())
print(f'{len(s)} step workgroups; start p0 {(s[:,2].min()-t0)/100:.1f} max {(s[:,2].max()-t0)/100:.1f} us; end p10 {(np.percentile(s[:,3],10)-t0)/100:.0f} p50 {(np.percentile(s[:,3],50)-t0)/100:.0f} p90 {(np.percentile(s[:,3],90)-t0)/100:.0f} max {(s[:,3].max()-t0)/100:.0f} us')
per_cu = collections.defaultdict(list)
for row in s:
    f = fields(int(row[0]), int(row[1])); per_cu[cu_key(f)].append(int(f['simd']))
print(f'CUs holding step workgroups: {len(per_cu)}; step workgroups per CU: {collections.Counter(len(v) for v in per_cu.values())}')
pat = collections.Counter(tuple(sorted(collections.Counter(v).values(), reverse=True)) for v in per_cu.values())
print('step wavefronts per SIMD within a CU (sorted counts): ', dict(pat))
# rasteriser: SIMDs of a workgroup's four wavefronts
rw = c[:, [6, 7, 8, 10]]
simds = (rw >> 4) & 3
print('distinct SIMDs among a rasteriser workgroup\'s 4 wavefronts:', dict(collections.Counter(len(set(r.tolist())) for r in simds)))
start = (c[:, 9] - t0) / 100.0; end = start + c[:, 4] / 100.0
for tt in (20, 50, 100, 150, 200, 300):
    res = (start <= tt) & (end > tt)
    cus = collections.Counter(cu_key(fields(int(r[0]) & 0xFFFFFFFF, int(r[0]) >> 32)) for r in rw[res])
    step_res = (s[:, 2] - t0 <= tt * 100) & (s[:, 3] - t0 > tt * 100)
    print(f't={tt:3d} us: step workgroups resident {int(step_res.sum()):4d}; rasteriser workgroups resident {int(res.sum()):4d} on {len(cus)} CUs; per CU {dict(sorted(collections.Counter(cus.values()).items()))}')
# beside a CU's step wavefronts at t = 50 us: rasteriser wavefronts per SIMD against step wavefronts per SIMD
tt = 50
res = (start <= tt) & (end > tt)
occ = collections.defaultdict(lambda: [0, 0])
for row in s[(s[:, 2] - t0 <= tt * 100) & (s[:, 3] - t0 > tt * 100)]:
    f = fields(int(row[0]), int(row[1])); occ[cu_key(f) + (int(f['simd']),)][0] += 1
for r in rw[res]:
    for w in r:
        f = fields(int(w) & 0xFFFFFFFF, int(w) >> 32); occ[cu_key(f) + (int(f['simd']),)][1] += 1
print('t=50 us, SIMDs by (step wavefronts, rasteriser wavefronts):', dict(sorted(collections.Counter(tuple(v) for v in occ.values()).items())))
if os.environ.get('PLACEMENT_DUMP'):
    np.savez(os.environ['PLACEMENT_DUMP'], step=s, raster=c, cost=np.zeros(1))
# the dispatcher's order: CU of the step workgroups blockIdx = x, x + 8, ... of XCD x
for x in (0, 1):
    seq = []
    for b in range(x, len(s), 8):
        f = fields(int(s[b, 0]), int(s[b, 1])); seq.append('%d:%d.%d.%d/%d' % (f['xcc'], f['se'], f['sh'], f['cu'], f['simd']))
    print(f'blockIdx {x} + 8 i ->', ' '.join(seq[:72]))
