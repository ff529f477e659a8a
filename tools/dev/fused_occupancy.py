"""Both kernels of one fused env-step on one time axis (-DMGX_RASTER_CLOCKS build, see fused_timeline.py): step workgroups resident /
finished and rasteriser workgroups waiting / working / finished every 50 us, per-CU co-residency, and the LDS a CU has in use.
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/fused_occupancy.py [task] [dt_us]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import magical_amd
task = sys.argv[1] if len(sys.argv) > 1 else 'ClusterColour-Demo-LoRes4E-v0'
dt = int(sys.argv[2]) if len(sys.argv) > 2 else 50
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0', max_episode_steps=100000)
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(100, N)).astype(np.int32), device='cuda:0')
for s in range(64):
    env.step(tape[s])
lds_s, lds_r = env._lib.mgx_engine_lds_bytes(env._engine, 0), env._lib.mgx_engine_lds_bytes(env._engine, 1)
print(f'{task}: lanes/env {env.lanes_per_env}, LDS k_step {lds_s} B, k_raster {lds_r} B')
for fused in (True, False):
    env.overlap = fused
    for s in range(64, 68):
        env.step(tape[s])
    clk = torch.zeros((N, 16), dtype=torch.int64, device='cuda:0')
    sclk = torch.zeros((N, 4), dtype=torch.int64, device='cuda:0')
    env._lib.mgx_engine_debug_raster_clocks(env._engine, C.c_void_p(clk.data_ptr()))
    env._lib.mgx_engine_debug_step_clocks(env._engine, C.c_void_p(sclk.data_ptr()))
    torch.cuda.synchronize()
    env.step(tape[70]); torch.cuda.synchronize()
    env._lib.mgx_engine_debug_raster_clocks(env._engine, None)
    env._lib.mgx_engine_debug_step_clocks(env._engine, None)
    s_ = sclk.cpu().numpy(); s_ = s_[s_[:, 2] != 0]
    c = clk.cpu().numpy().astype(np.float64)
    t0 = min(s_[:, 2].min(), c[:, 9].min())
    sb, se = (s_[:, 2] - t0) / 100.0, (s_[:, 3] - t0) / 100.0
    rs = (c[:, 9] - t0) / 100.0; rg = rs + c[:, 0] / 100.0; re = rs + c[:, 4] / 100.0
    print(f'fused={fused}: {len(s_)} step workgroups, duration p10 {np.percentile(se - sb, 10):.0f} p50 {np.percentile(se - sb, 50):.0f} p90 {np.percentile(se - sb, 90):.0f} max {(se - sb).max():.0f} us, '
          f'last start {sb.max():.0f}, last end {se.max():.0f} us; rasteriser workgroup work p10 {np.percentile(re - rg, 10):.0f} p50 {np.percentile(re - rg, 50):.0f} p90 {np.percentile(re - rg, 90):.0f} us, '
          f'first start {rs.min():.0f}, last end {re.max():.0f} us')
    print('    t us | step: resident  done | raster: waiting working  done | LDS in use per CU (mean KB)')
    for tt in range(0, int(max(se.max(), re.max())) + dt, dt):
        sres = int(((sb <= tt) & (se > tt)).sum()); sdone = int((se <= tt).sum())
        rwait = int(((rs <= tt) & (rg > tt)).sum()); rwork = int(((rg <= tt) & (re > tt)).sum()); rdone = int((re <= tt).sum())
        print(f'  {tt:6d} | {sres:14d} {sdone:5d} | {rwait:15d} {rwork:7d} {rdone:5d} | {(sres * lds_s + (rwait + rwork) * lds_r) / 256 / 1024:6.1f}')
env.close()
