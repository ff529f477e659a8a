"""GPU probe (needs a -DMGX_STEP_PROBE build): how well does an env's contact count BEFORE a step predict its workgroup's
duration?  Basis for launching the light envs' step + render ahead of the heavy envs' tail (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import magical_amd
task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-v0'
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0')
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(80, N)).astype(np.int32), device='cuda:0')
for s in range(40):
    env.step(tape[s])
L = env.lanes_per_env
blocks = N * L // 64
epb = 64 // L
for rep in range(3):
    ncache = env.state_i[1].cpu().numpy().copy()
    clk = torch.zeros((blocks, 32), dtype=torch.int64, device='cuda:0')
    env._lib.mgx_engine_debug_step_clocks(env._engine, C.c_void_p(clk.data_ptr()))
    env.substeps(tape[41 + rep], 10); torch.cuda.synchronize()
    env._lib.mgx_engine_debug_step_clocks(env._engine, None)
    tt = clk.cpu().numpy().astype(np.float64)[:, :20].sum(axis=1)
    wg = np.arange(blocks)
    wgm = (wg & 7) * (blocks >> 3) + (wg >> 3) if blocks % 8 == 0 else wg          # the kernel's XCD remap
    pred = ncache[(wgm[:, None] * epb + np.arange(epb)[None, :])].max(axis=1)
    after = env.state_i[1].cpu().numpy()[(wgm[:, None] * epb + np.arange(epb)[None, :])].max(axis=1)
    print(f'{task} rep {rep}: wg cycles p50 {np.percentile(tt,50):.0f} p90 {np.percentile(tt,90):.0f} max {tt.max():.0f}')
    for k in range(0, int(pred.max()) + 1):
        sel = pred == k
        if sel.any():
            print('   max ncache before = %d: %4d wgs, cycles p50 %7.0f p90 %7.0f max %7.0f' % (k, sel.sum(), np.percentile(tt[sel], 50), np.percentile(tt[sel], 90), tt[sel].max()))
    env.step(tape[50 + rep])
