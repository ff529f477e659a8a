import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, magical_amd
n, t = 3, 9
tape = np.random.RandomState(23).randint(0, 18, size=(t, n)).astype(np.int32)
env = magical_amd.make('MatchRegions-Demo-v0', n_envs=n, device='cuda:0')
env.reset()
for s in range(t): env.step(tape[s])
ref = {}
for qcap, ecap in ((1 << 20, 1 << 20), (64, 1 << 20), (32, 1<<20), (8, 1<<20), (1, 1 << 20), (1 << 20, 1), (1<<20, 4)):
    env._lib.mgx_engine_debug_raster_qcap(env._engine, qcap); env._lib.mgx_engine_debug_raster_ecap(env._engine, ecap)
    for view in ('ego', 'allo'):
        frame = torch.zeros((n, 96, 96, 3), dtype=torch.uint8, device='cuda:0')
        env.render_frames(frame, view=view, layout='frame'); f = frame.cpu().numpy()
        if view not in ref: ref[view] = f
        d = (f != ref[view]).any(-1)
        print('qcap', qcap, 'ecap', ecap, view, 'pixels differing', int(d.sum()), np.argwhere(d)[:4].tolist(), flush=True)
stacks = []
for qcap, ecap in ((1 << 20, 1 << 20), (64, 1 << 20), (1, 1 << 20), (1 << 20, 1)):
    env._lib.mgx_engine_debug_raster_qcap(env._engine, qcap); env._lib.mgx_engine_debug_raster_ecap(env._engine, ecap)
    stack = torch.arange(n * 96 * 96 * 12, device='cuda:0').remainder(251).to(torch.uint8).reshape(n, 96, 96, 12)
    old = stack.cpu().numpy().copy()
    env.render_frames(stack, view='ego', layout='stack4')
    got = stack.cpu().numpy()
    d = (got[..., :9] != old[..., 3:]).any(-1)
    print('stack4 qcap', qcap, 'ecap', ecap, 'shift errors', int(d.sum()), np.argwhere(d)[:6].tolist(), flush=True)
    d2 = (got[..., 9:] != ref['ego']).any(-1)
    print('   new frame bytes differing from the frame layout', int(d2.sum()), np.argwhere(d2)[:6].tolist(), flush=True)
