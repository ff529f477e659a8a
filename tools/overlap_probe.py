"""GPU probe: do k_step and k_raster overlap usefully when issued on two streams?  (timing only; development tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd

task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-v0'
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0')
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(200, N)).astype(np.int32), device='cuda:0')
for s in range(30):
    env.step(tape[s])
stack = torch.zeros((N, 96, 96, 12), dtype=torch.uint8, device='cuda:0')
K = 50
def run(mode):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        if mode == 'serial':
            env.substeps(tape[30 + k], 10); env.render_frames(stack, view='ego', layout='stack4')
        elif mode == 'step':
            env.substeps(tape[30 + k], 10)
        elif mode == 'raster':
            env.render_frames(stack, view='ego', layout='stack4')
        else:
            with torch.cuda.stream(s1): env.substeps(tape[30 + k], 10)
            with torch.cuda.stream(s2): env.render_frames(stack, view='ego', layout='stack4')
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
for mode in ('serial', 'step', 'raster', 'overlap', 'serial', 'overlap'):
    print('%-8s %.3f ms per step' % (mode, run(mode)))
