"""GPU probe: fused vs two-call env-step at large batch sizes (no episode end inside), with hand-off statistics (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
for n in (32768, 65536, 131072):
    for ov in (True, False):
        e = magical_amd.make('MoveToCorner-Demo-LoRes4E-v0', n_envs=n, device='cuda:0', overlap=ov, max_episode_steps=100000); e.reset()
        tape = torch.as_tensor(np.random.RandomState(2).randint(0, 18, size=(60, n)).astype(np.int32), device='cuda:0')
        for s in range(20): e.step(tape[s])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(20, 60): e.step(tape[s])
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 40
        print('%6d envs  fused=%s  %.3f ms/step  %.2f M env-steps/s  hand-off (gave up, timeouts) %s' % (n, ov, t * 1e3, n / t / 1e6, e.handoff_stats() if ov else '-'), flush=True)
        e.close()
