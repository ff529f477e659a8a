"""GPU probe: does splitting one batch over K engines on K HIP streams (k_step of one part overlapping k_raster of another)
beat one engine over the whole batch?  (development tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
args = [a for a in sys.argv[1:] if not a.startswith('--')]
name = args[0] if args else 'MoveToCorner-Demo-LoRes4E-v0'
JOIN = '--join' in sys.argv      # all parts wait for each other after every step (one caller consuming the whole batch)
N, T, W = 4096, 300, 20
for K in (1, 2, 4, 8):
    n = N // K
    streams = [torch.cuda.Stream() for _ in range(K)]
    envs = []
    for k in range(K):
        with torch.cuda.stream(streams[k]):
            e = magical_amd.make(name, n_envs=n, device='cuda:0'); e.seed(k); e.reset(); envs.append(e)
    acts = [torch.from_numpy(np.random.RandomState(k).randint(0, 18, size=(T + W, n)).astype(np.int32)).cuda() for k in range(K)]
    torch.cuda.synchronize()
    for phase, (a, b) in (('warm', (0, W)), ('timed', (W, W + T))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(a, b):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    envs[k].step(acts[k][s])
            if JOIN and K > 1:
                evs = [torch.cuda.Event() for _ in range(K)]
                for k in range(K): evs[k].record(streams[k])
                for k in range(K):
                    for j in range(K):
                        if j != k: streams[k].wait_event(evs[j])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{name}{" (joined every step)" if JOIN else ""} K={K} engines x {n} envs: {N * T / dt / 1e6:.2f} M env-steps/s, {dt / T * 1e3:.3f} ms per full-batch step', flush=True)
    for e in envs: e.close()
