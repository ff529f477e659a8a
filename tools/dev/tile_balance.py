"""How evenly do the MIXED tiles (more than one colour: a proxy for the tiles phase T classifies pixel by pixel) spread over the four
wavefronts of a rasteriser workgroup under the shipped assignment (tile = wave + 4 i), and under a balanced one?  (development tool)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import magical_amd
task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner-Demo-LoRes4E-v0'
N = 4096
env = magical_amd.make(task, n_envs=N, device='cuda:0')
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(60, N)).astype(np.int32), device='cuda:0')
for s in range(60):
    obs, *_ = env.step(tape[s])
f = obs[..., 9:12].to(torch.int32)                                   # newest frame [N, 96, 96, 3]
key = (f[..., 0] | (f[..., 1] << 8) | (f[..., 2] << 16)).reshape(N, 24, 4, 6, 16).permute(0, 1, 3, 2, 4).reshape(N, 144, 64)
mixed = (key.max(dim=2).values != key.min(dim=2).values).cpu().numpy()          # [N, 144], tile = row * 6 + col
wave = np.arange(144) % 4
per = np.stack([mixed[:, wave == w].sum(axis=1) for w in range(4)], axis=1)
print(task, 'mixed tiles per frame: mean %.1f' % mixed.sum(axis=1).mean())
print('  shipped assignment: slowest wavefront holds %.2f mixed tiles on average, the mean wavefront %.2f  (ratio %.3f)' % (per.max(axis=1).mean(), per.mean(), per.max(axis=1).mean() / per.mean()))
bal = np.ceil(mixed.sum(axis=1) / 4)
print('  balanced          : %.2f  (ratio %.3f)' % (bal.mean(), bal.mean() / per.mean()))
