import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, time
import magical_amd
N = 4096
env = magical_amd.make('MoveToCorner-Demo-LoRes4E-v0', n_envs=N, device='cuda:0')
env.reset()
tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(60, N)).astype(np.int32), device='cuda:0')
a = torch.randn(8192, 8192, device='cuda:0'); 
for s in range(40):
    obs, *_ = env.step(tape[s])
    b = a @ a            # ~10 ms of "policy" on the caller's stream, reading nothing of ours
    c = obs.float().mean()   # and a reader of the observation
torch.cuda.synchronize()
print('with user work between steps: (deferred, timeouts) =', env.handoff_stats())
