import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
magical_amd.register_envs()
N = 4096
for name in sys.argv[1:]:
    env = magical_amd.make(name, n_envs=N, device='cuda:0')
    env.seed(3)
    t0 = time.perf_counter(); env.reset(); torch.cuda.synchronize(); t_reset = time.perf_counter() - t0
    T = env.max_episode_steps + 20
    tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(T, N)).astype(np.int32), device='cuda:0')
    for s in range(5): env.step(tape[s])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr = 0.0
    for s in range(5, T):
        obs, rew, done, info = env.step(tape[s])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ovf = int((env.state_i[2] > 0).sum())
    print(f'{name}: first reset {t_reset*1e3:.0f} ms, {N*(T-5)/dt/1e6:.2f}M env-steps/s over {T-5} steps (one episode end inside), L={env.lanes_per_env}, '
          f'lds step {env._lib.mgx_engine_lds_bytes(env._engine,0)} raster {env._lib.mgx_engine_lds_bytes(env._engine,1)}, overflow envs {ovf}, mean score {info["eval_score"].mean():.3f}')
    env.close()
