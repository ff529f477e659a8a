"""GPU stress: every task in the variant given (default Demo; e.g. TestAll, TestCountPlus), 4096 envs, two episodes of random
actions; reports contact-capacity overflows, NaNs and bodies outside the arena (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
TASKS = ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']
N = 4096
VARIANT = sys.argv[1] if len(sys.argv) > 1 else 'Demo'
magical_amd.register_envs()
for task in TASKS:
    if f'{task}-{VARIANT}-v0' not in magical_amd.ALL_REGISTERED_ENVS:
        continue
    env = magical_amd.make(f'{task}-{VARIANT}-v0', n_envs=N, device='cuda:0')
    env.seed(11)
    env.reset()
    T = 2 * env.max_episode_steps
    tape = torch.as_tensor(np.random.RandomState(1).randint(0, 18, size=(T, N)).astype(np.int32), device='cuda:0')
    overflow = 0; worst = 0.0; nan = 0; max_cache = 0
    for s in range(T):
        env.step(tape[s])
        if s % 10 == 9 or s == T - 1:
            overflow = max(overflow, int(env.state_i[2].max().item()))
            max_cache = max(max_cache, int(env.state_i[1].max().item()))
            p = env.get_poses_tensor()
            nan += int(torch.isnan(p).sum().item())
            present = torch.as_tensor(env.entity_enabled[:, [e.ent_id for e in env._entities if e.body is not None]].any(axis=1))   # (absent blocks sit wherever they were)
            worst = max(worst, float(p[:, 1:, :2].abs().max().item()) if VARIANT in ('Demo',) or not env.variable_worlds else float((p[:, 1:7, :2]).abs().max().item()))
    print('%-14s steps %4d  overflow flag max %d  live cache slots max %d / %d  NaNs %d  max |x|,|y| of any body %.3f' % (
        task, T, overflow, max_cache, env._info('cache_slots') if not env.variable_worlds else -1, nan, worst))
    env.close()
