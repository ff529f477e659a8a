"""GPU soak of the fused step: many steps with staggered episode clocks on every task / several batch sizes; reports consumer give-ups
(served by the clean-up launch) and timeouts of the hand-off, and compares the final state with a two-call engine (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
tasks = ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']
for task in tasks:
    for n, variant in ((4096, 'Demo'), (1000, 'TestJitter'), (37, 'Demo')):
        name = f'{task}-{variant}-LoRes4E-v0'
        a = magical_amd.make(name, n_envs=n, device='cuda:0'); b = magical_amd.make(name, n_envs=n, device='cuda:0', overlap=False)
        a.seed(4); b.seed(4); a.reset(); b.reset()
        ep = a.max_episode_steps
        clocks = np.zeros(n, dtype=np.int64); clocks[n // 2:] = ep // 3          # two groups: most steps take the fused path, two partial episode ends per episode
        a.set_episode_steps(clocks); b.set_episode_steps(clocks)
        tape = torch.as_tensor(np.random.RandomState(2).randint(0, 18, size=(T, n)).astype(np.int32), device='cuda:0')
        t0 = time.perf_counter()
        for s in range(T):
            oa, _, da, ia = a.step(tape[s])
        torch.cuda.synchronize(); ta = time.perf_counter() - t0
        for s in range(T):
            ob, _, db, ib = b.step(tape[s])
        torch.cuda.synchronize()
        same = torch.equal(oa, ob) and torch.equal(a.state_p, b.state_p) and torch.equal(a.state_f, b.state_f) and np.array_equal(ia['eval_score'], ib['eval_score'])
        d, to = a.handoff_stats()
        print('%-38s n=%4d  %5.2f M env-steps/s  fused steps gave up %d consumers of %d, timeouts %d, equal to the two-call engine: %s' % (
            name, n, n * T / ta / 1e6, d, n * T, to, same), flush=True)
        assert same and to == 0
        a.close(); b.close()
