"""GPU sweep: the rasteriser's fast path (tile / pixel classification in fp32 with error bounds, fp64 only where needed)
against its own exact per-sample painter (k_raster_native at 384x384, then the 4x4 box filter) on random rollout states:
every env, every few steps, byte for byte.  Complements the oracle comparisons with many more states (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
from magical_amd.saved_trajectories import area_resize_4x
names = sys.argv[1:] or [f'{t}-{v}-LoRes4E-v0' for t, v in (
    ('MoveToCorner', 'Demo'), ('MoveToRegion', 'TestAll'), ('MatchRegions', 'Demo'), ('MakeLine', 'TestAll'), ('FindDupe', 'TestCountPlus'),
    ('FixColour', 'TestAll'), ('ClusterColour', 'Demo'), ('ClusterShape', 'TestAll'), ('MatchRegions', 'TestCountPlus'))]
N, T = 32, 30
bad = 0
for name in names:
    env = magical_amd.make(name, n_envs=N, device='cuda:0', max_episode_steps=12)
    env.seed(int(os.environ.get('MGX_TEST_SEED', '3')))
    obs = env.reset()
    tape = np.random.RandomState(17).randint(0, 18, size=(T, N)).astype(np.int32)
    checked = mism = 0
    for s in range(T):
        obs, _, _, _ = env.step(tape[s])
        if s % 3 == 2:
            o = obs.cpu().numpy()
            for k in range(N):
                small = area_resize_4x(env.render(env=k)['ego'])
                checked += 1
                if not np.array_equal(small, o[k, :, :, 9:12]):
                    mism += 1
                    d = np.argwhere(small != o[k, :, :, 9:12])
                    print('  MISMATCH', name, 'step', s, 'env', k, len(d), 'entries, first', d[0].tolist(), int(small[tuple(d[0])]), int(o[k, :, :, 9:12][tuple(d[0])]))
    print(f'{name}: {checked} frames checked, {mism} mismatches')
    bad += mism
    env.close()
sys.exit(1 if bad else 0)
