"""GPU check: the physics result does not depend on the lanes-per-env launch geometry (bitwise), so the engine may pick it
per launch from the worlds currently loaded."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magical_amd
for name in ['ClusterColour-Demo-v0', 'MatchRegions-Demo-v0', 'MoveToCorner-Demo-v0']:
    outs = {}
    for L in (16, 32, 64):
        try:
            env = magical_amd.make(name, n_envs=64, device='cuda:0', lanes_per_env=L)
        except Exception as ex:
            print(name, L, 'n/a', str(ex)[:60]); continue
        env.seed(5); env.reset()
        tape = np.random.RandomState(1).randint(0, 18, size=(60, 64)).astype(np.int32)
        for s in range(60): env.step(tape[s])
        outs[L] = (env.state_p.clone(), env.state_f[:env._motion_rows.max() + 1].clone())
        env.close()
    ks = sorted(outs)
    print(name, {L: bool(torch.equal(outs[L][0], outs[ks[0]][0]) and torch.equal(outs[L][1], outs[ks[0]][1])) for L in ks})
