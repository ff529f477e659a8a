cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "fused or other_preproc or autoreset or rollouts" 2>&1 | tail -3
python - <<'PY'
import time, numpy as np, torch, magical_amd
for name in ['MoveToCorner-Demo-LoRes3EA-v0', 'MoveToCorner-Demo-LoResStack-v0', 'ClusterColour-Demo-LoRes3EA-v0']:
    for ov in (False, True):
        e = magical_amd.make(name, n_envs=4096, device='cuda:0', overlap=ov, max_episode_steps=100000); e.reset()
        tape = torch.as_tensor(np.random.RandomState(2).randint(0, 18, size=(240, 4096)).astype(np.int32), device='cuda:0')
        for s in range(40): e.step(tape[s])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(40, 240): e.step(tape[s])
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 200
        print('%-34s overlap=%s  %.3f ms/step  %.2f M' % (name, ov, t * 1e3, 4096 / t / 1e6)); e.close()
PY
