cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -k "rollouts or golden or batched_draws" 2>&1 | tail -4
for n in 1 2 3 5; do MGX_N=$n python - <<'PY'
import os, numpy as np, torch, magical_amd
n=int(os.environ['MGX_N'])
magical_amd.register_envs()
bad=0
for name in magical_amd.ALL_REGISTERED_ENVS:
    if name.count('-')!=2 or '-Demo-' in name: continue
    for seed in (1,2,3):
        a=magical_amd.make(name,n_envs=n,device='cuda:0',max_episode_steps=1); b=magical_amd.make(name,n_envs=n,device='cuda:0',max_episode_steps=1,batch_draws=False)
        a.seed(seed); b.seed(seed); a.reset(); b.reset()
        for _ in range(3):
            a.step(np.zeros(n,dtype=np.int32)); b.step(np.zeros(n,dtype=np.int32))
            if not (torch.equal(a.state_p,b.state_p) and np.array_equal(a.entity_shape_types,b.entity_shape_types) and np.array_equal(a.entity_colours,b.entity_colours)): bad+=1; print('MISMATCH',name,n,seed)
        a.close(); b.close()
print('n',n,'mismatches',bad)
PY
done
