cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python tools/rollout_all_tasks.py --variant all --envs 4096 > gpurun_out/r02_rollout_all_60_variants_4096x1gpu.jsonl 2> gpurun_out/rollout.err
wc -l gpurun_out/r02_rollout_all_60_variants_4096x1gpu.jsonl; tail -2 gpurun_out/rollout.err
