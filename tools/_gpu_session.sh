cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r02_gputests_e.log
tail -4 gpurun_out/r02_gputests_e.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_fused_20.json 2> gpurun_out/r02_bench.err
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_fused_400.json 2>> gpurun_out/r02_bench.err
timeout 300 python bench.py --no-cpu-baseline --task ClusterColour-Demo-LoRes4E-v0 > gpurun_out/r02_bench_fused_cc.json 2>> gpurun_out/r02_bench.err
timeout 600 python tools/rollout_all_tasks.py --envs 1024 --concurrent > gpurun_out/r02_config5_1gpu_1024.json 2>> gpurun_out/r02_bench.err
cat gpurun_out/r02_config5_1gpu_1024.json
