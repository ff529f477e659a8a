cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "fused or planar or fleet" 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench400_$i.json 2>gpurun_out/err.txt; python - <<PY
import json; d=json.loads(open('gpurun_out/bench400_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['episodes_finished'], d['roofline']['avg_launch_ms'], d['roofline']['other_kernels']['k_step']['avg_launch_ms'])
PY
done
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench20_$i.json 2>gpurun_out/err.txt; python - <<PY
import json; d=json.loads(open('gpurun_out/bench20_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['episodes_finished'], d['roofline']['avg_launch_ms'], d['roofline']['other_kernels']['k_step']['avg_launch_ms'])
PY
done
