cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -x -k "planar" 2>&1 | tail -2
for i in 1 2 3 4 5; do timeout 300 python bench.py --no-cpu-baseline --task MoveToCorner-Demo-LoResCHW4E-v0 --obs-ring 35 2>/dev/null > gpurun_out/ring_$i.json; python -c "
import json,sys; d=json.loads(open('gpurun_out/ring_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['episodes_finished'], d['roofline']['avg_launch_ms'])"; done
