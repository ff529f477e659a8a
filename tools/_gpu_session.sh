cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
MGX_DEBUG_LAUNCH=1 timeout 300 python tools/task_step_times.py 2>&1 | grep "ms/step\|HBM\|compact"
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
timeout 900 python tools/raster_consistency_sweep.py 2>&1 | grep -v amdgpu | grep -c "0 mismatches"
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M  %.4f ms/step; k_raster %.3f k_step %.3f alone %s' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['other_kernels']['k_step']['avg_launch_ms'], r['kernel_alone']['avg_launch_ms']))"; done
