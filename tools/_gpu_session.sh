cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
MGX_DEBUG_LAUNCH=1 timeout 300 python tools/task_step_times.py 2>&1 | grep "ms/step\|compact"
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
timeout 900 python tools/raster_consistency_sweep.py 2>&1 | grep -v amdgpu | grep -c "0 mismatches"
