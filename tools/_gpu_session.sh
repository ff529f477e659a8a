cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
timeout 900 python tools/stress_all_tasks.py TestAll 2>&1 | grep -v amdgpu | tail -9
