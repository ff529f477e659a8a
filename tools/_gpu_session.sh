cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== default"; timeout 300 python tools/task_step_times.py 2>&1 | grep -v amdgpu
for q in 768 576; do echo "== QCAP=$q"; MGX_LIB_PATH=$PWD/build/libmgx_q$q.so timeout 300 python tools/task_step_times.py 2>&1 | grep -v amdgpu; done
