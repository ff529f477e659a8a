set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -80 > gpurun_out/r02_gputests_c.log
tail -5 gpurun_out/r02_gputests_c.log
