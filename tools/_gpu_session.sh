cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "planar" 2>&1 | grep -v "^$" | head -60
