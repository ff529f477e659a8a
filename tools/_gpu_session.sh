cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 900 python tools/fused_soak.py 200 2>&1 | grep -v amdgpu | awk '{print $1, $2, $3, $4, $9, $10, $11, $12, $13, $14, $15, $16}' | grep -v " 0 consumers" | head
