cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "batched_draws" 2>&1 | tail -25
timeout 300 python tools/reset_profile.py ClusterColour-TestAll-LoRes4E-v0 2>&1 | head -12
timeout 300 python tools/reset_profile.py MatchRegions-TestCountPlus-LoRes4E-v0 2>&1 | head -4
