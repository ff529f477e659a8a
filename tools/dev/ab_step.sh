#!/bin/bash
# A/B of k_step changes on one box: bash tools/dev/ab_step.sh "" _base -- state-only + fused lines, then the physics parity tests on the first
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; ok=(r.get("other_kernels") or {}).get("k_step",{}); print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; k_step", round(ka.get("k_step", r.get("avg_launch_ms") if r.get("kernel")=="k_step" else ok.get("avg_launch_ms",0)),4))'
for t in MoveToCorner ClusterColour MatchRegions FindDupe; do for v in "$@"; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-secondary --task $t-Demo-v0 2>/dev/null | python -c "$P" "state-only $t$v"
done; done
for t in MoveToCorner ClusterColour MatchRegions; do for v in "$@"; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-secondary --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" "fused $t$v"
done; done
if [ -z "$NO_TESTS" ]; then
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "tracks_oracle or one_step or determinism or lanes_per_env or rollouts or per_env_worlds_match or longest_first or fused or variants_match_golden or capacity" 2>&1 | tail -4
fi
