#!/bin/bash
# k_step A/B: bash tools/dev/ab_step.sh "" _prev   (state-only lines + fused lines)
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
for t in ${TASKS:-MoveToCorner-Demo-v0 ClusterColour-Demo-v0 FindDupe-Demo-v0 MoveToCorner-Demo-LoRes4E-v0 ClusterColour-Demo-LoRes4E-v0 FindDupe-Demo-LoRes4E-v0}; do for v in "$@"; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary --task $t 2>/dev/null | python -c "$P" "$t$v"
done; done
