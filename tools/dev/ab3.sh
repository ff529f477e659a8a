#!/bin/bash
# bash tools/dev/ab3.sh "" _narrow ... : fused + alone over six tasks for several library builds
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
for t in ${TASKS:-MoveToCorner MoveToRegion MatchRegions FixColour FindDupe ClusterColour}; do for v in "$@"; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-secondary --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" "$t$v"
done; done
