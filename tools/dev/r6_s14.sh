#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_s14; mkdir -p $O
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
run() { MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary $3 2>/dev/null | python -c "$P" "$2"; }
{
for rep in 1 2 3 4; do for v in "" _d1; do run "$v" "mtc$v"; done; done
for rep in 1 2; do for t in ClusterColour MatchRegions FixColour MoveToRegion; do for v in "" _d1; do run "$v" "$t$v" "--task $t-Demo-LoRes4E-v0"; done; done; done
} > $O/ab.txt 2>&1
