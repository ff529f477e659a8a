#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_s15; mkdir -p $O
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
run() { MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary $3 2>/dev/null | python -c "$P" "$2"; }
{
for rep in 1 2 3; do for v in "" _st2 _st5; do run "$v" "mtc$v"; done; done
for t in ClusterColour MatchRegions; do for v in "" _st2 _st5; do run "$v" "$t$v" "--task $t-Demo-LoRes4E-v0"; done; done
} > $O/ab.txt 2>&1
