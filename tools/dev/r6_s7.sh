#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_s7; mkdir -p $O
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4), "fused launch k_raster", round(r.get("avg_launch_ms") or 0,4))'
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary $2 2>/dev/null | python -c "$P" "$1"; }
{
for t in MatchRegions-TestAll ClusterColour-TestAll ClusterShape-TestAll ClusterColour-TestCountPlus FindDupe-TestAll MoveToCorner-TestAll; do
  run "$t fused" "--task $t-LoRes4E-v0"
  MGX_NO_OVERLAP=1 run "$t serial" "--task $t-LoRes4E-v0"
  run "$t fused" "--task $t-LoRes4E-v0"
  MGX_NO_OVERLAP=1 run "$t serial" "--task $t-LoRes4E-v0"
done
} > $O/ab.txt 2>&1
