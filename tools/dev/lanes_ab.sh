#!/bin/bash
# lanes per env against the fused env-step: bash tools/dev/lanes_ab.sh "Task ..." "16 32"
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
for rep in 1 2; do for t in ${1:-ClusterColour ClusterShape FindDupe FixColour}; do for L in ${2:-16 32}; do
  python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary --task $t-Demo-LoRes4E-v0 --lanes $L 2>/dev/null | python -c "$P" "$t lanes=$L"
done; done; done
