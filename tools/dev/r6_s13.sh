#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_s13; mkdir -p $O
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
run() { MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so python bench.py --steps $4 --warmup 20 --no-cpu-baseline --no-secondary $3 2>/dev/null | python -c "$P" "$2"; }
{
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_d2.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "render or stack or preprocessors or ring or raster or obs or fused" 2>&1 | tail -3
for rep in 1 2 3; do for v in _d1 _d2; do run "$v" "mtc$v" "" 400; done; done
for rep in 1 2; do for t in ClusterColour MatchRegions FixColour; do for v in _d1 _d2; do run "$v" "$t$v" "--task $t-Demo-LoRes4E-v0" 400; done; done; done
for rep in 1 2; do for t in ClusterColour-TestAll MatchRegions-TestShape FixColour-TestAll; do for v in _d1 _d2; do run "$v" "$t$v" "--task $t-LoRes4E-v0" 200; done; done; done
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_d2.so timeout 600 python tools/raster_consistency_sweep.py 2>&1 | tail -10
} > $O/ab.txt 2>&1
