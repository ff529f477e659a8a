#!/bin/bash
# quick A/B on one box: bash tools/dev/ab_quick.sh "" _base   (suffixes of magical_amd/libmagical_hip<suffix>.so); first suffix also runs the rasteriser's tests
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
for rep in 1 2; do for v in "$@"; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$P" "mtc$v"
done; done
for t in ClusterColour MatchRegions FixColour; do for v in "$@"; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-secondary --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" "$t$v"
done; done
if [ -z "$NO_TESTS" ]; then
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_reference_vectors.py -q -m gpu -x -k "render or stack or golden or preprocessors or ring or raster or obs or fused" 2>&1 | tail -3
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so timeout 600 python tools/raster_consistency_sweep.py 2>&1 | tail -3
fi
