#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in _clocks _w2clocks; do echo "== $v"; MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python tools/dev/raster_phase_clocks.py MoveToCorner-Demo-v0 2>&1 | grep -v amdgpu; done
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
for rep in 1 2; do for v in "" _w2 _base; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$P" "mtc$v"
done; done
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_w2.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "render or stack or preprocessors or ring or raster or obs or fused" 2>&1 | tail -3
} > gpurun_out/r6_wonce_ab2.log 2>&1
