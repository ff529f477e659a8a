#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for t in MoveToCorner MoveToRegion ClusterColour FixColour; do echo "== clocks (depth 3) $t"; MGX_WONCE=1 MGX_WCAP=144 MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/raster_phase_clocks.py $t-Demo-v0 2>&1 | grep -v amdgpu; done
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
run() { MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary $3 2>/dev/null | python -c "$P" "$2"; }
for rep in 1 2; do
  run "" mtc_d2_wcap64; run _w3 mtc_d3_wcap64; run _w5x6 mtc_d2_w5x6; run _base mtc_base
  MGX_WCAP=144 run "" mtc_d2_wcap144; MGX_WCAP=48 run "" mtc_d2_wcap48; MGX_WONCE=0 run "" mtc_wonce0; MGX_WONCE=0 run _w5x6 mtc_wonce0_w5x6
done
for t in ClusterColour FixColour MoveToRegion; do
  run "" $t "--task $t-Demo-LoRes4E-v0"; MGX_WONCE=1 run "" ${t}_wonce1 "--task $t-Demo-LoRes4E-v0"; MGX_WONCE=1 run _w3 ${t}_wonce1_d3 "--task $t-Demo-LoRes4E-v0"; run _base ${t}_base "--task $t-Demo-LoRes4E-v0"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "render or stack or preprocessors or ring or raster or obs or fused" 2>&1 | tail -3
MGX_WCAP=8 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "render or stack or preprocessors or ring or raster or obs or fused" 2>&1 | tail -3
MGX_WCAP=8 timeout 600 python tools/raster_consistency_sweep.py 2>&1 | tail -10
} > gpurun_out/r6_wonce_ab3.log 2>&1
