#!/bin/bash
# round 6 session 3: list-form broadphase unrolled 2x / 4x, forced hand-off failures, the rest of the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_s3; mkdir -p $O
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
run() { MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary $3 2>/dev/null | python -c "$P" "$2"; }
{
for t in MoveToCorner ClusterColour ClusterShape FindDupe MatchRegions; do for v in "" _u2 _u4 _base; do
  run "$v" "$t$v fused" "--task $t-Demo-LoRes4E-v0"; run "$v" "$t$v state-only" "--task $t-Demo-v0"
done; done
} > $O/ab.txt 2>&1
for t in MoveToCorner ClusterColour; do for v in _u2probe _u4probe; do
  echo "== $t $v"; MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so timeout 300 python tools/step_phase_probe.py $t-Demo-v0 2>&1 | grep -v amdgpu | grep "launch\|ph_broad \|ph_narrow \|per-workgroup"
done; done > $O/phase_cycles.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "forced or long_run" > $O/new_tests.txt 2>&1
timeout 3000 python -m pytest tests -q -m gpu -x > $O/full_suite.txt 2>&1
tail -n 3 $O/new_tests.txt $O/full_suite.txt
