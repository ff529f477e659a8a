#!/bin/bash
# round 6 session 2: ph_narrow's slots by ballot, the sort-and-sweep broadphase A/B, forced hand-off failures, the fp64 drift table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_s2; mkdir -p $O
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
run() { MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary $3 2>/dev/null | python -c "$P" "$2"; }
{
for t in MoveToCorner ClusterColour ClusterShape FindDupe MatchRegions; do for v in "" _base _sap _slotsall; do
  run "$v" "$t$v fused" "--task $t-Demo-LoRes4E-v0"; run "$v" "$t$v state-only" "--task $t-Demo-v0"
done; done
} > $O/ab.txt 2>&1
for t in MoveToCorner ClusterColour ClusterShape FindDupe; do for v in _probe _sapprobe; do
  echo "== $t $v"; MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so timeout 300 python tools/step_phase_probe.py $t-Demo-v0 2>&1 | grep -v amdgpu
done; done > $O/phase_cycles.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "forced or f64_drift or long_run" > $O/new_tests.txt 2>&1
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_sap.so timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "not variants_match and not batched_draws" > $O/sap_suite.txt 2>&1
timeout 1200 python tools/drift_table.py --envs 32 --norm l2 > $O/pose_drift_f64_vs_oracle.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu -x > $O/full_suite.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver_line.err
tail -3 $O/new_tests.txt $O/sap_suite.txt $O/full_suite.txt
