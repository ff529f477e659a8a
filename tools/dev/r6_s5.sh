#!/bin/bash
# round 6 session 5: ph_arbiters_joints' prefix sums by ballot (A/B against MGX_ARB_BALLOT=0), per-env-world variants against round 5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_s5; mkdir -p $O
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
run() { MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary $3 2>/dev/null | python -c "$P" "$2"; }
{
for t in MoveToCorner ClusterColour ClusterShape FindDupe MatchRegions; do for v in "" _arb0 _base; do
  run "$v" "$t$v fused" "--task $t-Demo-LoRes4E-v0"; run "$v" "$t$v state-only" "--task $t-Demo-v0"
done; done
for t in MatchRegions-TestAll ClusterColour-TestAll ClusterShape-TestAll ClusterColour-TestCountPlus; do for v in "" _base; do
  run "$v" "$t$v fused" "--task $t-LoRes4E-v0"; run "$v" "$t$v state-only" "--task $t-v0"
done; done
} > $O/ab.txt 2>&1
for t in MoveToCorner ClusterColour; do for v in _probe; do
  echo "== $t $v"; MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so timeout 300 python tools/step_phase_probe.py $t-Demo-v0 2>&1 | grep -v amdgpu
done; done > $O/phase_cycles.txt 2>&1
python tools/lanes_invariance.py > $O/lanes_invariance.txt 2>&1
timeout 3000 python -m pytest tests -q -m gpu -x > $O/full_suite.txt 2>&1
tail -n 3 $O/full_suite.txt
