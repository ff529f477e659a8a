# A/B of the two-lane contact split (MGX_CONTACT_SPLIT=0 is the control): phase cycles, state-only + fused throughput, tests
V=${1:-_nosplit}
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_probe.so python tools/step_phase_probe.py MoveToCorner-Demo-v0 2>&1 | grep -v amdgpu
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_probe.so python tools/step_phase_probe.py ClusterColour-Demo-v0 2>&1 | grep -v amdgpu | head -4
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,3), round(d["ms_per_step"],4), round(r["avg_launch_ms"],4), (r.get("kernel_alone") or {}).get("avg_launch_ms"))'
for rep in 1 2; do for v in "" $V; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$P" mtc$v
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary --task MoveToCorner-Demo-v0 2>/dev/null | python -c "$P" mtc_state$v
done; done
for t in MatchRegions ClusterColour FixColour; do for v in "" $V; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary --steps 240 --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" ${t}$v
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary --steps 240 --task $t-Demo-v0 2>/dev/null | python -c "$P" ${t}_state$v
done; done
for L in 16 32; do python bench.py --no-cpu-baseline --no-secondary --task MoveToCorner-Demo-v0 --lanes $L 2>/dev/null | python -c "$P" mtc_state_lanes$L; python bench.py --no-cpu-baseline --no-secondary --lanes $L 2>/dev/null | python -c "$P" mtc_lanes$L; done
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
