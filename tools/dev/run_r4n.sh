# round 4, after the phase-Q shrink: re-checks of earlier decisions that were taken at 63 spilled registers
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,3), round(d["ms_per_step"],4), round(r["avg_launch_ms"],4), (r.get("kernel_alone") or {}).get("avg_launch_ms"))'
for rep in 1 2; do for v in "" _g4 _sa; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$P" mtc$v
done; done
for t in MatchRegions ClusterColour FixColour MakeLine FindDupe; do for v in "" _g4 _sa; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary --steps 240 --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" ${t}$v
done
MGX_EXPERIMENT_CAP5=1 python bench.py --no-cpu-baseline --no-secondary --steps 240 --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" ${t}_cap5
done
