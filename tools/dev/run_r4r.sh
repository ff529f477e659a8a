P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,3), round(d["ms_per_step"],4), round(r["avg_launch_ms"],4), (r.get("kernel_alone") or {}).get("avg_launch_ms"))'
for rep in 1 2; do for v in _head ""; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary --task MoveToCorner-Demo-v0 2>/dev/null | python -c "$P" mtc_state$v
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary --steps 240 --task ClusterColour-Demo-v0 2>/dev/null | python -c "$P" cc_state$v
done; done
