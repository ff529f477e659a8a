# A/B of the shipped library against magical_amd/libmagical_hip<suffix>.so (a -D build made with _native.build(defines=[...], out=...)):
#   bash tools/dev/run_ab_variant.sh _suffix     -- bench.py on four tasks, then the rasteriser GPU tests and the consistency sweep under the variant
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,3), round(d["ms_per_step"],4), round(r["avg_launch_ms"],4), (r.get("kernel_alone") or {}).get("avg_launch_ms"))'
for rep in 1 2; do for v in "" $1; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$P" mtc$v
done; done
for t in MatchRegions ClusterColour FixColour; do for v in "" $1; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary --steps 240 --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" ${t}$v
done; done
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_vectors.py -q -m gpu -x -k "render or stack or golden or preprocessors or ring or raster or obs" 2>&1 | tail -1
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$1.so timeout 600 python tools/raster_consistency_sweep.py 2>&1 | grep -v amdgpu.ids | awk '{print $NF, $(NF-1)}' | sort | uniq -c
