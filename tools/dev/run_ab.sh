# A/B of two builds of the library: bench.py (MoveToCorner 400 steps x2, ClusterColour, MatchRegions) + the rasteriser's GPU tests on the first
# usage: bash tools/dev/run_ab.sh "" _base      (suffixes of magical_amd/libmagical_hip<suffix>.so)
for rep in 1 2; do for v in "$@"; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('mtc$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), d['roofline']['kernel_alone']['avg_launch_ms'])"
done; done
for v in "$@"; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('mtc20$v', round(d['value']/1e6,3), round(d['ms_per_step'],4))"
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-secondary --task ClusterColour-Demo-LoRes4E-v0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cc$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), d['roofline']['kernel_alone']['avg_launch_ms'])"
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-secondary --task MatchRegions-Demo-LoRes4E-v0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('mr$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), d['roofline']['kernel_alone']['avg_launch_ms'])"
done
python -m pytest tests/test_gpu_parity.py tests/test_reference_vectors.py -q -m gpu -k "render or stack or golden or preprocessors or ring or raster or obs" 2>&1 | tail -3
python tools/raster_consistency_sweep.py 2>&1 | tail -3
