for rep in 1 2; do for v in "" _nosort; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('mtc$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), d['roofline']['kernel_alone']['avg_launch_ms'])"
done; done
for v in "" _nosort; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 240 --warmup 20 --no-cpu-baseline --task ClusterColour-Demo-LoRes4E-v0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cc$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), d['roofline']['kernel_alone']['avg_launch_ms'])"
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 240 --warmup 20 --no-cpu-baseline --task MatchRegions-Demo-LoRes4E-v0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('mr$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), d['roofline']['kernel_alone']['avg_launch_ms'])"
done
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "render or stack or golden or preprocessors or ring" 2>&1 | tail -2
