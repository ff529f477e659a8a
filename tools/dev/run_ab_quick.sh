# quick A/B of library builds: MoveToCorner bench, 400 steps, twice each.   usage: bash tools/dev/run_ab_quick.sh _suffix1 _suffix2 ...
for rep in 1 2; do for v in "$@"; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('mtc$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), d['roofline']['kernel_alone']['avg_launch_ms'])"
done; done
