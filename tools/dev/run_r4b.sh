# round 4: RCCL on one GPU (test + smoke), and the default bench line with its one-rank collective and four secondary lines
set -x
timeout 600 python -m pytest tests/test_distributed.py -q -m gpu -x -s 2>&1 | tail -12
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4b_bench20.json 2> gpurun_out/r4b_bench20.err; tail -5 gpurun_out/r4b_bench20.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4b_bench20.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('collective'))
for k, v in d['secondary'].items():
    print(k, v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != 'roofline'})
print(d['roofline']['scratch_bytes'], d['roofline']['other_kernels']['k_step']['scratch_bytes'])
PY
