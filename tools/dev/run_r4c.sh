# round 4: whole GPU suite on the rebuilt library (k_score_points in LDS columns, host fixes, tightened fp64 gate, RCCL, capture consumers)
set -x
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "contact_coverage" 2>&1 | grep -v amdgpu.ids > gpurun_out/r4c_f64_gate.txt; tail -3 gpurun_out/r4c_f64_gate.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4c_bench20.json 2> gpurun_out/r4c_bench20.err; wc -l gpurun_out/r4c_bench20.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4c_bench20.json').read())
print(d['value'], d['ms_per_step'], d.get('collective'))
print({k: v for k, v in d['secondary'].items() if not isinstance(v, dict)})
PY
