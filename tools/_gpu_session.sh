cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 600 python bench.py > $O/r02_bench_mtc_lores4e.json 2> $O/err1.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_bench_mtc_lores4e_20steps.json 2> $O/err2.txt
MGX_NO_OVERLAP=1 timeout 300 python bench.py --no-cpu-baseline > $O/r02_bench_mtc_lores4e_serial.json 2> $O/err3.txt
timeout 300 python bench.py --no-cpu-baseline --task ClusterColour-Demo-LoRes4E-v0 > $O/r02_bench_clustercolour_lores4e.json 2> $O/err4.txt
MGX_NO_OVERLAP=1 timeout 300 python bench.py --no-cpu-baseline --task ClusterColour-Demo-LoRes4E-v0 > $O/r02_bench_clustercolour_lores4e_serial.json 2> $O/err4.txt
timeout 300 python bench.py --no-cpu-baseline --task MoveToCorner-Demo-v0 > $O/r02_bench_mtc_state_only.json 2> $O/err5.txt
timeout 300 python bench.py --no-cpu-baseline --dtype f64 > $O/r02_bench_mtc_lores4e_f64.json 2> $O/err6.txt
timeout 300 python bench.py --no-cpu-baseline --dtype f64 --task ClusterColour-Demo-LoRes4E-v0 > $O/r02_bench_clustercolour_lores4e_f64.json 2> $O/err6.txt
timeout 300 python bench.py --no-cpu-baseline --task MoveToCorner-Demo-LoResCHW4E-v0 --obs-ring 35 > $O/r02_bench_mtc_loreschw4e_ring35.json 2> $O/err7.txt
timeout 300 python bench.py --no-cpu-baseline --task MoveToCorner-Demo-LoResCHW4E-v0 > $O/r02_bench_mtc_loreschw4e_inplace_stack.json 2> $O/err7.txt
timeout 600 python bench.py --config5 --envs5 1024 > $O/r02_bench_config5_1gpu_8x1024.json 2> $O/err8.txt
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/p1 -o mtc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/r02_bench_mtc_lores4e_under_rocprof.json 2> /dev/null
cp /tmp/p1/mtc_kernel_stats.csv $O/r02_bench_mtc_lores4e_kernel_stats.csv
MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p3 -o mtcs -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1
cp /tmp/p3/mtcs_kernel_stats.csv $O/r02_bench_mtc_lores4e_serial_kernel_stats.csv
rocprofv3 --kernel-trace --stats -f csv -d /tmp/p2 -o cc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --task ClusterColour-Demo-LoRes4E-v0 > /dev/null 2>&1
cp /tmp/p2/cc_kernel_stats.csv $O/r02_bench_cc_lores4e_kernel_stats.csv
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/rollout_all_tasks.py --variant all --envs 4096 > $O/r02_rollout_all_60_variants_4096x1gpu.jsonl 2> $O/err9.txt
cd $O; for f in *.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],4), r.get('avg_launch_ms'), (r.get('other_kernels') or {}).get('k_step',{}).get('avg_launch_ms'), r.get('frac'), (r.get('kernel_alone') or {}).get('avg_launch_ms'), (r.get('kernel_alone') or {}).get('frac'))
except Exception as ex: print('$f', 'ERR', ex)
PY
done
python - <<'PY'
import csv,glob
for f in sorted(glob.glob('*kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        if 'mgx::' in r['Name']: print(f[:40], r['Name'][10:36], r['Calls'], r['AverageNs'])
PY
wc -l r02_rollout_all_60_variants_4096x1gpu.jsonl
