# one --pmc pass over bench.py's kernels (launched one after the other), medians per launch.
# usage: bash tools/dev/pmc_quick.sh "COUNTER ..." [VAR=value ...]      e.g. "SQ_INSTS_VALU SQ_WAVE_CYCLES" MGX_RASTER_PATCHES=1
cd /tmp; export TMPDIR=/tmp
C="$1"; shift
for kv in "$@"; do export "$kv"; done
tag=$(echo "$*" | tr -c 'A-Za-z0-9\n' '_'); rm -rf /tmp/pq_$tag
MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $C -f csv -d /tmp/pq_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 60 $BENCH_ARGS > /tmp/pq_$tag.log 2>&1
python - <<PY
import csv, glob, collections, statistics
fs = glob.glob('/tmp/pq_$tag/**/*counter_collection.csv', recursive=True)
if not fs: print('no counter file'); print(open('/tmp/pq_$tag.log').read()[-1500:]); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name']
    if 'k_raster<' in k and 'deferred' not in k: acc['k_raster'][r['Counter_Name']].append(float(r['Counter_Value']))
    elif 'k_step<' in k: acc['k_step'][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print('[$*]', k, {n: round(statistics.median(x)) for n, x in sorted(c.items())})
PY
