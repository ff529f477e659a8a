# instruction-cache counters of k_raster for two builds of the library (usage: bash tools/dev/icache_ab.sh _base _v_plain)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' '; echo
for v in "$@"; do
  rm -rf /tmp/ic$v
  MGX_LIB_PATH=$GRAFT_REPO_ROOT/magical_amd/libmagical_hip$v.so MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM -f csv -d /tmp/ic$v -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 60 > /tmp/ic$v.log 2>&1
  python - <<PY
import csv, glob, collections, statistics
fs = glob.glob('/tmp/ic$v/**/*counter_collection.csv', recursive=True)
if not fs: print('$v: no counter file'); print(open('/tmp/ic$v.log').read()[-1500:]); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name']
    if 'k_raster<' in k and 'deferred' not in k: acc['k_raster'][r['Counter_Name']].append(float(r['Counter_Value']))
    elif 'k_step<' in k: acc['k_step'][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print('$v', k, {n: round(statistics.median(x)) for n, x in sorted(c.items())})
PY
done
