# per-phase instruction counts of k_raster: the kernel truncated after phase S / C / T / Q (MGX_RASTER_PROBE build) under rocprofv3 --pmc
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/rphase; mkdir -p $O
MGX_LIB_PATH=$GRAFT_REPO_ROOT/magical_amd/libmagical_hip_rprobe.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -f csv -d /tmp/rph -o run -- python $GRAFT_REPO_ROOT/tools/raster_phase_probe.py MoveToCorner-Demo-v0 > $O/probe.log 2>&1
python - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/rph/**/*_counter_collection.csv', recursive=True)[0])))
# dispatches in order; keep k_raster<double,1,5> (stack4) launches
byd = collections.OrderedDict()
for r in rows:
    if 'k_raster<' not in r['Kernel_Name'] or 'deferred' in r['Kernel_Name']: continue
    byd.setdefault((int(r['Dispatch_Id']), r['Kernel_Name'][:40]), {})[r['Counter_Name']] = float(r['Counter_Value'])
items = list(byd.items())
print(len(items), 'raster dispatches')
for (d, k), c in items[-60:]:
    print(d, k, {n: round(v / 16384) for n, v in c.items() if n.startswith('SQ_INSTS')}, 'wave_cyc/wave', round(c.get('SQ_WAVE_CYCLES', 0) / 16384), 'wait', round(c.get('SQ_WAIT_ANY', 0) / 16384), 'lanes', round(c.get('SQ_THREAD_CYCLES_VALU', 0) / max(c.get('SQ_INSTS_VALU', 1), 1) / 64, 2))
PY
