#!/bin/bash
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/psp -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary > /tmp/psp.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/psp/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:60], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'total ms', round(float(r['TotalDurationNs'])/1e6,2))
t = glob.glob('/tmp/psp/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(t))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one fused step in the middle: print start/end relative
names = [(r['Kernel_Name'][:24], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
mid = len(names) // 2
t0 = names[mid][1]
for n, s, e in names[mid:mid + 14]:
    print('%-26s start %8.1f us  end %8.1f us' % (n, (s - t0) / 1e3, (e - t0) / 1e3))
PY
