# the kernels and copies of the episode-end step inside bench.py's 20-step window, with their gaps (rocprofv3 kernel + memory-copy trace)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/endk
rocprofv3 --kernel-trace --memory-copy-trace -f csv -d /tmp/endk -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > /dev/null 2>&1
python - <<'PY'
import csv, glob
ev = []
for r in csv.DictReader(open(glob.glob('/tmp/endk/**/*kernel_trace.csv', recursive=True)[0])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('mgx::', '')[:44]))
mc = glob.glob('/tmp/endk/**/*memory_copy_trace.csv', recursive=True)
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', ''))))
ev.sort()
# the LAST k_reset of the run that follows fused steps = the one inside the timed window
idx = [i for i, e in enumerate(ev) if e[2].startswith('k_reset')]
i = idx[-1]
lo = max(0, i - 8); t0 = ev[lo][0]
for s, e, n in ev[lo:i + 6]:
    print('%8.1f us  +%7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
