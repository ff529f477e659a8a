# gaps between the kernels of consecutive fused env-steps, from rocprofv3's kernel trace (start / end timestamps)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/gaps
rocprofv3 --kernel-trace -f csv -d /tmp/gaps -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 200 > /dev/null 2>&1
python - <<'PY'
import csv, glob, statistics as st
f = glob.glob('/tmp/gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'mgx::' in r['Kernel_Name']]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('mgx::')[1][:18]) for r in rows))
steps = []   # (k_step, k_raster, k_raster_deferred) triples; k_env_order (behind k_step, on its stream) is listed apart
order = [e for e in ev if e[2].startswith('k_env_order')]
ev = [e for e in ev if not e[2].startswith('k_env_order')]
i = 0
while i + 3 < len(ev):
    names = [ev[i + k][2] for k in range(3)]
    if names[0].startswith('k_step<') and names[1].startswith('k_raster<') and names[2].startswith('k_raster_deferred'):
        steps.append(ev[i:i + 3]); i += 3
    else: i += 1
pairs = [(a, b) for a, b in zip(steps, steps[1:]) if b[0][0] - a[2][1] < 200000]
q = lambda xs: 'p50 %.1f  p90 %.1f us' % (st.median(xs) / 1e3, sorted(xs)[int(len(xs) * 0.9)] / 1e3)
print(len(pairs), 'consecutive fused steps')
print('k_step start -> k_raster start      ', q([a[1][0] - a[0][0] for a, _ in pairs]))
print('k_step duration                     ', q([a[0][1] - a[0][0] for a, _ in pairs]))
print('k_raster duration                   ', q([a[1][1] - a[1][0] for a, _ in pairs]))
print('k_raster end -> clean-up start      ', q([a[2][0] - a[1][1] for a, _ in pairs]))
print('clean-up duration                   ', q([a[2][1] - a[2][0] for a, _ in pairs]))
print('clean-up end -> next k_step start   ', q([b[0][0] - a[2][1] for a, b in pairs]))
if order:
    ks = {a[0][0]: a for a, _ in pairs}
    import bisect
    starts = sorted(ks)
    d = []
    for o in order:
        j = bisect.bisect_right(starts, o[0]) - 1
        if j >= 0: d.append((o[0] - ks[starts[j]][0][1], o[1] - o[0], ks[starts[j]][1][1] - o[1]))
    print('k_step end -> k_env_order start     ', q([x[0] for x in d])); print('k_env_order duration               ', q([x[1] for x in d])); print('k_env_order end -> k_raster end     ', q([x[2] for x in d]))
print('k_step start -> next k_step start   ', q([b[0][0] - a[0][0] for a, b in pairs]))
PY
