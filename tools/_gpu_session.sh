cd /tmp
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace -f csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --steps 200 --warmup 10 > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,statistics
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[]
for r in rows:
    n=r['Kernel_Name']
    k='deferred' if 'k_raster_deferred' in n else 'raster' if 'k_raster' in n else 'step' if 'k_step' in n else None
    if k: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),k))
ev.sort()
# per fused step: step start, raster start, step end, raster end, deferred start/end, next step start
seq=[]
i=0
steps=[e for e in ev if e[2]=='step']; rast=[e for e in ev if e[2]=='raster']; dfr=[e for e in ev if e[2]=='deferred']
print(len(steps),len(rast),len(dfr))
import bisect
gaps=[];ov=[];r_after=[];d_gap=[];period=[]
for a,b in zip(steps[50:-2],steps[51:-1]):
    period.append((b[0]-a[0])/1e3)
# match raster launches to steps by time
rs=[r[0] for r in rast]
for s in steps[50:-2]:
    j=bisect.bisect_left(rs,s[0]-20000)
    if j<len(rast) and abs(rast[j][0]-s[0])<200000:
        r=rast[j]
        ov.append((r[0]-s[0])/1e3); r_after.append((r[1]-s[1])/1e3)
        # next step start after raster end
        k=bisect.bisect_right([x[0] for x in steps], r[1])
        if k<len(steps): gaps.append((steps[k][0]-r[1])/1e3)
print('step period us: median %.1f'%statistics.median(period))
print('raster start - step start us: median %.1f'%statistics.median(ov))
print('raster end - step end us: median %.1f'%statistics.median(r_after))
print('next step start - raster end us: median %.1f  p90 %.1f'%(statistics.median(gaps), sorted(gaps)[int(len(gaps)*0.9)]))
print('step dur %.1f raster dur %.1f'%(statistics.median([(e[1]-e[0])/1e3 for e in steps]), statistics.median([(e[1]-e[0])/1e3 for e in rast])))
PY
