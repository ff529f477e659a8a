cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc; rm -rf $O; mkdir -p $O
cd /tmp
for t in cc; do
  task=ClusterColour-Demo-LoRes4E-v0
  for c in FETCH_SIZE WRITE_SIZE; do
    MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc_${t}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --task $task > /dev/null 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_${t}_FETCH_SIZE /tmp/pmc_${t}_WRITE_SIZE > $O/r02_pmc_traffic_${t}_lores4e.json
  cat $O/r02_pmc_traffic_${t}_lores4e.json | python -c "
import json,sys; d=json.load(sys.stdin); print('$t', {k:(v['FETCH_SIZE_x2_bytes']/1e6, v['WRITE_SIZE_bytes_median']/1e6, v['hbm_traffic_bytes_per_launch']/1e6, v['launches_fetch_pass']) for k,v in d.items() if k!='calibration'})"
done
