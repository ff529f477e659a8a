#!/bin/bash
# round 5, GPU session 1: new tests, the driver's line with pose_l2, ClusterColour baseline + both kernels on one time axis
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "state_only_full_size or f32_engine_one_step or owns_nothing" > $O/tests_new.txt 2>&1
tail -5 $O/tests_new.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
CC=ClusterColour-Demo-LoRes4E-v0
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 240 --task $CC > $O/bench_cc.json 2>> $O/err.txt
MGX_NO_OVERLAP=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 240 --task $CC > $O/bench_cc_serial.json 2>> $O/err.txt
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_mtc.json 2>> $O/err.txt
for t in ClusterColour MatchRegions MoveToCorner; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so timeout 300 python tools/dev/fused_occupancy.py $t-Demo-LoRes4E-v0 2>&1 | grep -v amdgpu.ids > $O/fused_occupancy_$t.txt
done
timeout 300 python tools/task_step_times.py 2>&1 | grep -v amdgpu > $O/task_step_times.txt
if [ -f magical_amd/libmagical_hip_probe.so ]; then
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_probe.so timeout 300 python tools/step_phase_probe.py ClusterColour-Demo-v0 2>&1 | grep -v amdgpu > $O/step_phase_cycles_cc.txt
fi
python - <<'PY'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/s1'
for f in sorted(glob.glob(O+'/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print(os.path.basename(f), round(d['value']/1e6,3), round(d['ms_per_step'],4), r.get('avg_launch_ms'), (r.get('kernel_alone') or {}).get('avg_launch_ms'), r.get('traffic_source'))
        if 'pose_l2' in d.get('config',{}): print(json.dumps(d['config']['pose_l2'])[:1500])
        if 'secondary' in d: print({k:v for k,v in d['secondary'].items() if k.endswith('_per_s')})
    except Exception as ex: print(f,'ERR',ex)
PY
cat $O/task_step_times.txt; head -60 $O/fused_occupancy_ClusterColour.txt
