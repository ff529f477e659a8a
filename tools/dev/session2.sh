#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s2; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "state_only_full_size or f32_engine_one_step or owns_nothing" > $O/tests_new.txt 2>&1
tail -15 $O/tests_new.txt | cut -c1-300
grep "calm samples" $O/tests_new.txt | cut -c1-600
timeout 600 python tools/raster_waves_probe.py ClusterColour MatchRegions FindDupe MakeLine FixColour 2>&1 | grep -v amdgpu > $O/raster_waves_probe.txt
cat $O/raster_waves_probe.txt
for t in ClusterColour MoveToCorner MatchRegions; do
  echo "== $t"; MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so timeout 300 python tools/dev/raster_phase_clocks.py $t-Demo-v0 2>&1 | grep -v amdgpu.ids
done > $O/raster_phase_clocks.txt
cat $O/raster_phase_clocks.txt
