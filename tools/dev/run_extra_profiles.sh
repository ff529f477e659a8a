O=$GRAFT_REPO_ROOT/gpurun_out/extra; rm -rf $O; mkdir -p $O
for t in mtc:MoveToCorner-Demo-v0 cc:ClusterColour-Demo-v0; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_probe.so python tools/step_phase_probe.py ${t#*:} 2>&1 | grep -v amdgpu > $O/r03_step_phase_cycles_${t%%:*}.txt
done
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/fused_timeline.py 2>&1 | grep -v amdgpu > $O/r03_fused_timeline_mtc_lores4e.txt
python tools/dev/fused_sensitivity.py 2>&1 | grep iterations > $O/r03_fused_vs_step_length_mtc.txt
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/raster_phase_clocks.py 2>&1 | grep -v amdgpu > $O/r03_raster_phase_clocks_mtc.txt
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/nq_stats.py 2>&1 | grep -v amdgpu > $O/r03_raster_queue_load_by_task.txt
tools/sincos_check > $O/r03_sincos_fp64_accuracy.txt 2>&1
python tools/task_step_times.py 2>&1 | grep -v amdgpu > $O/r03_task_step_times.txt
head -30 $O/r03_step_phase_cycles_mtc.txt
