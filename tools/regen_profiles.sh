#!/bin/bash
# Re-measure the round's profile set on the GPU box (run through gpurun from the repo root:
#   gpurun --timeout 3600 -- 'bash tools/regen_profiles.sh r03'); results land in gpurun_out/final/, to be copied into profiles/.
R=${1:-r06}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final; rm -rf $O; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py"
NS="--no-cpu-baseline --no-secondary"
CC=ClusterColour-Demo-LoRes4E-v0
timeout 600 $B > $O/${R}_bench_mtc_lores4e.json 2> $O/err.txt
timeout 300 $B --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_mtc_lores4e_20steps.json 2>> $O/err.txt
MGX_NO_OVERLAP=1 timeout 300 $B $NS > $O/${R}_bench_mtc_lores4e_serial.json 2>> $O/err.txt
timeout 300 $B $NS --task $CC > $O/${R}_bench_clustercolour_lores4e.json 2>> $O/err.txt
MGX_NO_OVERLAP=1 timeout 300 $B $NS --task $CC > $O/${R}_bench_clustercolour_lores4e_serial.json 2>> $O/err.txt
timeout 300 $B $NS --task MoveToCorner-Demo-v0 > $O/${R}_bench_mtc_state_only.json 2>> $O/err.txt
timeout 300 $B $NS --dtype f64 > $O/${R}_bench_mtc_lores4e_f64.json 2>> $O/err.txt
timeout 300 $B $NS --dtype f64 --task $CC > $O/${R}_bench_clustercolour_lores4e_f64.json 2>> $O/err.txt
timeout 300 $B $NS --task MoveToCorner-Demo-LoResCHW4E-v0 --obs-ring 35 > $O/${R}_bench_mtc_loreschw4e_ring35.json 2>> $O/err.txt
timeout 300 $B $NS --task MoveToCorner-Demo-LoResCHW4E-v0 > $O/${R}_bench_mtc_loreschw4e_inplace_stack.json 2>> $O/err.txt
timeout 600 $B --config5 --envs5 1024 > $O/${R}_bench_config5_1gpu_8x1024.json 2>> $O/err.txt
cd /tmp
# kernel-trace statistics of the same commands (fused, one-after-the-other, ClusterColour)
rocprofv3 --kernel-trace --stats -f csv -d /tmp/p1 -o mtc -- $B $NS > $O/${R}_bench_mtc_lores4e_under_rocprof.json 2> /dev/null
cp /tmp/p1/mtc_kernel_stats.csv $O/${R}_bench_mtc_lores4e_kernel_stats.csv
MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p3 -o mtcs -- $B $NS > /dev/null 2>&1
cp /tmp/p3/mtcs_kernel_stats.csv $O/${R}_bench_mtc_lores4e_serial_kernel_stats.csv
rocprofv3 --kernel-trace --stats -f csv -d /tmp/p2 -o cc -- $B $NS --task $CC > /dev/null 2>&1
cp /tmp/p2/cc_kernel_stats.csv $O/${R}_bench_cc_lores4e_kernel_stats.csv
# HBM traffic: one counter per pass, kernel-trace only (MI355X_MICROARCH.md); kernels one after the other
for t in mtc cc; do
  task=MoveToCorner-Demo-LoRes4E-v0; [ $t = cc ] && task=$CC
  for c in FETCH_SIZE WRITE_SIZE; do
    MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc_${t}_$c -o run -- $B $NS --steps 100 --task $task > /dev/null 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_${t}_FETCH_SIZE /tmp/pmc_${t}_WRITE_SIZE > $O/${R}_pmc_traffic_${t}_lores4e.json
done
# ... and of the driver's line's two other workloads (round 6): the all-fp64 build, the state-only engine
for t in mtc_lores4e_f64 mtc_state_only; do
  extra="--dtype f64"; [ $t = mtc_state_only ] && extra="--task MoveToCorner-Demo-v0"
  for c in FETCH_SIZE WRITE_SIZE; do
    MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc_${t}_$c -o run -- $B $NS --steps 100 $extra > /dev/null 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_${t}_FETCH_SIZE /tmp/pmc_${t}_WRITE_SIZE > $O/${R}_pmc_traffic_${t}.json
done
# what bounds the kernels: SQ counters, three passes of eight (tools/pmc_alu_summary.py), kernels one after the other
PA=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CU_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU
PB=SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM,SQ_INSTS_FLAT_FLATSEG,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_SCA
PC=SQ_THREAD_CYCLES_VALU,SQ_INSTS_VALU_FMA_F32,SQ_INSTS_VALU_FMA_F64,SQ_INSTS_VALU_ADD_F64,SQ_INSTS_VALU_MUL_F64,SQ_INSTS_SMEM,SQ_INSTS_BRANCH,SQ_INSTS_VALU_TRANS_F32
for t in mtc cc; do
  task=MoveToCorner-Demo-LoRes4E-v0; [ $t = cc ] && task=$CC
  for p in A B C; do
    eval set=\$P$p
    MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc ${set//,/ } -f csv -d /tmp/alu_${t}_$p -o run -- $B --no-cpu-baseline --no-secondary --steps 60 --task $task > /dev/null 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_alu_summary.py /tmp/alu_${t}_A /tmp/alu_${t}_B /tmp/alu_${t}_C > $O/${R}_pmc_alu_${t}_lores4e.json
done
for t in mtc_lores4e_f64 mtc_state_only; do
  extra="--dtype f64"; [ $t = mtc_state_only ] && extra="--task MoveToCorner-Demo-v0"
  for p in A B C; do
    eval set=\$P$p
    MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc ${set//,/ } -f csv -d /tmp/alu_${t}_$p -o run -- $B --no-cpu-baseline --no-secondary --steps 60 $extra > /dev/null 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_alu_summary.py /tmp/alu_${t}_A /tmp/alu_${t}_B /tmp/alu_${t}_C > $O/${R}_pmc_alu_${t}.json
done
cd $GRAFT_REPO_ROOT
# the parity tables the GPU tests print (one-step error quantiles against the oracle and its replicas, drift against the perturbation
# envelope, the fp64 build's absolute one-step error with its contact-coverage shares and offenders) and the long drift table
timeout 1700 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "one_step or drift or contact_coverage or tracks_oracle" 2>&1 | grep -v "amdgpu.ids" > $O/${R}_gpu_parity_tables.txt
timeout 1200 python tools/drift_table.py --envs 32 --norm l2 2>&1 | grep -v "amdgpu.ids" > $O/${R}_pose_drift_f64_vs_oracle.txt
timeout 1200 python tools/drift_table.py --envs 48 --norm linf 2>&1 | grep -v "amdgpu.ids" > $O/${R}_pose_drift_vs_oracle_envelope.txt
python tools/kernel_resources.py > $O/${R}_kernel_resources.txt 2>> $O/err.txt
timeout 1500 python tools/rollout_all_tasks.py --variant all --envs 4096 > $O/${R}_rollout_all_60_variants_4096x1gpu.jsonl 2>> $O/err.txt
timeout 300 python tools/task_step_times.py 2>&1 | grep -v amdgpu > $O/${R}_task_step_times.txt
timeout 600 python tools/lds_table.py 2>&1 | grep -v amdgpu > $O/${R}_lds_footprints_all_variants.txt
cd $O; for f in ${R}_bench*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],4), r.get('avg_launch_ms'), (r.get('other_kernels') or {}).get('k_step',{}).get('avg_launch_ms'), r.get('frac'), (r.get('kernel_alone') or {}).get('avg_launch_ms'), (r.get('kernel_alone') or {}).get('frac'))
except Exception as ex: print('$f', 'ERR', ex)
PY
done
python - <<'PY'
import csv,glob,json
for f in sorted(glob.glob('*kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        if 'mgx::' in r['Name']: print(f[:40], r['Name'][10:36], r['Calls'], r['AverageNs'])
for f in sorted(glob.glob('*pmc_traffic*.json')):
    d=json.load(open(f)); print(f, {k:(round(v['FETCH_SIZE_x2_bytes']/1e6,1), round(v['WRITE_SIZE_bytes_median']/1e6,1), round(v['hbm_traffic_bytes_per_launch']/1e6,1)) for k,v in d.items() if k not in ('calibration', '_stamp')})
PY
cat ${R}_task_step_times.txt; wc -l ${R}_rollout_all_60_variants_4096x1gpu.jsonl
# development builds (python -c "import os; from magical_amd import _native; r = os.getcwd();
#   _native.build(force=True, defines=['MGX_STEP_PROBE'], out=r + '/magical_amd/libmagical_hip_probe.so');
#   _native.build(force=True, defines=['MGX_RASTER_CLOCKS'], out=r + '/magical_amd/libmagical_hip_clocks.so')" before the gpurun call): phase cycles, timelines
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/extra; rm -rf $O; mkdir -p $O
if [ -f magical_amd/libmagical_hip_probe.so ]; then
  for t in mtc:MoveToCorner-Demo-v0 cc:ClusterColour-Demo-v0; do
    MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_probe.so python tools/step_phase_probe.py ${t#*:} 2>&1 | grep -v amdgpu > $O/${R}_step_phase_cycles_${t%%:*}.txt
  done
fi
if [ -f magical_amd/libmagical_hip_clocks.so ]; then
  for t in ClusterColour MoveToCorner MatchRegions; do
    MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so timeout 300 python tools/dev/fused_occupancy.py $t-Demo-LoRes4E-v0 2>&1 | grep -v amdgpu > $O/${R}_fused_occupancy_$t.txt
  done
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/fused_timeline.py 2>&1 | grep -v amdgpu > $O/${R}_fused_timeline_mtc_lores4e.txt
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/raster_phase_clocks.py 2>&1 | grep -v amdgpu > $O/${R}_raster_phase_clocks_mtc.txt
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/nq_stats.py 2>&1 | grep -v amdgpu > $O/${R}_raster_queue_load_by_task.txt
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/placement_probe.py 2>&1 | grep -v amdgpu | head -12 > $O/${R}_fused_placement_mtc_lores4e.txt
fi
if [ -f magical_amd/libmagical_hip_rprobe.so ]; then      # (-DMGX_RASTER_PROBE build: the rasteriser truncated after each phase, under rocprofv3 --pmc)
  bash tools/dev/raster_phase_pmc.sh 2>&1 | grep -v amdgpu > $O/${R}_raster_phase_instructions_mtc.txt
fi
bash tools/dev/step_gaps.sh 2>&1 | grep -v amdgpu > $O/${R}_fused_step_gaps_mtc.txt
cd $GRAFT_REPO_ROOT
for t in ClusterColour-TestAll-LoRes4E-v0 MatchRegions-TestCountPlus-LoRes4E-v0; do
  timeout 300 python tools/reset_profile.py $t 2>&1 | grep -v amdgpu.ids > $O/${R}_reset_profile_$t.txt
done
# resets ten env-steps apart (the profile above) against an episode's spacing; against the size of the host pool
for t in ClusterColour-TestAll-LoRes4E-v0 MatchRegions-TestCountPlus-LoRes4E-v0; do for gap in 10 80; do
  MGX_DEBUG_VARIANTS=1 timeout 300 python tools/reset_profile.py $t $gap 2>&1 | grep "steady\|reset [0-9]:\|mgx: set_env" | tail -15 | cut -c1-215
done; done > $O/${R}_reset_spacing_10_vs_80_steps.txt 2>&1
for T in 32 48 64 96 128; do for t in ClusterColour-TestAll-LoRes4E-v0 MatchRegions-TestCountPlus-LoRes4E-v0; do
  echo "== MGX_HOST_THREADS=$T $t"; MGX_HOST_THREADS=$T timeout 300 python tools/reset_profile.py $t 2>&1 | grep "steady-state\|under cProfile"
done; done > $O/${R}_reset_threads_pool.txt 2>&1
timeout 600 python tools/window20_probe.py 2>&1 | grep -v amdgpu.ids > $O/${R}_window20_probe.txt
timeout 600 python tools/raster_consistency_sweep.py 2>&1 | tail -20 > $O/${R}_raster_consistency_sweep_tail.txt
# the fused hand-off under load: 8 fused engines on 8 streams (configs[4]'s shape), 20 000 env-steps each, shipped and with the failure paths forced
{ timeout 900 python tools/dev/hang_hunt.py --fleet 20000 1024; FORCE=1 timeout 900 python tools/dev/hang_hunt.py --fleet 5000 1024; } 2>&1 | grep -v amdgpu.ids > $O/${R}_hang_hunt_fleet.txt
timeout 900 python tools/fused_soak.py 2>&1 | grep -v amdgpu.ids > $O/${R}_fused_soak_600steps_all_tasks.txt
head -30 $O/${R}_step_phase_cycles_mtc.txt
ls $GRAFT_REPO_ROOT/gpurun_out/final $GRAFT_REPO_ROOT/gpurun_out/extra | head -80
