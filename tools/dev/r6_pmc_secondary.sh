#!/bin/bash
# the counter passes of regen_profiles.sh for the driver's line's two other workloads only (all-fp64 build, state-only engine)
R=${1:-r06}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final2; rm -rf $O; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py"; NS="--no-cpu-baseline --no-secondary"
cd /tmp
PA=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CU_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU
PB=SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM,SQ_INSTS_FLAT_FLATSEG,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_SCA
PC=SQ_THREAD_CYCLES_VALU,SQ_INSTS_VALU_FMA_F32,SQ_INSTS_VALU_FMA_F64,SQ_INSTS_VALU_ADD_F64,SQ_INSTS_VALU_MUL_F64,SQ_INSTS_SMEM,SQ_INSTS_BRANCH,SQ_INSTS_VALU_TRANS_F32
for t in mtc_lores4e_f64 mtc_state_only; do
  extra="--dtype f64"; [ $t = mtc_state_only ] && extra="--task MoveToCorner-Demo-v0"
  for c in FETCH_SIZE WRITE_SIZE; do
    MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc_${t}_$c -o run -- $B $NS --steps 100 $extra > /dev/null 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_${t}_FETCH_SIZE /tmp/pmc_${t}_WRITE_SIZE > $O/${R}_pmc_traffic_${t}.json
  for p in A B C; do
    eval set=\$P$p
    MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc ${set//,/ } -f csv -d /tmp/alu_${t}_$p -o run -- $B $NS --steps 60 $extra > /dev/null 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_alu_summary.py /tmp/alu_${t}_A /tmp/alu_${t}_B /tmp/alu_${t}_C > $O/${R}_pmc_alu_${t}.json
done
ls -la $O
