#!/bin/bash
# SQ counters of k_raster for several library builds on one box: bash tools/dev/pmc_ab.sh "" _base
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  bash tools/dev/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" MGX_LIB_PATH=$GRAFT_REPO_ROOT/magical_amd/libmagical_hip$v.so 2>&1 | grep k_raster
  bash tools/dev/pmc_quick.sh "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_LDS" MGX_LIB_PATH=$GRAFT_REPO_ROOT/magical_amd/libmagical_hip$v.so 2>&1 | grep k_raster
done
