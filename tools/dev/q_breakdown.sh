#!/bin/bash
# where phase Q's instructions go: the per-phase probe (rprobe build) and the whole kernel without the line loops / without the polygon coverage (wrong pixels)
cd $GRAFT_REPO_ROOT
bash tools/dev/raster_phase_pmc.sh 2>&1 | awk '{print substr($0,1,260)}' | awk 'NR==1 || (NR-2)%5==0' | tail -14
for v in "" _qnl _qnp; do
  bash tools/dev/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM" MGX_LIB_PATH=$GRAFT_REPO_ROOT/magical_amd/libmagical_hip$v.so 2>&1 | grep k_raster
done
