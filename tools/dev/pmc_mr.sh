cd $GRAFT_REPO_ROOT
export BENCH_ARGS="--task MatchRegions-Demo-LoRes4E-v0"
for v in "" _narrow; do
  bash tools/dev/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" MGX_LIB_PATH=$GRAFT_REPO_ROOT/magical_amd/libmagical_hip$v.so 2>&1 | grep k_raster
done
