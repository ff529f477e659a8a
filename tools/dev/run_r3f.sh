mkdir -p gpurun_out/r3f; O=$GRAFT_REPO_ROOT/gpurun_out/r3f
B="python $GRAFT_REPO_ROOT/bench.py"
( time $B --steps 20 --warmup 5 > $O/bench20.json 2> $O/err20.txt ) 2> $O/time20.txt; tail -3 $O/time20.txt
python - <<PY
import json; d=json.load(open('$O/bench20.json'))
print(d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['scratch_bytes'], d['roofline']['kernel_alone']['avg_launch_ms'])
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='roofline'}) for k,v in d['secondary'].items()})
print(d['cpu_baseline'])
PY
cd /tmp; export TMPDIR=/tmp
PA=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CU_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU
PB=SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM,SQ_INSTS_FLAT_FLATSEG,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_SCA
PC=SQ_THREAD_CYCLES_VALU,SQ_INSTS_VALU_FMA_F32,SQ_INSTS_VALU_FMA_F64,SQ_INSTS_VALU_ADD_F64,SQ_INSTS_VALU_MUL_F64,SQ_INSTS_SMEM,SQ_INSTS_BRANCH,SQ_INSTS_VALU_TRANS_F32
for t in mtc; do
  task=MoveToCorner-Demo-LoRes4E-v0
  for p in A B C; do
    eval set=\$P$p
    MGX_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc ${set//,/ } -f csv -d /tmp/alu_${t}_$p -o run -- $B --no-cpu-baseline --no-secondary --steps 60 --task $task > $O/alu_$p.log 2>&1
    ls /tmp/alu_${t}_$p/* | head -3
  done
  python $GRAFT_REPO_ROOT/tools/pmc_alu_summary.py /tmp/alu_${t}_A /tmp/alu_${t}_B /tmp/alu_${t}_C > $O/r03_pmc_alu_${t}_lores4e.json
done
cat $O/r03_pmc_alu_mtc_lores4e.json | head -120
