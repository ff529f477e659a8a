cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
for cfg in "mtc:MoveToCorner-Demo-LoRes4E-v0" "cc:ClusterColour-Demo-LoRes4E-v0"; do
  key=${cfg%%:*}; task=${cfg##*:}
  rm -rf /tmp/prof_$key /tmp/pf_$key /tmp/pw_$key
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$key -- python $R/bench.py --no-cpu-baseline --task $task > $O/r02_bench_${key}_lores4e_under_rocprof.json 2> $O/rocprof_$key.err
  f=$(find /tmp/prof_$key -name "*kernel_stats.csv" | head -1); cp "$f" $O/r02_bench_${key}_lores4e_kernel_stats.csv
  MGX_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d /tmp/pf_$key -- python $R/bench.py --no-cpu-baseline --steps 60 --warmup 5 --task $task > /dev/null 2> $O/pmc_f_$key.err
  MGX_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d /tmp/pw_$key -- python $R/bench.py --no-cpu-baseline --steps 60 --warmup 5 --task $task > /dev/null 2> $O/pmc_w_$key.err
  python $R/tools/pmc_summary.py /tmp/pf_$key /tmp/pw_$key > $O/r02_pmc_traffic_${key}_lores4e.json 2> $O/pmc_sum_$key.err
done
rm -rf /tmp/prof_ser
MGX_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_ser -- python $R/bench.py --no-cpu-baseline > $O/r02_bench_mtc_lores4e_serial_under_rocprof.json 2>> $O/rocprof_mtc.err
f=$(find /tmp/prof_ser -name "*kernel_stats.csv" | head -1); cp "$f" $O/r02_bench_mtc_lores4e_serial_kernel_stats.csv
cd $R
python bench.py > $O/r02_bench_mtc_lores4e.json 2> $O/r02_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_bench_mtc_lores4e_20steps.json 2>> $O/r02_bench.err
MGX_NO_OVERLAP=1 python bench.py --no-cpu-baseline > $O/r02_bench_mtc_lores4e_serial.json 2>> $O/r02_bench.err
python bench.py --no-cpu-baseline --task ClusterColour-Demo-LoRes4E-v0 > $O/r02_bench_cc_lores4e.json 2>> $O/r02_bench.err
python bench.py --no-cpu-baseline --task MoveToCorner-Demo-v0 > $O/r02_bench_mtc_state_only.json 2>> $O/r02_bench.err
python bench.py --no-cpu-baseline --config5 --envs5 1024 --steps 120 --warmup 5 > $O/r02_bench_config5_1gpu_8x1024.json 2>> $O/r02_bench.err
tail -3 $O/r02_bench.err
