set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -s 2>&1 | tail -300 > gpurun_out/r02_gputests_a.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_mtc_20.json 2> gpurun_out/r02_bench_mtc_20.err
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_mtc_400.json 2>> gpurun_out/r02_bench_mtc_20.err
python bench.py --no-cpu-baseline --task ClusterColour-Demo-LoRes4E-v0 > gpurun_out/r02_bench_cc_400.json 2>> gpurun_out/r02_bench_mtc_20.err
python bench.py --no-cpu-baseline --dtype f64 > gpurun_out/r02_bench_mtc_f64.json 2>> gpurun_out/r02_bench_mtc_20.err
python bench.py --no-cpu-baseline --dtype f64 --task ClusterColour-Demo-LoRes4E-v0 > gpurun_out/r02_bench_cc_f64.json 2>> gpurun_out/r02_bench_mtc_20.err
# PMC passes for ClusterColour (separate passes, kernel-trace only)
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cc_fetch -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 5 --task ClusterColour-Demo-LoRes4E-v0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cc_write -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 5 --task ClusterColour-Demo-LoRes4E-v0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/pmc_cc_fetch gpurun_out/pmc_cc_write > gpurun_out/r02_pmc_traffic_cc_lores4e.json 2> gpurun_out/pmc_cc.err
rm -rf gpurun_out/pmc_cc_fetch gpurun_out/pmc_cc_write
tail -5 gpurun_out/r02_gputests_a.log
cat gpurun_out/r02_smoke.log | tail -2
