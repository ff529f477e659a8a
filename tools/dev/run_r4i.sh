# round 4: k_env_order after its rewrite (classes from LDS, non-empty classes only): duration, digests, A/B again
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "interchangeable or determinism or fused_step_render_equals or full_size" 2>&1 | tail -2
B="python $PWD/bench.py --no-cpu-baseline --no-secondary"
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,3), round(d["ms_per_step"],4), round(r["avg_launch_ms"],4), round(r["other_kernels"]["k_step"]["avg_launch_ms"],4))'
for rep in 1 2; do
  MGX_NO_ENV_PACK=1 $B 2>/dev/null | python -c "$P" mtc_nopack
  $B 2>/dev/null | python -c "$P" mtc_pack
done
MGX_NO_ENV_PACK=1 $B --steps 240 --task MatchRegions-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" mr_nopack
$B --steps 240 --task MatchRegions-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" mr_pack
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/p1 -o mtc -- $B > /dev/null 2>&1
cp /tmp/p1/mtc_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r04_bench_mtc_lores4e_kernel_stats.csv
grep "mgx::" /tmp/p1/mtc_kernel_stats.csv | cut -c1-120
