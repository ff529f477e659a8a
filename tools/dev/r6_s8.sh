#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_s8; mkdir -p $O
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; ka=(r.get("kernel_alone") or {}).get("avg_launch_ms") or {}; print(sys.argv[1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms; alone k_raster", round(ka.get("k_raster",0),4), "k_step", round(ka.get("k_step",0),4))'
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary $2 2>/dev/null | python -c "$P" "$1"; }
{
for t in MoveToCorner-TestAll MoveToCorner-TestShape MakeLine-TestAll MakeLine-TestShape FixColour-TestAll FixColour-TestCountPlus FindDupe-TestAll FindDupe-TestShape; do
  run "$t narrow" "--task $t-LoRes4E-v0"
  MGX_RASTER_WIDE_ENV=1 run "$t wide(r5)" "--task $t-LoRes4E-v0"
done
timeout 900 python tools/raster_consistency_sweep.py 2>&1 | tail -10
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "variant or render or stack or preprocessors or ring or raster or obs or fused or batched or env_world or terminal" 2>&1 | tail -3
} > $O/ab.txt 2>&1
