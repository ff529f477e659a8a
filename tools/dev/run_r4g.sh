# round 4: heavy envs together (env order by the previous launch's cost keys) against MGX_NO_ENV_PACK=1, same library
B="python bench.py --no-cpu-baseline --no-secondary"
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,3), round(d["ms_per_step"],4), "fused launch ms", {k: round(v,4) for k,v in [(r["kernel"], r["avg_launch_ms"])]+[(k,v["avg_launch_ms"]) for k,v in r["other_kernels"].items()]}, "alone", (r.get("kernel_alone") or {}).get("avg_launch_ms"))'
for rep in 1 2; do
  MGX_NO_ENV_PACK=1 $B --steps 400 --warmup 20 2>/dev/null | python -c "$P" mtc_nopack
  $B --steps 400 --warmup 20 2>/dev/null | python -c "$P" mtc_pack
done
MGX_NO_ENV_PACK=1 $B --steps 20 --warmup 5 2>/dev/null | python -c "$P" mtc20_nopack
$B --steps 20 --warmup 5 2>/dev/null | python -c "$P" mtc20_pack
for t in MatchRegions-Demo-LoRes4E-v0 FixColour-Demo-LoRes4E-v0 MoveToCorner-Demo-v0; do
  MGX_NO_ENV_PACK=1 $B --steps 240 --warmup 20 --task $t 2>/dev/null | python -c "$P" ${t}_nopack
  $B --steps 240 --warmup 20 --task $t 2>/dev/null | python -c "$P" ${t}_pack
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "determinism or lanes or fused or rollouts or full_size or lores4e_stack or checkpoint or ragged" 2>&1 | tail -3
