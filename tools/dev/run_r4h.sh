# round 4: what the driver's 20-step window spends beyond 20 x the steady env-step: the collective, the timing events
B="python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5"
P='import json,sys; d=json.load(sys.stdin); print(sys.argv[1], round(d["value"]/1e6,3), round(d["ms_per_step"]*d["steps"],3), "ms window;", (d.get("collective") or {}).get("ms"))'
for rep in 1 2 3; do
  $B 2>/dev/null | python -c "$P" default
  $B --no-collective 2>/dev/null | python -c "$P" no_collective
  MGX_BENCH_TIMING_EVERY=5 $B 2>/dev/null | python -c "$P" timing_every_5
  MGX_BENCH_TIMING_EVERY=5 $B --no-collective 2>/dev/null | python -c "$P" both
done
