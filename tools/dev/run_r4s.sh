# what the driver's 20-step window pays beyond 20 env-steps: collective, timing events
P='import json,sys; d=json.load(sys.stdin); print(sys.argv[1], round(d["value"]/1e6,3), round(d["ms_per_step"]*d["steps"],3), (d.get("collective") or {}).get("ms_on_stream"))'
for rep in 1 2 3; do
python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "$P" default
python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --no-collective 2>/dev/null | python -c "$P" no_collective
MGX_BENCH_TIMING_EVERY=1000 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "$P" no_events
MGX_BENCH_TIMING_EVERY=1000 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --no-collective 2>/dev/null | python -c "$P" neither
done
