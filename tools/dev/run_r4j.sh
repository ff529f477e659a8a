# round 4: the hand-off to the rasteriser after the last substep's position update (shipped) against the hand-off at the end of
# the step kernel (libmagical_hip_late.so = -DMGX_EARLY_HANDOFF=0); bench.py, 4096 envs, fused env-step
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "interchangeable or determinism or fused or full_size or lores4e_stack or checkpoint or ragged or terminal or rollouts" 2>&1 | tail -2
P='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,3), round(d["ms_per_step"],4), round(r["avg_launch_ms"],4), round(r["other_kernels"]["k_step"]["avg_launch_ms"],4))'
for rep in 1 2; do for v in _late ""; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$P" mtc$v
done; done
for t in ClusterColour MatchRegions FindDupe; do for v in _late ""; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary --steps 240 --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "$P" $t$v
done; done
for v in _late ""; do MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "$P" mtc20$v; done
