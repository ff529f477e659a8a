mkdir -p gpurun_out/r3d
for v in "" _v128 _v80 _v64; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/r3d/bench400$v.json 2>gpurun_out/r3d/err$v.txt
  python -c "
import json; d=json.load(open('gpurun_out/r3d/bench400$v.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_alone']['avg_launch_ms'], d['roofline']['other_kernels']['k_step']['avg_launch_ms'])"
done
for v in "" _v128; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 240 --warmup 20 --no-cpu-baseline --task ClusterColour-Demo-LoRes4E-v0 > gpurun_out/r3d/bench_cc$v.json 2>gpurun_out/r3d/err_cc$v.txt
  python -c "
import json; d=json.load(open('gpurun_out/r3d/bench_cc$v.json')); print('cc$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_alone']['avg_launch_ms'], d['roofline']['other_kernels']['k_step']['avg_launch_ms'])"
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_step_error or f64_engine or per_env_worlds_match or fused_step or lanes" 2>&1 | tail -4
