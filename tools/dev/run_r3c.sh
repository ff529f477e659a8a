mkdir -p gpurun_out/r3c
tools/sincos_check > gpurun_out/r3c/sincos.txt 2>&1; cat gpurun_out/r3c/sincos.txt
for v in "" _w3 _w4; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/r3c/bench400$v.json 2>gpurun_out/r3c/err$v.txt
  python -c "
import json; d=json.load(open('gpurun_out/r3c/bench400$v.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_alone']['avg_launch_ms'], d['roofline']['other_kernels']['k_step']['avg_launch_ms'])"
done
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_probe.so python tools/step_phase_probe.py MoveToCorner-Demo-v0 > gpurun_out/r3c/phase_mtc.txt 2>&1; cat gpurun_out/r3c/phase_mtc.txt
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_probe.so python tools/step_phase_probe.py ClusterColour-Demo-v0 > gpurun_out/r3c/phase_cc.txt 2>&1; cat gpurun_out/r3c/phase_cc.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_step_error or drift_within or f64_engine or reset_with_per_env" 2>&1 | tail -4
