mkdir -p gpurun_out/r3g
for v in "" _q768 _q640 _q512; do
  MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip$v.so python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/r3g/bench400$v.json 2>gpurun_out/r3g/err$v.txt
  python -c "
import json; d=json.load(open('gpurun_out/r3g/bench400$v.json')); print('$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel_alone']['avg_launch_ms'])"
done
MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_q640.so MGX_DEBUG_LAUNCH=1 python tools/dev/fused_timeline.py 2>&1 | grep -E "fused=|wait|mgx:|t= (1|2|3|4)"
