MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_clocks.so python tools/dev/placement_probe.py 2>&1 | grep -v amdgpu | head -12
bash tools/dev/run_ab_variant.sh _nocol
