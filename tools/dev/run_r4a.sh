# round 4, first A/B: phase Q one DPP row per queued pixel (libmagical_hip_q16.so) against the round-3 library (libmagical_hip_r3.so)
set -x
export MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip_q16.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_vectors.py -q -m gpu -x -k "render or stack or golden or preprocessors or ring or raster or obs" 2>&1 | tail -8
timeout 600 python tools/raster_consistency_sweep.py 2>&1 | tail -4
unset MGX_LIB_PATH
bash tools/dev/run_ab.sh _r3 _q16 2>&1 | grep -v "passed\|mismatch" 
