#!/bin/bash
# round 6: the write-once rasteriser path against round 5's kernels on one box: bash tools/dev/r6_wonce_ab.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
MGX_DEBUG_LAUNCH=1 python -c "
import magical_amd
e = magical_amd.make('MoveToCorner-Demo-LoRes4E-v0', n_envs=64, device='cuda:0'); e.reset(); e.close()" 2>&1 | grep k_raster
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "render or stack or preprocessors or ring or raster or obs or fused" 2>&1 | tail -5
timeout 600 python tools/raster_consistency_sweep.py 2>&1 | tail -10
NO_TESTS=1 bash tools/dev/ab_quick.sh "" _base
} > gpurun_out/r6_wonce_ab.log 2>&1
