#!/bin/bash
# per-phase instruction counts of k_raster (rprobe build): one line per stop level and layout
cd $GRAFT_REPO_ROOT
bash tools/dev/raster_phase_pmc.sh 2>&1 | awk '{print substr($0,1,260)}' | awk 'NR==1 || (NR-2)%5==0' | tail -14
