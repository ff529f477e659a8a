#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in ClusterColour-Demo-LoRes4E-v0 MatchRegions-Demo-LoRes4E-v0 FindDupe-Demo-LoRes3EA-v0 ClusterShape-TestAll-LoRes4E-v0 MoveToCorner-Demo-LoResCHW4E-v0; do
  TASK=$t REPS=2 STEPS=${STEPS:-8000} bash tools/dev/hang_hunt.sh 2>&1 | cut -c1-250
done
