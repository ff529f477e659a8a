#!/bin/bash
# Host-side A/B of csrc/mgx_world.cpp between a git revision (default: HEAD~1) and the working tree, no GPU needed:
#   world_blob_hash.cpp  -- hash of the serialised blobs (header, ints, reals, pose words; with and without the draw list) of 9 000
#                           random world variants of three tasks;
#   placement_hash.cpp   -- hash of poses, return codes, final MT19937 positions of 4 500 randomise_all_poses calls (goal regions with
#                           per-env sizes, limits, an ignored entity) and of placement_collides on the results, with the time per call.
# Equal hashes = the same bits.  usage: bash tools/dev/world_host_ab.sh [revision]
set -e
cd "$(dirname "$0")/../.."
REV=${1:-HEAD~1}
T=$(mktemp -d)
git show $REV:magical_amd/csrc/mgx_world.cpp > $T/old_world.cpp
mkdir -p $T/oldinc
for h in mgx_world.h mgx_tmpl.h; do git show $REV:magical_amd/csrc/$h > $T/oldinc/$h; done
for prog in world_blob_hash placement_hash; do
  g++ -O2 -std=c++17 -I$T/oldinc tools/dev/$prog.cpp $T/old_world.cpp -o $T/${prog}_old
  g++ -O2 -std=c++17 -Imagical_amd/csrc tools/dev/$prog.cpp magical_amd/csrc/mgx_world.cpp -o $T/${prog}_new
  echo "$prog  $REV: $($T/${prog}_old)"
  echo "$prog  tree: $($T/${prog}_new)"
done
rm -rf $T
