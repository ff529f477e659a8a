#!/bin/bash
# build the library of another revision's kernel sources for an A/B on one box: tools/dev/build_rev.sh <git-rev> <suffix> [-Dxxx ...]
#   -> magical_amd/libmagical_hip<suffix>.so   (run with MGX_LIB_PATH=$PWD/magical_amd/libmagical_hip<suffix>.so)
set -e
rev=$1; suf=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
d=$(mktemp -d)
git -C $root archive $rev magical_amd/csrc include | tar -x -C $d
cd $d/magical_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-parameter -Wno-extern-c-compat -pthread -o $root/magical_amd/libmagical_hip$suf.so "$@" mgx_api.hip mgx_world.cpp
rm -rf $d
ls -la $root/magical_amd/libmagical_hip$suf.so
