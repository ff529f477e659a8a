"""Registers, spills and scratch bytes per lane of every kernel in a built libmagical_hip.so (llvm-readelf --notes of its gfx950 image).
usage: python tools/kernel_resources.py [path/to/lib.so] [name filter]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
if len(sys.argv) > 1 and sys.argv[1].endswith('.so'):
    os.environ['MGX_LIB_PATH'] = os.path.abspath(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
else:
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
import bench  # noqa: E402

print('# code object notes of %s (llvm-readelf --notes, gfx950): registers, spills, scratch bytes per lane' % os.environ.get('MGX_LIB_PATH', 'the shipped libmagical_hip.so'))
for name, r in sorted(bench.kernel_resources().items()):
    if flt in name:
        print('%-110s vgpr %3d agpr %3d sgpr_spill %3d vgpr_spill %3d scratch_B_per_lane %4d' % (name, r['vgpr'], r['agpr'], r['sgpr_spill'], r['vgpr_spill'], r['scratch_bytes_per_lane']))
