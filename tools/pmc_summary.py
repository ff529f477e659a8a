"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (development tool).

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> > profiles/rNN_pmc_traffic.json

rocprofv3 reports both counters in KiB.  WRITE_SIZE is exact for 16 B/lane streaming stores (the 452 984 832 B fill
of the observation tensor in the same run reads 442 368.0); FETCH_SIZE on gfx950 tallies 128 B read requests at
64 B (MI355X_MICROARCH.md, "HBM"), so fetch bytes are doubled.
"""
import collections, csv, glob, json, statistics, sys

def per_kernel(d):
    f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        k = None if ('k_raster_deferred' in k or 'k_step_order' in k) else 'k_raster' if 'k_raster' in k else 'k_step' if 'k_step' in k else 'fill_u8' if 'FillFunctor<unsigned char>' in k else None
        if k:
            out[k].append(float(r['Counter_Value']) * 1024.0)
    return out

fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
res = {}
for k in ('k_raster', 'k_step'):
    if not fetch[k] or not write[k]:
        continue                      # (a state-only workload launches no rasteriser)
    fr, wr = statistics.median(fetch[k]), statistics.median(write[k])
    res[k] = {'launches_fetch_pass': len(fetch[k]), 'launches_write_pass': len(write[k]),
              'FETCH_SIZE_raw_bytes_median': fr, 'FETCH_SIZE_x2_bytes': 2 * fr, 'WRITE_SIZE_bytes_median': wr,
              'hbm_traffic_bytes_per_launch': 2 * fr + wr}
if 'fill_u8' in write:
    res['calibration'] = {'kernel': 'at::native FillFunctor<unsigned char> over the [N,96,96,12] u8 tensor',
                          'WRITE_SIZE_bytes': max(write['fill_u8'])}
# the stamp bench.py checks before it quotes this file: the kernel sources these counters were measured on (tools/csrc_hash.py)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.csrc_hash import csrc_sha16
res['_stamp'] = {'csrc_sha16': csrc_sha16()}
print(json.dumps(res, indent=1))
