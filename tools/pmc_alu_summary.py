"""Summarise rocprofv3 --pmc passes of SQ counters per kernel: what actually bounds k_step and k_raster (development tool).

usage: python tools/pmc_alu_summary.py <passA_dir> <passB_dir> <passC_dir> > profiles/rNN_pmc_alu_<workload>.json

Each pass is `rocprofv3 --kernel-trace --pmc <up to 8 SQ counters> -f csv -d <dir> -o run -- python bench.py ...` (kernel-trace
only beside --pmc, MI355X_MICROARCH.md "rocprofv3 PMC slots": the SQ block has 8 slots per pass), run with MGX_NO_OVERLAP=1 because
counter collection serialises kernels.  Counter values are summed over the chip by rocprofv3; per kernel the MEDIAN over launches
is reported, next to the kernel's median duration in that same pass (kernel trace).

Derived figures (formulas spelled out so they can be recomputed):
  valu_util            = SQ_INSTS_VALU / duration / VALU_PEAK_WAVE_INSTS_PER_S, VALU_PEAK = 1024 SIMDs x 2.4 GHz / 2 cycles per
                         wave64 instruction (MI355X_MICROARCH.md "Per-instruction cycle constants": v_fma_f32 wave64 = 2 cyc;
                         = the 157 TFLOP/s dense fp32 vector peak / 128 flops per wave64 FMA)
  valu_busy            = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CU_CYCLES x 4 SIMDs / CU): share of SIMD-cycles with a VALU op in flight
                         (quad-cycle units, guide "s_memtime tick vs SQ PMC units")
  occupancy_waves_per_simd = SQ_WAVE_CYCLES x 4 / (duration x 2.4 GHz x 1024 SIMDs): average resident wavefronts per SIMD
  wait_share / issue_stall_share / active_share = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint)
  lds_conflict_share   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  scratch              : static footprint comes from the code object (.private_segment_fixed_size x 64 lanes x waves), see
                         profiles/rNN_kernel_resources.txt; SQ_INSTS_FLAT_FLATSEG / SQ_INSTS_VMEM count the instructions
"""
import collections, csv, glob, json, statistics, sys

CLOCK_HZ, N_SIMD = 2.4e9, 1024
VALU_PEAK = N_SIMD * CLOCK_HZ / 2.0


def kname(k):
    if 'k_raster_deferred' in k or 'k_step_order' in k:
        return None
    return 'k_raster' if 'k_raster' in k else 'k_step' if 'k_step' in k else None


def read_pass(d):
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)
    for r in csv.DictReader(open(f[0])) if f else []:
        k = kname(r['Kernel_Name'])
        if k:
            counters[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)
    for r in csv.DictReader(open(f[0])) if f else []:
        k = kname(r['Kernel_Name'])
        if k:
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
    return counters, dur


res = {'units': {'clock_hz': CLOCK_HZ, 'n_simd': N_SIMD, 'valu_peak_wave_insts_per_s': VALU_PEAK}, 'passes': sys.argv[1:]}
merged, durs = collections.defaultdict(dict), collections.defaultdict(list)
for d in sys.argv[1:]:
    c, du = read_pass(d)
    for k, cs in c.items():
        for name, vals in cs.items():
            merged[k][name] = statistics.median(vals)
        merged[k].setdefault('_launches', len(next(iter(cs.values()))))
    for k, v in du.items():
        durs[k].append(statistics.median(v))
for k, m in merged.items():
    g = lambda n: m.get(n)
    dur = statistics.median(durs[k]) if durs[k] else None
    out = {'launches_per_pass': m.pop('_launches', None), 'median_duration_s_under_pmc': dur, 'counters_median_per_launch': dict(sorted(m.items()))}
    der = {}
    if dur and g('SQ_INSTS_VALU') is not None:
        der['valu_util'] = g('SQ_INSTS_VALU') / dur / VALU_PEAK
    if g('SQ_ACTIVE_INST_VALU') is not None and g('SQ_BUSY_CU_CYCLES'):
        der['valu_busy'] = g('SQ_ACTIVE_INST_VALU') * 4 / (g('SQ_BUSY_CU_CYCLES') * 4)
    if dur and g('SQ_WAVE_CYCLES') is not None:
        der['occupancy_waves_per_simd'] = g('SQ_WAVE_CYCLES') * 4 / (dur * CLOCK_HZ * N_SIMD)
    if g('SQ_WAVE_CYCLES'):
        for name, key in (('SQ_WAIT_ANY', 'wait_share'), ('SQ_WAIT_INST_ANY', 'issue_stall_share'), ('SQ_ACTIVE_INST_ANY', 'active_share')):
            if g(name) is not None:
                der[key] = g(name) / g('SQ_WAVE_CYCLES')
    if g('SQ_LDS_IDX_ACTIVE'):
        der['lds_conflict_share'] = (g('SQ_LDS_BANK_CONFLICT') or 0.0) / g('SQ_LDS_IDX_ACTIVE')
    if g('SQ_INSTS_VALU') and g('SQ_WAVES'):
        der['valu_insts_per_wave'] = g('SQ_INSTS_VALU') / g('SQ_WAVES')
        for n in ('SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM', 'SQ_INSTS_BRANCH'):
            if g(n) is not None:
                der[n.lower()[3:] + '_per_wave'] = g(n) / g('SQ_WAVES')
    out['derived'] = der
    res[k] = out
# the stamp bench.py checks before it quotes this file: the kernel sources these counters were measured on (tools/csrc_hash.py)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.csrc_hash import csrc_sha16
res['_stamp'] = {'csrc_sha16': csrc_sha16()}
print(json.dumps(res, indent=1))
