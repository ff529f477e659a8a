"""Static instruction counts per phase of k_step (development aid).

Compiles magical_amd/csrc/mgx_step.hip alone (-S, device only, gfx950, ~3 s) with -DMGX_PHASE_MARKERS, which
puts every phase's name into the assembly as a comment, and counts the instructions between markers, by class.
Usage: python tools/step_asm_phases.py [R P L] [-Dxxx ...]   (default: float double 16)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('-')]
    defs = [a for a in sys.argv[1:] if a.startswith('-')]
    R, P, L = (args + ['float', 'double', '16'])[:3] if len(args) >= 3 else ('float', 'double', '16')
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, 'mark.hip')
        with open(src, 'w') as f:
            f.write('#define MGX_PHASE_MARKERS 1\n#include "%s/magical_amd/csrc/mgx_step.hip"\n' % ROOT)
            kern = 'k_step_env<%s,%s>' % (R, P) if L == '64' else 'k_step<%s,%s,%s>' % (R, P, L)
            f.write('template __global__ void mgx::%s(mgx::TmplDev, %s*, %s*, int32_t*, const int32_t*, uint8_t*, int,int,int,int, mgx::StepHandoff);\n' % (kern, P, R))
        out = os.environ.get('MGX_ASM_OUT', os.path.join(d, 'mark.s'))
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', '-o', out, src,
                               '-Wno-unused-parameter', '-Wno-unused-command-line-argument'] + defs)
        lines = open(out).read().split('\n')
    phase, counts, order, maxv = 'prologue', collections.defaultdict(collections.Counter), ['prologue'], {}
    for l in lines:
        m = re.search(r'; MGX_PHASE (\w+)', l)
        if m:
            phase = m.group(1)
            if phase not in order:
                order.append(phase)
            continue
        if re.match(r'\s*\.(vgpr_count|sgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):', l):
            print(l.strip())
        m = re.match(r'\t([a-z_0-9]+)', l)
        if not m or l.startswith('\t.') or l.startswith('\t;'):
            continue
        op = m.group(1)
        cls = ('salu' if op.startswith('s_') and not op.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch')) else
               'wait' if op.startswith('s_waitcnt') else 'nop' if op.startswith('s_nop') else
               'branch' if op.startswith(('s_cbranch', 's_branch')) else 'lds' if op.startswith('ds_') else
               'vmem' if op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')) else
               'f64' if '_f64' in op else 'valu')
        counts[phase][cls] += 1
        for r in re.findall(r'\bv(\d+)\b', l) + [x[1] for x in re.findall(r'\bv\[(\d+):(\d+)\]', l)]:
            maxv[phase] = max(maxv.get(phase, 0), int(r))
    cols = ['valu', 'f64', 'lds', 'salu', 'wait', 'nop', 'branch', 'vmem']
    print('%-22s %6s  ' % ('phase', 'total') + ' '.join('%6s' % c for c in cols) + '   max v#')
    for ph in order:
        c = counts[ph]
        print('%-22s %6d  ' % (ph, sum(c.values())) + ' '.join('%6d' % c[k] for k in cols) + '   %5d' % maxv.get(ph, 0))
    print('%-22s %6d' % ('all', sum(sum(c.values()) for c in counts.values())))


if __name__ == '__main__':
    main()
