"""CPU tool: operation counts of the rasteriser's resolve path from the host emulation (development tool)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.emu.emu import EmuBatch, lib
from tests.util import new_ref, ref_entities_as_tuples

task = sys.argv[1] if len(sys.argv) > 1 else 'MoveToCorner'
n, T = 16, 40
ref = new_ref(task)
ents, max_steps = ref_entities_as_tuples(ref), ref.max_episode_steps
em = EmuBatch(ents, max_steps, n, 'mixed')
em.reset()
tape = np.random.RandomState(0).randint(0, 18, size=(T, n))
for s in range(T):
    em.run(tape[s])
st = (C.c_long * 8)(); rs = (C.c_long * 16)()
lib().emu_stats(st); lib().emu_rstats(rs)
for e in range(n):
    em.render(e, 'ego')
lib().emu_stats(st); lib().emu_rstats(rs)
st, rs = np.array(st[:], float) / n, np.array(rs[:], float) / n
print('per env: tiles %d mixed %d (prims/mixed tile %.2f); pixels classified %d, queued %d' % (st[0], st[1], st[2] / max(st[1], 1), st[3], st[4]))
print('resolve calls %.0f; prims in mask %.2f per call' % (rs[0], rs[1] / rs[0]))
print('line prims visited %.2f per call, touched %.2f per call; touched samples %.2f per touched line, segs %.2f; exact samples %.2f per touched line; lower prims %.2f; generic %.3f' % (
    rs[2] / rs[0], rs[3] / rs[0], rs[4] / max(rs[3], 1), rs[5] / max(rs[3], 1), rs[6] / max(rs[3], 1), rs[7] / max(rs[3], 1), rs[8] / max(rs[3], 1)))
print('opaque coverage evaluations (top level) %.2f per call' % (rs[9] / rs[0]))
