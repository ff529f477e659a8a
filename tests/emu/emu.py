"""ctypes wrapper for the host emulation of the kernel phases (tests only; never used by the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_SO = os.path.join(_HERE, 'libmgx_emu.so')
_SRCS = [os.path.join(_HERE, 'mgx_emu.cpp')] + [
    os.path.join(_ROOT, 'magical_amd', 'csrc', f) for f in ('mgx_world.cpp', 'mgx_world.h', 'mgx_sim.h', 'mgx_tmpl.h', 'mgx_raster.h')]

MODES = {'f32': 0, 'mixed': 1, 'f64': 2}


def build(defines=()):
    """The emulation library; `defines` (e.g. ('MGX_CONTACT_SPLIT=0',)) builds a variant of the phase code beside it."""
    so = _SO if not defines else _SO.replace('.so', '_' + '_'.join(d.replace('=', '') for d in defines) + '.so')
    newest = max(os.path.getmtime(p) for p in _SRCS)
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-ffp-contract=off', '-DMGX_RASTER_STATS', '-shared', '-o', so,
                               _SRCS[0], _SRCS[1]] + ['-D' + d for d in defines])
    return so


_libs = {}


def lib(defines=()):
    defines = tuple(defines)
    if defines not in _libs:
        L = C.CDLL(build(defines))
        L.emu_world_new.restype = C.c_void_p
        for name, args in {
            'emu_free': [C.c_void_p], 'emu_add_robot': [C.c_void_p] + [C.c_double] * 3,
            'emu_add_shape': [C.c_void_p, C.c_int, C.c_int] + [C.c_double] * 3,
            'emu_add_goal': [C.c_void_p] + [C.c_double] * 4 + [C.c_int],
            'emu_finalize': [C.c_void_p, C.c_int, C.c_int], 'emu_rows': [C.c_void_p, C.c_int],
            'emu_n_state': [C.c_void_p], 'emu_state_row': [C.c_void_p, C.c_int], 'emu_n_bodies': [C.c_void_p],
            'emu_contacts': [C.c_void_p, C.c_int, C.c_void_p, C.c_int],
            'emu_render': [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
            'emu_reset': [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
            'emu_run': [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                        C.c_int, C.c_void_p],
        }.items():
            getattr(L, name).argtypes = args
        _libs[defines] = L
    return _libs[defines]


class EmuBatch:
    """N envs of one world stepped by the emulated kernel phases."""

    def __init__(self, entities, max_steps, n_envs, mode='mixed', defines=()):
        L = lib(defines)
        self.L, self.mode, self.n = L, MODES[mode], n_envs
        self.h = L.emu_world_new()
        for ent in entities:
            if ent[0] == 'robot':
                L.emu_add_robot(self.h, *ent[1:])
            elif ent[0] == 'shape':
                L.emu_add_shape(self.h, *ent[1:])
            else:
                L.emu_add_goal(self.h, *ent[1:])
        rc = L.emu_finalize(self.h, max_steps, n_envs)
        assert rc == 0, rc
        pt = np.float32 if mode == 'f32' else np.float64
        rt = np.float64 if mode == 'f64' else np.float32
        self.sp = np.zeros((L.emu_rows(self.h, 0), n_envs), dtype=pt)
        self.sf = np.zeros((L.emu_rows(self.h, 1), n_envs), dtype=rt)
        self.si = np.zeros((L.emu_rows(self.h, 2), n_envs), dtype=np.int32)
        self.n_state = L.emu_n_state(self.h)
        self.rows = [L.emu_state_row(self.h, r) for r in range(self.n_state)]
        self.n_bodies = L.emu_n_bodies(self.h)

    def __del__(self):
        try:
            self.L.emu_free(self.h)
        except Exception:
            pass

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).ctypes.data
        self.L.emu_reset(self.h, self.mode, self.sp.ctypes.data, self.sf.ctypes.data, self.si.ctypes.data, m)

    def run(self, actions, n_sub=10, nl=16, count_step=True):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        done = np.zeros(self.n, dtype=np.uint8)
        self.L.emu_run(self.h, self.mode, self.sp.ctypes.data, self.sf.ctypes.data, self.si.ctypes.data,
                       a.ctypes.data, n_sub, nl, int(count_step), done.ctypes.data)
        return done

    def bodies(self):
        """[N, n_bodies, 9] with non-persistent components left at 0."""
        out = np.zeros((self.n, self.n_bodies, 9), dtype=np.float64)
        for m in self.rows:
            comp, b, row = m & 15, (m >> 4) & 0xFF, m >> 12
            out[:, b, comp] = (self.sp if comp < 3 else self.sf)[row]
        return out

    def set_bodies(self, arr):
        for m in self.rows:
            comp, b, row = m & 15, (m >> 4) & 0xFF, m >> 12
            (self.sp if comp < 3 else self.sf)[row] = arr[:, b, comp]

    def contacts(self, max_rows=128):
        buf = np.zeros((max_rows, 11), dtype=np.float64)
        n = self.L.emu_contacts(self.h, self.mode, buf.ctypes.data, max_rows)
        return buf[:n]

    def render(self, env=0, view='ego', native=False):
        res = 384 if native else 96
        out = np.zeros((res, res, 3), dtype=np.uint8)
        self.L.emu_render(self.h, self.mode, self.sp.ctypes.data, env, 0 if view == 'ego' else 1, int(native), out.ctypes.data)
        return out
