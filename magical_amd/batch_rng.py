"""The per-episode draws of many envs at once.

Every env of a batch owns a `np.random.RandomState` (env k is seeded with `seed + k`, base_env.BaseEnv.seed) and the
reference's `on_reset()` draws from it call by call -- `rng.randint`, `rng.choice`, `rng.shuffle`, `rng.uniform`
(e.g. cluster.py:81-110, match_regions.py:51-117, find_dupe.py:84-112, fix_colour.py:78-113, make_line.py:100-110,
base_env.py:198-203).  Doing that in a Python loop over 4096 envs costs ~20 us per env and reset; `BatchRng` makes the same
draws for ALL envs of a reset per call, natively (include/mgx.h: mgx_rng_*_batch), directly on the live MT19937 states, so
every stream advances exactly as numpy would advance it (tests/test_host_api.py checks the primitives against numpy itself;
the GPU suite checks whole resets against the per-env loop).  The streams are independent, so issuing one kind of draw for
all envs before the next kind keeps each env's own order of draws, which is all the reference's order means.
"""
import ctypes as C

import numpy as np

from . import _native as nat

_U64P, _IP, _DP = C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_double)


def state_addresses(rngs):
    """uint64[m]: where each RandomState's MT19937 state lives (the native batch calls advance it in place)."""
    out = np.empty(len(rngs), dtype=np.uint64)
    for i, rng in enumerate(rngs):
        bg = rng._bit_generator
        assert type(bg).__name__ == 'MT19937'
        out[i] = bg.ctypes.state_address
    return out


class BatchRng:
    def __init__(self, rngs, lib=None, addrs=None):
        self.L = lib or nat.lib()
        self.m = len(rngs)
        # (addrs: state_addresses(rngs) if the caller has them already -- a pass over 4096 generators takes 1 ms)
        self.addrs = state_addresses(rngs) if addrs is None else np.ascontiguousarray(addrs, dtype=np.uint64)
        assert len(self.addrs) == self.m
        # one stream per env: the native calls spread large batches over host threads (include/mgx.h)
        assert len(np.unique(self.addrs)) == self.m, 'the same RandomState twice in one batch'
        self._rngs = rngs          # keep the generators (and so the states the addresses point at) alive

    def _sel(self, rows):
        if rows is None:
            return self.addrs, self.m
        a = np.ascontiguousarray(self.addrs[rows])
        return a, len(a)

    def randint(self, n, count=1, counts=None, rows=None):
        """count (or counts[k]) draws of rng.randint(0, n) per env -> int32[m', stride]; entries past an env's count are 0.
        rows: only these envs (index array / bool mask into the batch) draw."""
        addrs, m = self._sel(rows)
        stride = int(count if counts is None else (max(int(np.max(counts)), 1) if len(counts) else 1))
        out = np.zeros((m, max(stride, 1)), dtype=np.int32)
        if m == 0:
            return out
        cp = None if counts is None else np.ascontiguousarray(counts, dtype=np.int32)
        nat.check(self.L.mgx_rng_bounded_batch(m, addrs.ctypes.data_as(_U64P), None if cp is None else cp.ctypes.data_as(_IP), int(count if counts is None else 0),
                                               int(n) - 1, out.ctypes.data_as(_IP), out.shape[1]))
        return out

    def random_sample(self, count=1, counts=None, rows=None):
        """count (or counts[k]) draws of rng.random_sample() per env -> float64[m', stride]."""
        addrs, m = self._sel(rows)
        stride = int(count if counts is None else (max(int(np.max(counts)), 1) if len(counts) else 1))
        out = np.zeros((m, max(stride, 1)), dtype=np.float64)
        if m == 0:
            return out
        cp = None if counts is None else np.ascontiguousarray(counts, dtype=np.int32)
        nat.check(self.L.mgx_rng_doubles_batch(m, addrs.ctypes.data_as(_U64P), None if cp is None else cp.ctypes.data_as(_IP), int(count if counts is None else 0),
                                               out.ctypes.data_as(_DP), out.shape[1]))
        return out

    def shuffle(self, n_items, rows=None):
        """The permutation rng.shuffle() applies to a list of n_items[k] items: shuffled[i] = original[perm[k, i]] -> int32[m', max n]."""
        addrs, m = self._sel(rows)
        n = np.ascontiguousarray(np.broadcast_to(np.asarray(n_items, dtype=np.int32), (m,)))
        perm = np.zeros((m, max(int(n.max()) if m else 1, 1)), dtype=np.int32)
        if m:
            nat.check(self.L.mgx_rng_shuffle_batch(m, addrs.ctypes.data_as(_U64P), n.ctypes.data_as(_IP), perm.ctypes.data_as(_IP), perm.shape[1]))
        return perm


def uniform_hw(u, min_side, max_side, current_hw=None, linf_bound=None):
    """geom.randomise_hw (geom.py:344-360) on drawn samples u[m, 2] -> (h[m], w[m]); the same two multiplies and adds."""
    lo_h = lo_w = float(min_side)
    hi_h = hi_w = float(max_side)
    if linf_bound is not None:
        lo_h, hi_h = max(lo_h, current_hw[0] - linf_bound), min(hi_h, current_hw[0] + linf_bound)
        lo_w, hi_w = max(lo_w, current_hw[1] - linf_bound), min(hi_w, current_hw[1] + linf_bound)
    return lo_h + (hi_h - lo_h) * u[:, 0], lo_w + (hi_w - lo_w) * u[:, 1]
