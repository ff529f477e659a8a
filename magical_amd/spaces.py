"""Minimal stand-ins for the gym.spaces the reference exposes (`base_env.py:97-109`, `benchmarks/__init__.py:104-112`).

gym is not a dependency (and is not installed where this runs); the reference's users only read `.n`, `.shape`,
`.dtype`, `.low/.high`, `.contains()` and `.sample()` from these, so that is what is provided.  When gym /
gymnasium are importable `to_gym()` returns the real thing.
"""
import collections
import importlib

import numpy as np


class Discrete:
    def __init__(self, n):
        self.n, self.shape, self.dtype = int(n), (), np.dtype(np.int64)

    def contains(self, x):
        return float(x).is_integer() and 0 <= int(x) < self.n

    def sample(self, rng=np.random):
        return int(rng.randint(self.n))

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self.n

    def __repr__(self):
        return f'Discrete({self.n})'


class Box:
    def __init__(self, low, high, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self, rng=np.random):
        if self.dtype.kind in 'ui':
            return rng.randint(int(self.low.min()), int(self.high.max()) + 1, size=self.shape).astype(self.dtype)
        return rng.uniform(-1.0, 1.0, size=self.shape).astype(self.dtype)

    def __eq__(self, other):
        return isinstance(other, Box) and other.shape == self.shape and other.dtype == self.dtype \
            and np.array_equal(other.low, self.low) and np.array_equal(other.high, self.high)

    def __repr__(self):
        return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'


class Dict:
    def __init__(self, spaces):
        self.spaces = collections.OrderedDict(spaces)

    def contains(self, x):
        return set(x.keys()) == set(self.spaces.keys()) and all(s.contains(x[k]) for k, s in self.spaces.items())

    def sample(self, rng=np.random):
        return collections.OrderedDict((k, s.sample(rng)) for k, s in self.spaces.items())

    def __getitem__(self, k):
        return self.spaces[k]

    def __eq__(self, other):
        return isinstance(other, Dict) and other.spaces == self.spaces

    def __repr__(self):
        return 'Dict(' + ', '.join(f'{k}: {s!r}' for k, s in self.spaces.items()) + ')'


def to_gym(space):
    """The same space as a gymnasium / gym object, if either package is importable (else `space` itself)."""
    for modname in ('gymnasium', 'gym'):
        try:
            sp = importlib.import_module(modname + '.spaces')
        except Exception:
            continue
        if isinstance(space, Discrete):
            return sp.Discrete(space.n)
        if isinstance(space, Box):
            return sp.Box(low=space.low, high=space.high, dtype=space.dtype.type)
        return sp.Dict(collections.OrderedDict((k, to_gym(s)) for k, s in space.spaces.items()))
    return space
