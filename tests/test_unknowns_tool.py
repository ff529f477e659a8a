"""tools/oracle_unknowns.py, part A2 (round 6): episodes that start near a task's scoring threshold and run a scripted pusher on the ORACLE
must score in a good share of the cases -- the study's point is that its baseline score distribution is not identically zero -- and the
block placement must leave no two blocks interpenetrating.  CPU only (oracle/ is the checker here: test infrastructure)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize('task', ['MoveToCorner', 'FindDupe', 'ClusterColour'])
def test_scoring_start_episodes_score(task):
    from tools import oracle_unknowns as ou
    _, flags, lo, out = ou._episodes_scoring((task, 0, 0, 24))
    assert out.shape == (24, 2) and np.all((out[:, 0] >= 0) & (out[:, 0] <= 1))
    assert np.mean(out[:, 0] > 0) >= 0.2, (task, out[:, 0])
    assert out[:, 1].max() < 1.5          # blocks are shoved, not shot across the arena (no interpenetrating start)


def test_scoring_start_is_deterministic_and_paired_with_the_control():
    from tools import oracle_unknowns as ou
    a = ou._episodes_scoring(('MatchRegions', 0, 3, 9))[3]
    b = ou._episodes_scoring(('MatchRegions', 0, 3, 9))[3]
    c = ou._episodes_scoring(('MatchRegions', -1, 3, 9))[3]          # the control: initial poses +-1e-9
    assert np.array_equal(a, b)
    assert c.shape == a.shape and np.abs(c[:, 1] - a[:, 1]).max() < 0.5
