"""Pins for the oracle rasteriser + LoRes4E preprocessor restatement."""
import os

import numpy as np
import pytest

from oracle.env_ref import LoRes4ERef, RefEnv, area_downsample
from oracle.style_ref import COLOURS_RGB, darken_rgb, lighten_rgb, to_u8

IMG_DIR = '/root/reference/images'


def test_palette_matches_survey_appendix_d():
    # SURVEY.md Appendix D (u8 palette: base / darkened / lightened x2)
    expect = {
        'blue': ((135, 185, 211), (110, 170, 202), (194, 219, 233)),
        'yellow': ((254, 213, 123), (254, 201, 86), (254, 234, 188)),
        'red': ((245, 129, 165), (243, 94, 141), (250, 190, 209)),
        'green': ((195, 208, 130), (183, 198, 105), (224, 231, 191)),
        'grey': ((162, 163, 175), (144, 145, 159), (208, 208, 214)),
    }
    for name, (base, dark, light2) in expect.items():
        c = COLOURS_RGB[name]
        assert to_u8(c) == base
        assert to_u8(darken_rgb(c)) == dark
        assert to_u8(lighten_rgb(c, 2)) == light2
    assert to_u8(lighten_rgb(COLOURS_RGB['grey'], 4)) == (231, 231, 234)


def test_area_downsample_round_half_even():
    img = np.zeros((4, 4, 1), dtype=np.uint8)
    img[0, 0, 0] = 8          # mean 0.5 -> 0 (ties to even)
    assert area_downsample(img, 4)[0, 0, 0] == 0
    img[0, 1, 0] = 16         # mean 1.5 -> 2
    assert area_downsample(img, 4)[0, 0, 0] == 2
    img[0, 2, 0] = 1          # 25/16 = 1.5625 -> 2
    assert area_downsample(img, 4)[0, 0, 0] == 2
    rng = np.random.RandomState(0)
    big = rng.randint(0, 256, size=(384, 384, 12)).astype(np.uint8)
    out = area_downsample(big, 4)
    ref = np.rint(big.reshape(96, 4, 96, 4, 12).astype(np.float64).mean(axis=(1, 3)))
    assert np.array_equal(out, ref.astype(np.uint8))


def test_ego_frame_geometry():
    """Robot centre lands at (0.5 W, 0.15 H) from the bottom-left and faces up
    (base_env.py:294-301)."""
    e = RefEnv('MoveToCorner')
    e.reset()
    ego = e.render('ego')
    assert ego.shape == (384, 384, 3)
    # pixel just below the robot centre: inside the r=0.19 grey disc (the eyes sit above)
    row = 383 - int(0.15 * 384) + 8
    assert tuple(ego[row, 192]) == (162, 163, 175)
    # 0.195 units to the left/right of centre: the dark ring (0.19 < r < 0.2)
    off = int(round(0.195 * 384 / 2.04))
    assert tuple(ego[383 - int(0.15 * 384), 192 - off]) == (144, 145, 159)
    # far corner outside the arena: background
    e2 = RefEnv('MoveToRegion')
    e2.reset()
    assert tuple(e2.render('allo')[1, 1]) == (231, 231, 234)


def test_lores4e_stack_semantics():
    """FlattenFrameStack: reset fills 4 copies; step appends newest last
    (benchmarks/__init__.py:124-136)."""
    env = LoRes4ERef(RefEnv('MoveToCorner'))
    obs0 = env.reset()
    assert obs0.shape == (96, 96, 12) and obs0.dtype == np.uint8
    for k in range(1, 4):
        assert np.array_equal(obs0[..., :3], obs0[..., 3 * k:3 * k + 3])
    obs1, _, _, _ = env.step(4)
    obs2, _, _, _ = env.step(4)
    assert np.array_equal(obs1[..., :9], obs0[..., 3:])
    assert np.array_equal(obs2[..., :9], obs1[..., 3:])
    assert not np.array_equal(obs2[..., 9:], obs2[..., 6:9])
    # resize-after-stack == stack-after-resize (per-channel box filter)
    assert np.array_equal(obs2[..., 9:], env.env.render_lores('ego'))


@pytest.mark.skipif(not os.path.isdir(IMG_DIR), reason='reference images only exist in the build container')
@pytest.mark.parametrize('task', ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine',
                                  'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape'])
def test_allo_reset_frame_vs_reference_png(task):
    """Weak pin: images/static-<task>-demo-v0.png are 192x192
    allocentric renders of each Demo initial state."""
    from PIL import Image
    e = RefEnv(task)
    e.reset()
    half = area_downsample(e.render('allo'), 2).astype(int)
    ref = np.asarray(Image.open(os.path.join(IMG_DIR, f'static-{task.lower()}-demo-v0.png')).convert('RGB')).astype(int)
    d = np.abs(half - ref).max(axis=2)
    assert d.mean() < 1.0                  # same layout, palette and draw order
    assert (d > 24).mean() < 0.02          # only edge pixels differ (AA / resampling differences)


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('task', ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine',
                                  'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape'])
def test_allo_reset_frame_vs_golden_reference_frames(task):
    """The same weak pin as above, from the committed fixture (tests/golden/static_frames_48.npz = the reference's PNGs
    box-averaged to 48x48 by tests/golden/make_static_frames.py): runs wherever the repo is, no /root/reference needed."""
    gold = np.load(os.path.join(GOLDEN, 'static_frames_48.npz'))[task].astype(int)
    e = RefEnv(task)
    e.reset()
    small = area_downsample(e.render('allo'), 8).astype(int)
    d = np.abs(small - gold).max(axis=2)
    assert d.mean() < 1.5                  # same layout, palette and draw order
    assert (d > 24).mean() < 0.03          # only edge pixels differ (AA / resampling differences)


def test_oracle_reproduces_golden_vectors():
    """tests/golden/oracle_vectors.json (made by make_oracle_vectors.py): the oracle still renders byte-identical reset
    observations and, after the recorded tapes, reaches the recorded poses (to 1e-9: a different libm may move the last
    bit of a sin / cos, which six env-steps can amplify) and -- when the poses agree exactly -- the recorded frames."""
    import hashlib
    import json
    from oracle.env_ref import LoRes4ERef
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    gold = json.load(open(os.path.join(GOLDEN, 'oracle_vectors.json')))
    for task, rec in gold.items():
        env = LoRes4ERef(RefEnv(task))
        obs = env.reset()
        assert sha(obs) == rec['reset']['lores4e'] and sha(env.env.render_lores('ego')) == rec['reset']['ego']
        assert sha(env.env.render_lores('allo')) == rec['reset']['allo']
        assert np.array_equal(env.env.bodies()[:, :3], np.asarray(rec['reset']['bodies']))
        for a in rec['tape']:
            obs, _, _, _ = env.step(a)
        want = np.asarray(rec['final']['bodies'])
        assert np.abs(env.env.bodies() - want).max() < 1e-9, task
        if np.array_equal(env.env.bodies(), want):
            assert sha(obs) == rec['final']['lores4e'] and sha(env.env.render_lores('ego')) == rec['final']['ego']


def test_oracle_reproduces_variant_vectors():
    """tests/golden/variant_vectors.json (made by make_variant_vectors.py): seeded alike, the oracle's restatement of the
    reference's on_reset draws (counts, shape types, colours, layout; two consecutive resets of one stream) still yields
    the recorded worlds, first observations and scores."""
    import json
    import subprocess
    import sys
    import tempfile
    gold_path = os.path.join(GOLDEN, 'variant_vectors.json')
    gold = json.load(open(gold_path))
    sys.path.insert(0, GOLDEN)
    import make_variant_vectors as mk
    assert {f'{t}-{v}' for t, v, _ in mk.CASES} == set(gold)
    for task, variant, flags in mk.CASES:
        rec = gold[f'{task}-{variant}']
        assert rec['flags'] == flags and rec['seed'] == mk.SEED
        now = mk.record(task, flags, mk.SEED)
        for ep, (a, b) in enumerate(zip(now, rec['episodes'])):
            assert a['enabled'] == b['enabled'] and a['shape_types'] == b['shape_types'] and a['tape'] == b['tape'], (task, ep)
            for pa, pb in zip(a['poses'], b['poses']):
                assert (pa is None) == (pb is None) and (pa is None or np.abs(np.asarray(pa) - np.asarray(pb)).max() < 1e-12), (task, ep)
            if a['poses'] == b['poses']:
                assert a['lores4e'] == b['lores4e'], (task, ep)
            assert abs(a['score'] - b['score']) < 1e-9, (task, ep)
