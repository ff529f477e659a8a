"""Regenerates tests/golden/static_frames_48.npz from the reference's own data files (run in the build container only).

    python tests/golden/make_static_frames.py [/root/reference]

images/static-<task>-demo-v0.png are the reference's 192x192 allocentric renders of each Demo task's initial state
(its README assets): the only numeric record of the reference's output that exists without pymunk / pyglet.  They are
DATA, not code; to keep the fixture small and resolution-independent they are box-averaged 4x4 to 48x48x3 u8.
The fixture pins the oracle's (and through it the HIP rasteriser's) layout, palette and draw order -- weakly: edge
pixels differ by anti-aliasing / resampling, so tests compare with a tolerance (tests/test_oracle_render.py).
"""
import os
import sys

import numpy as np

TASKS = ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']


def main():
    from PIL import Image
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    out = {}
    for task in TASKS:
        img = np.asarray(Image.open(os.path.join(ref, 'images', f'static-{task.lower()}-demo-v0.png')).convert('RGB')).astype(np.float64)
        assert img.shape == (192, 192, 3), img.shape
        small = img.reshape(48, 4, 48, 4, 3).mean(axis=(1, 3))
        out[task] = np.rint(small).astype(np.uint8)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'static_frames_48.npz'), **out)
    print('wrote static_frames_48.npz:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
