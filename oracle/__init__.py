"""oracle/ -- TEST INFRASTRUCTURE (CPU restatement of the reference hot path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (magical_amd/) never does.

PARITY UNPINNED: the arithmetic of the reference path lives in pymunk 5.6 /
Chipmunk2D 7.0.x, pyglet/OpenGL and cv2, none of which can be installed in the
build container, and the reference's tests pin no numbers.  See DESIGN.md.
"""
