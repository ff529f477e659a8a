"""oracle/ -- TEST INFRASTRUCTURE (CPU restatement of the reference hot path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (magical_amd/) never does.

PARITY UNPINNED for the physics, the painter and the resize: that arithmetic lives
in pymunk 5.6 / Chipmunk2D 7.0.x, pyglet/OpenGL and cv2, none of which can be
installed in the build container, and the reference's tests pin no numbers.

PINNED on outputs of the reference's own code (tests/golden/reference_vectors.json,
generated in the build container by tests/golden/make_reference_vectors.py, checked
by tests/test_reference_vectors.py): style_ref.py (palette), entities_ref.PhysVars
(defaults, bounds, sample() draws), geom_ref.py's regular-polygon formulas,
tasks_ref.randomise_hw / longest_line / episode lengths / random-generation arrays,
and every task's score_on_end_of_traj() (the region tasks: on every overlap set).
See DESIGN.md section 6.
"""
