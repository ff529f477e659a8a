"""Entity descriptors + action table: the host-side mirror of magical/entities.py.

In the reference an Entity builds pymunk bodies/shapes/constraints and pyglet geoms in
`setup()`.  Here an entity is a plain description; `World.add_entities()` hands it to the
native library (mgx_world_add_*), which builds the batched physics + draw template.
Names and argument meaning follow the reference (entities.py:148-190, 217-236, 545-612, 769-788).
"""
import enum
import math


class RobotAction(enum.IntFlag):  # entities.py:148-155
    NONE = 0
    UP = 1
    DOWN = 2
    LEFT = 4
    RIGHT = 8
    OPEN = 16
    CLOSE = 32


def _build_action_table():
    """entities.py:162-182: id = 9*[close] + 3*lr + ud, ud in (none, UP, DOWN), lr in (none, LEFT, RIGHT)."""
    names_ud = {RobotAction.NONE: '', RobotAction.UP: 'Up', RobotAction.DOWN: 'Down'}
    names_lr = {RobotAction.NONE: '', RobotAction.LEFT: 'Left', RobotAction.RIGHT: 'Right'}
    names_grip = {RobotAction.OPEN: 'Open', RobotAction.CLOSE: 'Close'}
    table = []
    for grip in (RobotAction.OPEN, RobotAction.CLOSE):
        for lr in (RobotAction.NONE, RobotAction.LEFT, RobotAction.RIGHT):
            for ud in (RobotAction.NONE, RobotAction.UP, RobotAction.DOWN):
                table.append((len(table), (ud, lr, grip), names_ud[ud] + names_lr[lr] + names_grip[grip]))
    return tuple(table)


ACTION_NUMS_FLAGS_NAMES = _build_action_table()
ACTION_ID_TO_FLAGS = {act_id: flags for act_id, flags, _ in ACTION_NUMS_FLAGS_NAMES}
FLAGS_TO_ACTION_ID = {flags: act_id for act_id, flags, _ in ACTION_NUMS_FLAGS_NAMES}


class ShapeType(str, enum.Enum):  # entities.py:545-554
    TRIANGLE = 'triangle'
    SQUARE = 'square'
    PENTAGON = 'pentagon'
    HEXAGON = 'hexagon'
    OCTAGON = 'octagon'
    CIRCLE = 'circle'
    STAR = 'star'


class ShapeColour(str, enum.Enum):  # entities.py:557-561
    RED = 'red'
    GREEN = 'green'
    BLUE = 'blue'
    YELLOW = 'yellow'


SHAPE_TYPES = (ShapeType.SQUARE, ShapeType.PENTAGON, ShapeType.STAR, ShapeType.CIRCLE)       # entities.py:568-574
SHAPE_COLOURS = (ShapeColour.RED, ShapeColour.GREEN, ShapeColour.BLUE, ShapeColour.YELLOW)   # entities.py:575-581

# native enum values (include/mgx.h)
SHAPE_TYPE_ID = {t: i for i, t in enumerate(ShapeType)}
COLOUR_ID = {c: i for i, c in enumerate(ShapeColour)}
# plain-str view of SHAPE_COLOURS for rng.choice: numpy stringifies str-Enum members by their repr on this Python /
# numpy pair, the reference's environment by their value; the draws (one randint per element) are the same either way
SHAPE_COLOUR_NAMES = tuple(c.value for c in SHAPE_COLOURS)
SHAPE_COLOURS_OBJ = None     # np.asarray(SHAPE_COLOURS, dtype='object'), built on first use (rng.choice argument of two tasks)


def shape_colours_obj():
    global SHAPE_COLOURS_OBJ
    if SHAPE_COLOURS_OBJ is None:
        import numpy as np
        SHAPE_COLOURS_OBJ = np.asarray(SHAPE_COLOURS, dtype='object')
    return SHAPE_COLOURS_OBJ


SHAPE_TYPES_OBJ = None
SHAPE_TYPE_NAMES = tuple(t.value for t in SHAPE_TYPES)


def shape_types_obj():
    """np.asarray(SHAPE_TYPES, dtype='object'): rng.choice argument of the Test*Shape branches."""
    global SHAPE_TYPES_OBJ
    if SHAPE_TYPES_OBJ is None:
        import numpy as np
        SHAPE_TYPES_OBJ = np.asarray(SHAPE_TYPES, dtype='object')
    return SHAPE_TYPES_OBJ


def colour_id_of_draw():
    """native colour id of draw index i into SHAPE_COLOUR_NAMES (batched draws)."""
    import numpy as np
    return np.array([COLOUR_ID[c] for c in SHAPE_COLOURS], dtype=np.int64)


def type_id_of_draw():
    """native shape-type id of draw index i into SHAPE_TYPE_NAMES (batched draws)."""
    import numpy as np
    return np.array([SHAPE_TYPE_ID[t] for t in SHAPE_TYPES], dtype=np.int32)


def draw_choice(rng, seq, size=None):
    """rng.choice(seq[, size=n]) of the reference's per-episode draws, without numpy's argument handling: choice() is
    randint(0, len(seq)[, size]) followed by indexing, so the stream advances identically (tests/test_host_api.py);
    this is several times cheaper, and the draws run once per env and reset."""
    if size is None:
        return seq[rng.randint(0, len(seq))]
    return [seq[i] for i in rng.randint(0, len(seq), size=size)]


class Entity:
    ent_id = None      # index in the native world after add_entities()
    body = None        # native body index of the main body (None for goal regions)


class Robot(Entity):
    """entities.py:217-236."""

    def __init__(self, radius, init_pos, init_angle, mass=1.0):
        self.radius, self.init_pos, self.init_angle, self.mass = radius, tuple(init_pos), float(init_angle), mass
        self.finger_rot_limit_outer = math.pi / 8
        self.finger_rot_limit_inner = 0.0


class Shape(Entity):
    """entities.py:584-612."""

    def __init__(self, shape_type, colour_name, shape_size, init_pos, init_angle, mass=0.5):
        self.shape_type = ShapeType(shape_type)
        self.colour_name = ShapeColour(colour_name)
        self.shape_size, self.init_pos, self.init_angle, self.mass = shape_size, tuple(init_pos), float(init_angle), mass


class GoalRegion(Entity):
    """entities.py:769-788: (x, y) is the top-left corner."""

    def __init__(self, x, y, h, w, colour_name):
        assert h > 0 and w > 0
        self.x, self.y, self.h, self.w = float(x), float(y), float(h), float(w)
        self.colour_name = ShapeColour(colour_name)
