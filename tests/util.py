"""Shared helpers for the parity tests: oracle <-> product body index mapping."""
import numpy as np

from oracle.entities_ref import Robot as RefRobot, Shape as RefShape
from oracle.env_ref import RefEnv

TASKS = ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']


def ref_body_index(ref_env):
    """Oracle body indices of the non-static bodies, in creation order == product bodies 1..n."""
    idx = []
    for e in ref_env.world.entities:
        if isinstance(e, (RefRobot, RefShape)):
            idx += list(e.bodies)
    return idx


def comparable_mask(ref_env):
    """bool[n_dyn, 3] of pose components that are persistent in the product (control body and eye positions are not)."""
    idx = ref_body_index(ref_env)
    mask = np.ones((len(idx), 3), dtype=bool)
    rb = ref_env.task.robot.bodies          # robot, control, eye, eye, finger, finger
    mask[idx.index(rb[1])] = False
    for eye in rb[2:4]:
        mask[idx.index(eye), :2] = False
    return mask


def ref_entities_as_tuples(ref_env):
    """Entity list of an oracle env in the emulator's tuple format."""
    from oracle.entities_ref import GoalRegion
    st = {'triangle': 0, 'square': 1, 'pentagon': 2, 'hexagon': 3, 'octagon': 4, 'circle': 5, 'star': 6}
    co = {'red': 0, 'green': 1, 'blue': 2, 'yellow': 3}
    out = []
    for e in ref_env.world.entities:
        if isinstance(e, RefRobot):
            out.append(('robot', e.init_pos[0], e.init_pos[1], e.init_angle))
        elif isinstance(e, RefShape):
            out.append(('shape', st[e.shape_type], co[e.colour_name], e.init_pos[0], e.init_pos[1], e.init_angle))
        elif isinstance(e, GoalRegion):
            out.append(('goal', e.x, e.y, e.h, e.w, co[e.colour_name]))
    return out


def new_ref(task):
    e = RefEnv(task)
    e.reset()
    return e
