"""ctypes loader for oracle/libmagical_ref.so (built by oracle/Makefile)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libmagical_ref.so')


def build(force=False):
    src = os.path.join(_HERE, 'magical_ref.c')
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'libmagical_ref.so'])
    return _SO


_lib = None
_D = C.c_double
_I = C.c_int
_P = C.c_void_p
_DP = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    sig = {
        'ref_new': (_P, []),
        'ref_free': (None, [_P]),
        'ref_clone': (_P, [_P]),
        'ref_perturb_warm': (None, [_P, _D, C.c_uint]),
        'ref_clear_warm': (None, [_P]),
        'ref_set_space': (None, [_P, _I, _D]),
        'ref_set_bg': (None, [_P, _D, _D, _D]),
        'ref_set_gjk_warm': (None, [_P, _I]),
        'ref_add_body': (_I, [_P, _I, _D, _D, _D, _D, _D]),
        'ref_body_set_pos': (None, [_P, _I, _D, _D]),
        'ref_add_circle': (_I, [_P, _I, _D, _D, _I, _I]),
        'ref_add_poly': (_I, [_P, _I, _I, _DP, _D, _D, _I, _I]),
        'ref_add_segment': (_I, [_P, _I, _D, _D, _D, _D, _D, _D]),
        'ref_add_pivot': (_I, [_P, _I, _I, _D, _D, _D, _D]),
        'ref_add_gear': (_I, [_P, _I, _I, _D, _D]),
        'ref_add_spring': (_I, [_P, _I, _I, _D, _D, _D]),
        'ref_add_pin': (_I, [_P, _I, _I, _D, _D, _D, _D]),
        'ref_add_limit': (_I, [_P, _I, _I, _D, _D]),
        'ref_add_motor': (_I, [_P, _I, _I, _D]),
        'ref_joint_params': (None, [_P, _I, _D, _D, _D]),
        'ref_set_robot': (None, [_P, _I, _I, _I, _I, _I, _I, _D, _D, _D]),
        'ref_add_geom': (_I, [_P, _I, _I, _DP, _D, _D, _D, _I, _I, _D, _D,
                              _I, _D, _D, _D, _I]),
        'ref_space_step': (None, [_P, _D]),
        'ref_set_action': (None, [_P, _I]),
        'ref_robot_update': (None, [_P]),
        'ref_substep': (None, [_P, _D]),
        'ref_step': (None, [_P, _I, _D]),
        'ref_nbodies': (_I, [_P]),
        'ref_nshapes': (_I, [_P]),
        'ref_njoints': (_I, [_P]),
        'ref_narbiters': (_I, [_P]),
        'ref_get_bodies': (None, [_P, _DP]),
        'ref_set_bodies': (None, [_P, _DP]),
        'ref_get_body_mass': (None, [_P, _DP]),
        'ref_get_joint_acc': (None, [_P, _DP]),
        'ref_get_contacts': (_I, [_P, _DP, _I]),
        'ref_collide_shapes': (_I, [_P, _I, _I, _DP]),
        'ref_shape_info': (None, [_P, _I, C.POINTER(C.c_int)]),
        'ref_shape_world': (_I, [_P, _I, _DP, _DP, C.POINTER(C.c_int)]),
        'ref_episode_steps': (_I, [_P]),
        'ref_render': (None, [_P, _I, _I, C.c_void_p]),
        'ref_area_downsample': (None, [C.c_void_p, _I, _I, _I, C.c_void_p]),
        'ref_set_unknowns': (None, [_I]),          # sensitivity study only (tools/oracle_unknowns.py)
        'ref_get_unknowns': (_I, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
