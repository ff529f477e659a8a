"""ctypes binding of libmagical_hip.so (the C ABI in include/mgx.h).

The product path has NO fallback: if the HIP library is missing or no GPU is
visible, loading / engine creation raises.  Build with `python __graft_entry__.py`
(or `magical_amd._native.build()`), which runs hipcc for gfx950.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, 'csrc')
LIB_PATH = os.environ.get('MGX_LIB_PATH') or os.path.join(_PKG, 'libmagical_hip.so')      # (override: development builds with other -D knobs)
_SOURCES = ['mgx_api.hip', 'mgx_world.cpp']
_DEPS = _SOURCES + ['mgx_step.hip', 'mgx_raster.hip', 'mgx_raster_body.inc', 'mgx_score.hip', 'mgx_sim.h', 'mgx_raster.h', 'mgx_tmpl.h', 'mgx_world.h']

# enums (include/mgx.h)
MGX_F32, MGX_F64, MGX_F32_PURE = 0, 1, 2
VIEW_EGO, VIEW_ALLO = 0, 1
SCORE_CORNER, SCORE_LINE, SCORE_CLUSTER = 1, 2, 3
OBS_FRAME, OBS_STACK4, OBS_STACK3_HI, OBS_SLOT_LO, OBS_PLANAR = 0, 1, 2, 3, 4
INFO = {k: i for i, k in enumerate([
    'n_bodies', 'n_shapes', 'n_joints', 'n_pairs', 'n_prims', 'state_rows_p', 'state_rows_f', 'state_rows_i',
    'robot_body', 'n_entities', 'cache_slots', 'max_contacts', 'max_episode_steps', 'n_jacc', 'physvar_row'])}


class MgxError(RuntimeError):
    pass


class MgxCapacityWarning(RuntimeWarning):
    """An env's contacts / overlapping pairs did not fit its fixed-size working set (BaseEnv._check_capacity)."""


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(_CSRC, f)) > t for f in _DEPS)


def build(force=False, verbose=False, defines=(), out=None, extra=()):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  defines / out / extra: development builds with other -D knobs / compiler flags."""
    if not force and not defines and not needs_build():
        return LIB_PATH
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-parameter', '-Wno-extern-c-compat',
           '-pthread', '-o', out or LIB_PATH] + list(extra) + [f'-D{d}' for d in defines] + [os.path.join(_CSRC, f) for f in _SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
    return out or LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own ROCm runtime; it must be in the process before ours resolves libamdhip64, or the
    # two HIP runtimes fight over the device ("no HIP device visible")
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise MgxError(f'{LIB_PATH} not found: the HIP extension is not built (run `python __graft_entry__.py`); '
                       'magical_amd has no CPU fallback')
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.mgx_last_error.restype = C.c_char_p
    sig = {
        'mgx_version': [],
        'mgx_world_create': [C.POINTER(vp)],
        'mgx_world_set_phys_vars': [vp, dp],
        'mgx_world_add_robot': [vp, dbl, dbl, dbl],
        'mgx_world_add_shape': [vp, i32, i32, dbl, dbl, dbl],
        'mgx_world_add_goal': [vp, dbl, dbl, dbl, dbl, i32],
        'mgx_world_finalize': [vp, i32],
        'mgx_world_info': [vp, i32, ip],
        'mgx_world_entity': [vp, i32, ip, ip, ip, ip],
        'mgx_world_body_table': [vp, dp, dp],
        'mgx_world_n_state_entries': [vp],
        'mgx_world_state_entry': [vp, i32, ip, ip, ip],
        'mgx_world_goal_bb': [vp, i32, dp],
        'mgx_world_entity_shapes': [vp, i32, i32, ip, dp, ip, dp, i32],
        'mgx_world_prim_table': [vp, ip, ip, ip],
        'mgx_world_palette': [i32, i32],
        'mgx_rng_bounded_batch': [i32, C.POINTER(C.c_uint64), ip, i32, i32, ip, i32],
        'mgx_rng_doubles_batch': [i32, C.POINTER(C.c_uint64), ip, i32, dp, i32],
        'mgx_rng_shuffle_batch': [i32, C.POINTER(C.c_uint64), ip, ip, i32],
        'mgx_world_randomise_all_poses_batch': [vp, i32, dp, ip, i32, C.POINTER(C.c_uint8), dp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                                                dp, dp, i32, C.POINTER(C.c_uint64), dp],
        'mgx_world_placement_collides': [vp, i32, dp, C.POINTER(C.c_uint8), dp],
        'mgx_world_randomise_all_poses': [vp, dp, ip, i32, C.POINTER(C.c_uint8), dp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), dp, dp,
                                          C.POINTER(C.c_uint32), ip, dp],
        'mgx_engine_set_entity_colours': [vp, vp],
        'mgx_world_variant': [vp, C.POINTER(C.c_uint8), ip, C.POINTER(vp)],
        'mgx_engine_enable_env_worlds': [vp, vp],
        'mgx_engine_set_env_variants': [vp, i32, ip, C.POINTER(C.c_uint8), ip, vp],
        'mgx_engine_env_randomise_all_poses_batch': [vp, i32, ip, dp, ip, i32, C.POINTER(C.c_uint8), dp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                                                     dp, dp, i32, C.POINTER(C.c_uint64), dp],
        'mgx_engine_env_world_info': [vp, i32, i32, ip],
        'mgx_engine_set_goal_rects': [vp, vp],
        'mgx_engine_score_overlaps': [vp, vp, vp, vp, vp],
        'mgx_engine_n_goals': [vp],
        'mgx_engine_score_points': [vp, vp, i32, i32, ip, ip, i32, vp, dp, i32, i32, vp, vp, vp],
        'mgx_engine_create': [vp, i32, i32, i32, i32, C.POINTER(vp)],
        'mgx_engine_state_shape': [vp, ip, ip, ip, ip, ip],
        'mgx_engine_lanes_per_env': [vp],
        'mgx_engine_lds_bytes': [vp, i32],
        'mgx_engine_reset': [vp, vp, vp, vp, vp, vp],
        'mgx_engine_reset_poses': [vp, vp, vp, vp, vp, vp, vp],
        'mgx_engine_step': [vp, vp, vp, vp, vp, vp, vp],
        'mgx_engine_substeps': [vp, vp, vp, vp, vp, i32, vp],
        'mgx_engine_render': [vp, vp, vp, i64, i32, i32, vp, vp],
        'mgx_engine_step_render': [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp],
        'mgx_engine_handoff_stats': [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)],
        'mgx_engine_debug_handoff_peek': [vp, C.POINTER(C.c_uint)],
        'mgx_engine_render_native': [vp, vp, i32, vp, i32, vp],
        'mgx_engine_set_timing': [vp, i32],
        'mgx_engine_timing_read': [vp, i32, C.POINTER(C.c_float), i32],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.mgx_world_destroy.argtypes = [vp]
    L.mgx_world_destroy.restype = None
    L.mgx_engine_destroy.argtypes = [vp]
    L.mgx_engine_destroy.restype = None
    _lib = L
    return L


def check(rc):
    if rc < 0:
        raise MgxError(f'mgx error {rc}: {lib().mgx_last_error().decode()}')
    return rc


EXPORTED_SYMBOLS = [
    'mgx_last_error', 'mgx_version', 'mgx_world_create', 'mgx_world_destroy', 'mgx_world_set_phys_vars',
    'mgx_world_add_robot', 'mgx_world_add_shape', 'mgx_world_add_goal', 'mgx_world_finalize', 'mgx_world_info',
    'mgx_world_entity', 'mgx_world_body_table', 'mgx_world_n_state_entries', 'mgx_world_state_entry',
    'mgx_world_goal_bb', 'mgx_world_entity_shapes', 'mgx_world_prim_table', 'mgx_world_palette', 'mgx_rng_bounded_batch', 'mgx_rng_doubles_batch', 'mgx_rng_shuffle_batch', 'mgx_world_placement_collides', 'mgx_world_randomise_all_poses', 'mgx_world_randomise_all_poses_batch',
    'mgx_engine_create', 'mgx_engine_destroy', 'mgx_engine_set_entity_colours', 'mgx_engine_set_goal_rects', 'mgx_engine_score_overlaps', 'mgx_engine_n_goals', 'mgx_engine_score_points',
    'mgx_world_variant', 'mgx_engine_enable_env_worlds', 'mgx_engine_set_env_variants', 'mgx_engine_env_randomise_all_poses_batch', 'mgx_engine_env_world_info',
    'mgx_engine_state_shape', 'mgx_engine_lanes_per_env', 'mgx_engine_lds_bytes', 'mgx_engine_reset', 'mgx_engine_reset_poses',
    'mgx_engine_step', 'mgx_engine_substeps', 'mgx_engine_render', 'mgx_engine_step_render', 'mgx_engine_handoff_stats', 'mgx_engine_render_native',
    'mgx_engine_set_timing', 'mgx_engine_timing_read',
]
