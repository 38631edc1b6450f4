"""ctypes binding of csrc/libmpb_hip.so (C-ABI declared in include/mpb.h).

The HIP extension is the product: there is no CPU or PyTorch fallback.  If the shared library is
missing or does not export a symbol the header declares, importing this module's ``lib()`` raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MPB_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libmpb_hip.so')  # env: tuning builds

_f = ctypes.c_float
_i = ctypes.c_int
_p = ctypes.c_void_p
_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32

# name -> argtypes ; mirrors include/mpb.h one to one
SIGNATURES = {
    'mpb_version': [],
    'mpb_last_error': [],
    'mpb_geom_check': [_p, _i],
    'mpb_geom_flags': [_p, _i, _p],
    'mpb_cost_collision_eval': [_p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    'mpb_cost_collision_grad': [_p, _p, _i, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    'mpb_cost_terms_eval': [_p] * 7 + [_i] * 5 + [_u32] + [_f] * 7 + [_i, _i, _p],
    'mpb_cost_terms_grad': [_p] * 8 + [_i] * 5 + [_u32] + [_f] * 11 + [_i, _p],
    'mpb_traj_resample': [_p, _p, _p, _i, _i, _i, _i, _f, _p],
    'mpb_gp_factor_error': [_p, _p, _i, _i, _i, _f, _p],
    'mpb_traj_interpolate': [_p, _p, _i, _i, _i, _i, _p],
    'mpb_traj_finite_difference': [_p, _p, _i, _i, _i, _f, _p],
    'mpb_fk_collision_points': [_p, _p, _p, _i, _i, _i, _p],
    'mpb_fk_collision_points_vjp': [_p, _p, _p, _p, _i, _i, _i, _p],
    'mpb_field_cost_points': [_p, _p, _p, _i, _i, _p],
    'mpb_field_cost_points_vjp': [_p, _p, _p, _p, _i, _i, _p],
    'mpb_stomp_step': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _u64, _u32, _u32, _p],
    'mpb_stomp_workspace_bytes': [_i, _i, _i, _i],
    'mpb_stomp_run': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, ctypes.c_size_t, _i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _u64, _u32, _u32, _p],
    'mpb_stomp_run_checked': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, ctypes.c_size_t, _i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _u64, _u32, _u32, _p, _p, _p, _p],
    'mpb_stomp_run_status': [_p, _p, _p],
    'mpb_stomp_run_timed': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, ctypes.c_size_t, _i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _u64, _u32, _u32, _p, _p, _p, _p, _p],
    'mpb_stomp_workspace_init': [_p, ctypes.c_size_t, _p],
    'mpb_stomp_run_path': [_i, ctypes.c_size_t, _i, _i, _i, _i],
    'mpb_stomp_step_profile': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _u64, _u32, _u32, _p, _p, _p],
    'mpb_stomp_sample': [_p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _f, _f, _u64, _u32, _u32, _p],
    'mpb_stomp_update': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    'mpb_chomp_step': [_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _i, _p],
    'mpb_gpmp2_workspace_bytes': [_i, _i, _i],
    'mpb_gpmp2_linearize': [_p, _p, _i, _p, _i, _i, _i, _i, _p],
    'mpb_gpmp2_diag': [_p, _p, _i, _i, _i, _i, _f, _f, _f, _f, _f, _p],
    'mpb_gpmp2_solve': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _i, _f, _p],
    'mpb_gpmp2_step': [_p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _f, _f, _f, _f, _f, _f, _i, _f, _i, _i, _i, _p],
    'mpb_stoch_gpmp_costs': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i] + [_f] * 9 + [_p],
    'mpb_stoch_gpmp_step': [_p] * 11 + [_i] * 4 + [_f] * 10 + [_i, _u64, _p],
    'mpb_gp_prior_sample': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _u64, _p],
    'mpb_gp_prior_sample_dense': [_p, _p, _p, _p, _i, _i, _i, _i, _u64, _p],
    'mpb_mvn_sample_dense': [_p, _p, _p, _p, _i, _i, _i, _u64, _p],
    'mpb_stomp_plan_create': [_p] * 8 + [_i, _p, ctypes.c_size_t, _i, _i, _i, _i, _i, _f, _f, _f, _f, _u64, _u32, _p],
    'mpb_stomp_plan_launch': [_p, _i, _u32, _p, _p, _p],
    'mpb_stomp_plan_destroy': [_p],
    'mpb_point_dynamics': [_p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _i, _f, _p],
    'mpb_point_traj_cost': [_p, _p, _p, _p, _f, _f, _f, _f, _f, _p, _i, _i, _i, _i, _p],
    'mpb_mppi_step': [_p] * 11 + [_i] + [_p] * 6 + [_i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _i, _u64, _u32, _p],
}

# test aids (include/mpb_debug.h): a separate library, csrc/libmpb_hip_debug.so
DEBUG_SIGNATURES = {
    'mpb_debug_last_error': [],
    'mpb_debug_occupy': [_i, _u64, _p, _p],
    'mpb_debug_philox': [_p, _p, _p, _i, _i, _p],
    'mpb_debug_stomp_normals': [_p, _i, _i, _i, _i, _u64, _u32, _u32, _p],
    'mpb_debug_stomp_normals_h': [_p, _i, _i, _i, _i, _i, _u64, _u32, _u32, _p],
    'mpb_debug_mppi_normals': [_p, _i, _i, _i, _i, _i, _u64, _u32, _p],
}
ABI_VERSION = 6          # include/mpb.h MPB_ABI_VERSION

_lib = None
_debug_lib = None


class MPBError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; fail loudly if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MPBError(
            f'{LIB_PATH} not found: the HIP extension is not built. Run '
            f'`python -c "import __graft_entry__ as g; g.build()"` (or motion_planning_baselines_amd/build.py). '
            f'There is no CPU fallback.')
    # PyTorch-ROCm ships its own HIP runtime: it has to be in the process BEFORE this library is loaded, so that the
    # library's libamdhip64 dependency resolves to the runtime torch initialises (a second copy would see no device)
    import torch  # noqa: F401
    h = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise MPBError(f'{LIB_PATH} does not export {name} (declared in include/mpb.h)') from e
        fn.argtypes = argtypes
        fn.restype = (ctypes.c_char_p if name == 'mpb_last_error' else
                      ctypes.c_size_t if name in ('mpb_gpmp2_workspace_bytes', 'mpb_stomp_workspace_bytes') else ctypes.c_int)
    if (h.mpb_version() & 0xFFFF) != ABI_VERSION:
        # signatures change positionally between ABI versions: calling across them would pass pointers as ints
        raise MPBError(f'{LIB_PATH} reports ABI version {h.mpb_version() & 0xFFFF}, this binding is written for {ABI_VERSION} '
                       f'(include/mpb.h MPB_ABI_VERSION); rebuild with motion_planning_baselines_amd.build.build(force=True)')
    if (h.mpb_version() & 0x40000000) and not os.environ.get('MPB_LIB_PATH'):
        raise MPBError(f'{LIB_PATH} is a tuning build (compiled with wrong-result timing switches); rebuild with '
                       f'motion_planning_baselines_amd.build.build(force=True)')
    _lib = h
    return h


def debug_lib():
    """The test-aid library (include/mpb_debug.h); never loaded by a product path."""
    global _debug_lib
    if _debug_lib is not None:
        return _debug_lib
    path = os.path.join(_HERE, 'csrc', 'libmpb_hip_debug.so')
    if not os.path.exists(path):
        raise MPBError(f'{path} not found: run motion_planning_baselines_amd.build.build()')
    lib()                                        # (torch's HIP runtime first, as above)
    h = ctypes.CDLL(path)
    for name, argtypes in DEBUG_SIGNATURES.items():
        fn = getattr(h, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name == 'mpb_debug_last_error' else ctypes.c_int
    _debug_lib = h
    return h


def debug_check(code, what=''):
    if code != 0:
        msg = debug_lib().mpb_debug_last_error()
        raise MPBError(f'{what} failed with code {code}: {msg.decode() if msg else "?"}')


def check(code, what=''):
    if code != 0:
        msg = lib().mpb_last_error()
        raise MPBError(f'{what} failed with code {code}: {msg.decode() if msg else "?"}')


def geom_check(buf):
    """Validate a packed geometry buffer (numpy fp32) on the host side of the C-ABI."""
    buf = np.ascontiguousarray(buf, dtype=np.float32)
    check(lib().mpb_geom_check(buf.ctypes.data_as(ctypes.c_void_p), int(buf.size)), 'mpb_geom_check')


def geom_flags(buf):
    """mpb_geom_flags of a packed geometry buffer (numpy fp32, host): what lets a launcher pick its kernel."""
    buf = np.ascontiguousarray(buf, dtype=np.float32)
    out = ctypes.c_int(0)
    check(lib().mpb_geom_flags(buf.ctypes.data_as(ctypes.c_void_p), int(buf.size), ctypes.cast(ctypes.pointer(out), ctypes.c_void_p)),
          'mpb_geom_flags')
    return int(out.value)
