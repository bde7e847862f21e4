"""ctypes binding of libmi_alqp.so (include/mi_alqp.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``deq-mpc-corl_amd/csrc/build.sh`` with hipcc for gfx950. There is NO fallback:
if the shared object is missing this module raises, and so does every solver
entry point (the product path never routes through a CPU implementation).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_ALQP_LIB") or os.path.join(_HERE, "csrc", "libmi_alqp.so")   # MI_ALQP_LIB: A/B experiments with a second build of the same ABI

ABI_VERSION = 10


class AlqpDims(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("nx", C.c_int), ("nu", C.c_int)]


class AlqpParams(C.Structure):
    _fields_ = [("al_iter", C.c_int), ("max_newton", C.c_int), ("n_ls", C.c_int),
                ("flags", C.c_int), ("rho_scale", C.c_double), ("variant", C.c_int),
                ("skip_flag", C.c_void_p), ("exit_tol", C.c_double), ("newton_counts", C.c_void_p),
                ("exit_scratch", C.c_void_p), ("quad_stagger", C.c_int)]


class AlqpTrace(C.Structure):
    _fields_ = [("g", C.c_void_p), ("d", C.c_void_p), ("phi", C.c_void_p),
                ("phi_prev", C.c_void_p), ("k", C.c_void_p), ("accept", C.c_void_p)]


class AlqpObstacles(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("radius", C.c_double), ("nobs", C.c_int), ("state_estimator", C.c_int)]


class AlqpIpmParams(C.Structure):
    _fields_ = [("flags", C.c_int), ("max_iter", C.c_int), ("iter0", C.c_int), ("kkt_eps", C.c_double),
                ("variant", C.c_int)]


class AlqpRigidParams(C.Structure):
    _fields_ = [("mass", C.c_double), ("J", C.c_double * 9), ("Jinv", C.c_double * 9), ("g", C.c_double * 3),
                ("motor_dist", C.c_double), ("kf", C.c_double), ("bf", C.c_double), ("km", C.c_double),
                ("act_scale", C.c_double), ("u_hover", C.c_double), ("pend_L", C.c_double), ("ss", C.c_double * 12),
                ("bf_force", C.c_double)]


IPM_VARIANTS = {"auto": 0, "generic_lds": 1, "generic_ws": 2, "resident": 3}   # ALQP_IPM_VARIANT_*


ALQP_IPM_INIT, ALQP_IPM_RESID, ALQP_IPM_STEP, ALQP_IPM_LOOP, ALQP_IPM_FINAL = 1, 2, 4, 8, 16
ALQP_INIT_MERIT = 1
ALQP_DUAL_UPDATE = 2
ALQP_SAVE_FACTOR = 4
ALQP_WS_PRIMED = 8
ALQP_EXIT_IN_KERNEL = 16
ALQP_E_COOP = -4
VARIANT_AUTO, VARIANT_TEAM, VARIANT_QUAD = 0, 1, 2

ERRORS = {-1: "bad argument", -2: "unsupported (nx, nu) or horizon does not fit in LDS",
          -3: "kernel launch failed", -4: "grid too large for a cooperative launch"}

_P = C.c_void_p
_SIGS = {
    # name: (restype, argtypes) ; the f32/f64 pairs share a signature
    "alqp_solve_nonlin": (C.c_int, [C.POINTER(AlqpDims), C.POINTER(AlqpParams), C.c_int, C.c_double, _P, _P, _P, _P, _P,
                                    C.c_long, C.c_long, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "alqp_dyn_pendulum1l": (C.c_int, [C.c_long, _P, _P, C.c_double, _P, _P, _P, _P]),
    "alqp_dyn_cartpole1l": (C.c_int, [C.c_long, _P, _P, C.c_double, _P, _P, _P, _P]),
    "alqp_dyn_cartpole1l_v2": (C.c_int, [C.c_long, _P, _P, C.c_double, _P, _P, _P, _P]),
    "alqp_dyn_cartpole2l": (C.c_int, [C.c_long, _P, _P, C.c_double, _P, _P, _P, _P]),
    "alqp_dyn_rexquadrotor": (C.c_int, [C.c_long, C.POINTER(AlqpRigidParams), _P, _P, C.c_double, _P, _P, _P]),
    "alqp_dyn_flyingcartpole": (C.c_int, [C.c_long, C.POINTER(AlqpRigidParams), _P, _P, C.c_double, _P, _P, _P]),
    "alqp_solve_lin": (C.c_int, [C.POINTER(AlqpDims), C.POINTER(AlqpParams), _P, _P, _P, _P, _P, _P, _P,
                                 C.c_long, C.c_long, _P, _P, _P, _P, _P, _P, _P, _P,
                                 C.POINTER(AlqpTrace), _P, C.c_size_t, _P]),
    "alqp_newton_step": (C.c_int, [C.POINTER(AlqpDims), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                   C.c_long, C.c_long, _P, _P, _P, _P, _P]),
    "alqp_merit": (C.c_int, [C.POINTER(AlqpDims), C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                             C.c_long, C.c_long, _P, _P, _P]),
    "alqp_linesearch_pick": (C.c_int, [C.POINTER(AlqpDims), C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "alqp_dual_update": (C.c_int, [C.POINTER(AlqpDims), _P, _P, _P, _P, _P, C.c_long, C.c_long, _P, _P,
                                   C.c_double, _P]),
    "alqp_backward": (C.c_int, [C.POINTER(AlqpDims), _P, _P, _P, _P, _P, _P, _P, _P]),
    "alqp_backward_ws": (C.c_int, [C.POINTER(AlqpDims), _P, C.c_size_t, _P, _P, _P, _P, _P, _P, _P]),
    "alqp_newton_step_obs": (C.c_int, [C.POINTER(AlqpDims), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                       C.c_long, C.c_long, C.POINTER(AlqpObstacles), _P, _P, _P, _P, _P]),
    "alqp_merit_obs": (C.c_int, [C.POINTER(AlqpDims), C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                 C.c_long, C.c_long, C.POINTER(AlqpObstacles), _P, _P, _P]),
    "alqp_dual_update_obs": (C.c_int, [C.POINTER(AlqpDims), _P, _P, _P, _P, _P, C.c_long, C.c_long,
                                       C.POINTER(AlqpObstacles), _P, _P, C.c_double, _P]),
    "alqp_newton_step_ws": (C.c_int, [C.POINTER(AlqpDims), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                      C.c_long, C.c_long, _P, C.c_size_t, _P, _P, _P, _P]),
    "alqp_newton_step_ws_obs": (C.c_int, [C.POINTER(AlqpDims), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                          C.c_long, C.c_long, C.POINTER(AlqpObstacles), _P, C.c_size_t, _P, _P, _P, _P]),
    "alqp_merit_pick": (C.c_int, [C.POINTER(AlqpDims), C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                  C.c_long, C.c_long, C.POINTER(AlqpObstacles), _P, _P, _P, _P, _P, _P, _P]),
    "alqp_ipm_solve": (C.c_int, [C.POINTER(AlqpDims), C.POINTER(AlqpIpmParams), _P, _P, _P, _P, _P, _P, _P,
                                 C.c_long, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long, _P, C.c_size_t, _P,
                                 _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "alqp_ipm_backward": (C.c_int, [C.POINTER(AlqpDims), _P, _P, C.c_long, C.c_long, C.c_long, C.c_long, _P, _P, _P,
                                    _P, C.c_size_t, _P, _P, _P, _P, C.c_int, _P]),
}
_PLAIN = {
    "alqp_abi_version": (C.c_int, []),
    "alqp_supported": (C.c_int, [C.POINTER(AlqpDims), C.c_int]),
    "alqp_supported_variant": (C.c_int, [C.POINTER(AlqpDims), C.c_int, C.c_int]),
    "alqp_lds_bytes": (C.c_size_t, [C.POINTER(AlqpDims), C.c_int]),
    "alqp_qps_per_wave": (C.c_int, [C.POINTER(AlqpDims), C.c_int]),
    "alqp_workspace_bytes": (C.c_size_t, [C.POINTER(AlqpDims), C.c_int]),
    "alqp_exit_test": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "alqp_workspace_bytes_nonlin": (C.c_size_t, [C.POINTER(AlqpDims), C.c_int]),
    "alqp_ipm_workspace_bytes": (C.c_size_t, [C.POINTER(AlqpDims), C.c_int]),
}

EXPORTED_SYMBOLS = sorted([f"{n}_{s}" for n in _SIGS for s in ("f32", "f64")] + list(_PLAIN))

_lib = None


def load():
    """Load libmi_alqp.so (once). Raises RuntimeError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"mi_alqp: HIP extension not built ({LIB_PATH} missing). Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` - there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        for sfx in ("f32", "f64"):
            fn = getattr(lib, f"{name}_{sfx}")
            fn.restype = res
            fn.argtypes = args
    for name, (res, args) in _PLAIN.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.alqp_abi_version() != ABI_VERSION:
        raise RuntimeError("mi_alqp: libmi_alqp.so ABI version mismatch - rebuild the extension")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"mi_alqp: {what} failed: {ERRORS.get(rc, rc)}")
