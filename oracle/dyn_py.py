"""ctypes loaders for the dynamics-provider oracle.

TEST INFRASTRUCTURE - only tests/ and tools/ import this; the product path never does.
  restatement : oracle/dyn_oracle.c (ours)               -> pendulum1l(x, u, h)
  reference   : oracle/_ref/libpendulum1l_casadi.so, the reference's CasADi-generated code compiled
                from its own sources by `make -C oracle ref` (present in the build container and,
                as a built file, on the GPU box; the sources are never copied)  -> pendulum1l_ref
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libdyn_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libpendulum1l_casadi.so")
_REF_CP = os.path.join(_HERE, "_ref", "libcartpole1l_casadi.so")
_REF_CPV2 = os.path.join(_HERE, "_ref", "libcartpole1l_v2_casadi.so")
_REF_CP2 = os.path.join(_HERE, "_ref", "libcartpole2l_casadi.so")


def build():
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "dyn_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdyn_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/deqmpc/my_envs/pendulum1l/src") and not (os.path.exists(_REF) and os.path.exists(_REF_CP) and os.path.exists(_REF_CP2)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return _LIB


def pendulum1l(x, u, h):
    """x [K,2] = (theta, omega), u [K,1] = tau (float64) -> xn [K,2], A [K,2,2], B [K,2,1]."""
    build()
    lib = C.CDLL(_LIB)
    x = np.ascontiguousarray(x, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    K = x.shape[0]
    xn, A, B = np.empty((K, 2)), np.empty((K, 2, 2)), np.empty((K, 2, 1))
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.dyn_pendulum1l.argtypes = [C.c_long, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dyn_pendulum1l(K, P(x), P(u), float(h), P(xn), P(A), P(B))
    return xn, A, B


def have_ref():
    return os.path.exists(_REF)


def pendulum1l_ref(x, u, h):
    """The same through the reference's generated code (CasADi C ABI:
    f(const double** arg, double** res, long long* iw, double* w, int mem), generated_dynamics.c:142,
    inputs (q, qdot, tau, h), outputs in the order of dynamics_cpu.cpp:60-107)."""
    lib = C.CDLL(_REF)
    x = np.ascontiguousarray(x, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    K = x.shape[0]
    xn, A, B = np.empty((K, 2)), np.empty((K, 2, 2)), np.empty((K, 2, 1))
    dbl = C.c_double
    argT, res2T, res6T = C.POINTER(dbl) * 4, C.POINTER(dbl) * 2, C.POINTER(dbl) * 6
    for fn in (lib.eval_forward_dynamics, lib.eval_forward_derivatives):
        fn.restype = C.c_int
    iw = (C.c_longlong * 16)()
    w = (dbl * 256)()
    for i in range(K):
        ins = [dbl(x[i, 0]), dbl(x[i, 1]), dbl(u[i, 0]), dbl(h)]
        arg = argT(*[C.pointer(v) for v in ins])
        o2 = [dbl(), dbl()]
        lib.eval_forward_dynamics(arg, res2T(*[C.pointer(v) for v in o2]), iw, w, 0)
        xn[i] = [o2[0].value, o2[1].value]
        o6 = [dbl() for _ in range(6)]
        lib.eval_forward_derivatives(arg, res6T(*[C.pointer(v) for v in o6]), iw, w, 0)
        # (dq'/dq, dq'/dqdot, dq'/dtau, dqdot'/dq, dqdot'/dqdot, dqdot'/dtau)
        A[i] = [[o6[0].value, o6[1].value], [o6[3].value, o6[4].value]]
        B[i] = [[o6[2].value], [o6[5].value]]
    return xn, A, B


def cartpole1l(x, tau, h, fn="dyn_cartpole1l"):
    """x [K,4] = (cart x, theta, xdot, thetadot), tau [K,2] -> xn [K,4], J [K,4,6] = d xn / d(q, qd, tau)."""
    build()
    lib = C.CDLL(_LIB)
    x = np.ascontiguousarray(x, np.float64)
    tau = np.ascontiguousarray(tau, np.float64)
    K = x.shape[0]
    xn, J = np.empty((K, 4)), np.empty((K, 4, 6))
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    f = getattr(lib, fn)
    f.argtypes = [C.c_long, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    f(K, P(x), P(tau), float(h), P(xn), P(J))
    return xn, J


def cartpole1l_v2(x, tau, h):
    """The cartpole1l_v2 package's constants (oracle/dyn_oracle.c CART_V2)."""
    return cartpole1l(x, tau, h, fn="dyn_cartpole1l_v2")


def have_ref_cartpole():
    return os.path.exists(_REF_CP)


def have_ref_cartpole_v2():
    return os.path.exists(_REF_CPV2)


def cartpole1l_v2_ref(x, tau, h):
    return cartpole1l_ref(x, tau, h, path=_REF_CPV2)


def cartpole1l_ref(x, tau, h, path=None):
    """The same through the reference's generated code. Outputs of eval_forward_derivatives are six
    2x2 blocks (dq'/dq, dq'/dqdot, dq'/dtau, dqdot'/dq, dqdot'/dqdot, dqdot'/dtau), each stored
    COLUMN-major (CasADi dense), cf. cartpole1l/src/dynamics_cpu.cpp."""
    lib = C.CDLL(path or _REF_CP)
    x = np.ascontiguousarray(x, np.float64)
    tau = np.ascontiguousarray(tau, np.float64)
    K = x.shape[0]
    xn, J = np.empty((K, 4)), np.empty((K, 4, 6))
    dbl = C.c_double
    iw = (C.c_longlong * 64)()
    w = (dbl * 4096)()
    for i in range(K):
        ins = [(dbl * 2)(*x[i, :2]), (dbl * 2)(*x[i, 2:]), (dbl * 2)(*tau[i]), (dbl * 1)(h)]
        arg = (C.POINTER(dbl) * 4)(*[C.cast(a, C.POINTER(dbl)) for a in ins])
        o2 = [(dbl * 2)(), (dbl * 2)()]
        lib.eval_forward_dynamics(arg, (C.POINTER(dbl) * 2)(*[C.cast(a, C.POINTER(dbl)) for a in o2]), iw, w, 0)
        xn[i] = list(o2[0]) + list(o2[1])
        o6 = [(dbl * 4)() for _ in range(6)]
        lib.eval_forward_derivatives(arg, (C.POINTER(dbl) * 6)(*[C.cast(a, C.POINTER(dbl)) for a in o6]), iw, w, 0)
        blk = [np.array(list(b)).reshape(2, 2, order="F") for b in o6]
        J[i, :2, 0:2], J[i, :2, 2:4], J[i, :2, 4:6] = blk[0], blk[1], blk[2]
        J[i, 2:, 0:2], J[i, 2:, 2:4], J[i, 2:, 4:6] = blk[3], blk[4], blk[5]
    return xn, J


def cartpole2l(x, tau, h):
    """x [K,6] = (cart x, th1, th2 (relative), rates), tau [K,3] -> xn [K,6], J [K,6,9]."""
    build()
    lib = C.CDLL(_LIB)
    x = np.ascontiguousarray(x, np.float64)
    tau = np.ascontiguousarray(tau, np.float64)
    K = x.shape[0]
    xn, J = np.empty((K, 6)), np.empty((K, 6, 9))
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.dyn_cartpole2l.argtypes = [C.c_long, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    lib.dyn_cartpole2l(K, P(x), P(tau), float(h), P(xn), P(J))
    return xn, J


def have_ref_cartpole2():
    return os.path.exists(_REF_CP2)


def cartpole2l_ref(x, tau, h):
    """The same through the reference's generated code (six 3x3 blocks, column-major)."""
    lib = C.CDLL(_REF_CP2)
    x = np.ascontiguousarray(x, np.float64)
    tau = np.ascontiguousarray(tau, np.float64)
    K = x.shape[0]
    xn, J = np.empty((K, 6)), np.empty((K, 6, 9))
    dbl = C.c_double
    iw = (C.c_longlong * 256)()
    w = (dbl * 16384)()
    for i in range(K):
        ins = [(dbl * 3)(*x[i, :3]), (dbl * 3)(*x[i, 3:]), (dbl * 3)(*tau[i]), (dbl * 1)(h)]
        arg = (C.POINTER(dbl) * 4)(*[C.cast(a, C.POINTER(dbl)) for a in ins])
        o2 = [(dbl * 3)(), (dbl * 3)()]
        lib.eval_forward_dynamics(arg, (C.POINTER(dbl) * 2)(*[C.cast(a, C.POINTER(dbl)) for a in o2]), iw, w, 0)
        xn[i] = list(o2[0]) + list(o2[1])
        o6 = [(dbl * 9)() for _ in range(6)]
        lib.eval_forward_derivatives(arg, (C.POINTER(dbl) * 6)(*[C.cast(a, C.POINTER(dbl)) for a in o6]), iw, w, 0)
        blk = [np.array(list(b)).reshape(3, 3, order="F") for b in o6]
        for bi in range(2):
            for bj in range(3):
                J[i, 3 * bi:3 * bi + 3, 3 * bj:3 * bj + 3] = blk[3 * bi + bj]
    return xn, J
