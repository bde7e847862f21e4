"""ctypes loader for the interior-point CPU oracle (oracle/ipm_oracle.c).

TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this. The product path (deq-mpc-corl_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libipm_oracle.so")
_lib = None


def build(force=False):
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ipm_oracle.c", "ipm_oracle_impl.h"))
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libipm_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
    return _lib


def _np(dtype):
    return {"f64": np.float64, "f32": np.float32}[dtype]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def forward(dtype, Qd, p, F, f, x0, uhi, ulo, solver=0, exit_mode=0, eps=1e-12, not_improved_lim=3, max_iter=20,
            ry_fn=None):
    """pdipm_b_LU.forward on batch-major data: Qd, p [B,T,n]; F [B,T-1,nx,n]; f [B,T-1,nx]; x0 [B,nx];
    uhi, ulo [nu]. ry_fn(x [B,T*n]) -> [B,T*nx]: the caller's equality residual (true dynamics), or None
    for A x - b. Returns dict(zhat, nus, lams, slacks, resid, iters, iter_best, init_*)."""
    dt = _np(dtype)
    B, T, n = Qd.shape
    nx = x0.shape[1]
    nu = n - nx
    nz, ni, ne = T * n, 2 * T * nu, T * nx
    a = lambda v: np.ascontiguousarray(v, dtype=dt)
    Qd, p, F, f, x0 = a(Qd), a(p), a(F), a(f), a(x0)
    uhi, ulo = a(np.broadcast_to(uhi, (nu,))), a(np.broadcast_to(ulo, (nu,)))
    out = {k: np.zeros((B, m), dt) for k, m in (("zhat", nz), ("nus", ne), ("lams", ni), ("slacks", ni),
                                                ("init_x", nz), ("init_s", ni), ("init_z", ni), ("init_y", ne))}
    resid = np.zeros(B, dt)
    iter_best = np.zeros(B, np.int32)
    info = np.zeros(B, np.int32)
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)
    cb = None
    if ry_fn is not None:
        def _cb(xp, ryp, _ctx):
            x = np.ctypeslib.as_array(C.cast(xp, C.POINTER(C.c_double if dt == np.float64 else C.c_float)), shape=(B, nz))
            ry = np.ctypeslib.as_array(C.cast(ryp, C.POINTER(C.c_double if dt == np.float64 else C.c_float)), shape=(B, ne))
            ry[...] = np.asarray(ry_fn(x.copy()), dtype=dt)
        cb = CB(_cb)
    fn = getattr(lib(), "orc_ipm_forward_" + dtype)
    fn.restype = C.c_int
    iters = fn(B, T, nx, nu, _p(Qd), _p(p), _p(F), _p(f), _p(x0), _p(uhi), _p(ulo), int(solver), int(exit_mode),
               C.c_double(eps), int(not_improved_lim), int(max_iter), cb, None,
               _p(out["zhat"]), _p(out["nus"]), _p(out["lams"]), _p(out["slacks"]), _p(resid), _p(iter_best),
               _p(out["init_x"]), _p(out["init_s"]), _p(out["init_z"]), _p(out["init_y"]), _p(info))
    out.update(resid=resid, iters=int(iters), iter_best=iter_best, info=info)
    return out


def backward(dtype, Qd, F, lams, slacks, g, solver=0):
    """DenseQPFunction.backward's KKT solve: returns dx [B,nz], dlam [B,ni], dnu [B,ne]."""
    dt = _np(dtype)
    B, T, n = Qd.shape
    nx = F.shape[2]
    nu = n - nx
    nz, ni, ne = T * n, 2 * T * nu, T * nx
    a = lambda v: np.ascontiguousarray(v, dtype=dt)
    Qd, F, lams, slacks, g = a(Qd), a(F), a(lams), a(slacks), a(g)
    dx, dlam, dnu = np.zeros((B, nz), dt), np.zeros((B, ni), dt), np.zeros((B, ne), dt)
    getattr(lib(), "orc_ipm_backward_" + dtype)(B, T, nx, nu, _p(Qd), _p(F), _p(lams), _p(slacks), _p(g), int(solver),
                                                _p(dx), _p(dlam), _p(dnu))
    return dx, dlam, dnu
