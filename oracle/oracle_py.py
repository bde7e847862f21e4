"""ctypes loader for the CPU oracle (oracle/alqp_oracle.c).

TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this. The product path (deq-mpc-corl_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libalqp_oracle.so")


def build(force=False):
    src_m = max(os.path.getmtime(os.path.join(_HERE, f))
                for f in ("alqp_oracle.c", "alqp_oracle_impl.h"))
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libalqp_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


def _np(dtype):
    return {"f64": np.float64, "f32": np.float32}[dtype]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class _Trace64(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in
                ("g", "d", "phi", "phi_prev", "z", "k", "accept", "Hd", "Hs")] + \
               [("max_steps", C.c_int)]


def _bounds(ulo, uhi, B, T, nu, dt):
    """Broadcastable bounds -> (array, batch stride, stage stride) in elements."""
    ulo = np.asarray(ulo, dtype=dt)
    uhi = np.asarray(uhi, dtype=dt)
    if ulo.ndim <= 1:
        ulo = np.broadcast_to(ulo, (nu,)).copy()
        uhi = np.broadcast_to(uhi, (nu,)).copy()
        return ulo, uhi, 0, 0
    ulo = np.ascontiguousarray(np.broadcast_to(ulo, (B, T, nu)))
    uhi = np.ascontiguousarray(np.broadcast_to(uhi, (B, T, nu)))
    return ulo, uhi, T * nu, nu


def max_threads():
    return int(lib().orc_max_threads())


def set_threads(n):
    lib().orc_set_threads(int(n))


_obs_keep = {}


class obstacles:
    """Context manager: the oracle calls inside see `nobs` obstacle rows per stage (Obstacle_MPC):
    pos [B,T,nobs,3], radius. Process-global C state (orc_set_obstacles), restored on exit."""

    def __init__(self, dtype, pos, radius):
        self.dtype, self.pos, self.radius = dtype, (None if pos is None else _c(pos, _np(dtype))), float(radius)

    def __enter__(self):
        fn = getattr(lib(), "orc_set_obstacles_" + self.dtype)
        if self.pos is None:
            fn(None, 0, C.c_double(0.0))
        else:
            _obs_keep[self.dtype] = self.pos
            fn(_p(self.pos), int(self.pos.shape[2]), C.c_double(self.radius))
        return self

    def __exit__(self, *exc):
        getattr(lib(), "orc_set_obstacles_" + self.dtype)(None, 0, C.c_double(0.0))
        _obs_keep.pop(self.dtype, None)


class state_estimator:
    """Context manager: the state-estimator variant (no initial-state rows, zero cost gradient on the controls)."""

    def __init__(self, dtype, on=True):
        self.dtype, self.on = dtype, bool(on)

    def __enter__(self):
        getattr(lib(), "orc_set_state_estimator_" + self.dtype)(int(self.on))
        return self

    def __exit__(self, *exc):
        getattr(lib(), "orc_set_state_estimator_" + self.dtype)(0)


def grad_hess(dtype, z, xnext, F, x0, lam, rho, Qd, q, ulo, uhi):
    dt = _np(dtype)
    B, T, n = z.shape
    nx = x0.shape[1]
    nu = n - nx
    z, xnext, F, x0, lam, Qd, q = (_c(a, dt) for a in (z, xnext, F, x0, lam, Qd, q))
    rho = _c(np.reshape(rho, (B,)), dt)
    ulo, uhi, sb, st = _bounds(ulo, uhi, B, T, nu, dt)
    g = np.empty((B, T, n), dt)
    Hd = np.empty((B, T, n, n), dt)
    Hs = np.empty((B, max(T - 1, 1), n, n), dt)
    getattr(lib(), "orc_grad_hess_" + dtype)(
        B, T, nx, nu, _p(z), _p(xnext), _p(F), _p(x0), _p(lam), _p(rho), _p(Qd), _p(q),
        _p(ulo), _p(uhi), C.c_long(sb), C.c_long(st), _p(g), _p(Hd), _p(Hs))
    return g, Hd, Hs[:, : T - 1]


def newton_dir(dtype, g, Hd, Hs, nx, solver=0, want_factor=False):
    dt = _np(dtype)
    B, T, n = g.shape
    nu = n - nx
    g, Hd, Hs = _c(g, dt), _c(Hd, dt), _c(Hs, dt)
    d = np.empty((B, T, n), dt)
    info = np.zeros(B, np.int32)
    L = np.empty((B, T, n, n), dt) if want_factor else None
    W = np.empty((B, max(T - 1, 1), nx, n), dt) if want_factor else None
    getattr(lib(), "orc_newton_dir_" + dtype)(
        B, T, nx, nu, solver, _p(g), _p(Hd), _p(Hs), _p(d), _p(L), _p(W), _p(info))
    if want_factor:
        return d, info, L, W
    return d, info


def merit(dtype, z, xnext, x0, lam, rho, Qd, q, ulo, uhi):
    dt = _np(dtype)
    B, T, n = z.shape
    nx = x0.shape[1]
    nu = n - nx
    z, xnext, x0, lam, Qd, q = (_c(a, dt) for a in (z, xnext, x0, lam, Qd, q))
    rho = _c(np.reshape(rho, (B,)), dt)
    ulo, uhi, sb, st = _bounds(ulo, uhi, B, T, nu, dt)
    phi = np.empty(B, dt)
    rp2 = np.empty(B, dt)
    getattr(lib(), "orc_merit_" + dtype)(
        B, T, nx, nu, _p(z), _p(xnext), _p(x0), _p(lam), _p(rho), _p(Qd), _p(q),
        _p(ulo), _p(uhi), C.c_long(sb), C.c_long(st), _p(phi), _p(rp2))
    return phi, rp2


def linesearch_pick(dtype, phi_all, phi_prev):
    dt = _np(dtype)
    n_ls, B = phi_all.shape
    phi_all, phi_prev = _c(phi_all, dt), _c(phi_prev, dt)
    k = np.empty(B, np.int32)
    acc = np.empty(B, np.int32)
    pm = np.empty(B, dt)
    getattr(lib(), "orc_linesearch_pick_" + dtype)(
        B, n_ls, _p(phi_all), _p(phi_prev), _p(k), _p(acc), _p(pm))
    return k, acc, pm


def dual_update(dtype, z, xnext, x0, ulo, uhi, lam, rho):
    dt = _np(dtype)
    B, T, n = z.shape
    nx = x0.shape[1]
    nu = n - nx
    z, xnext, x0 = (_c(a, dt) for a in (z, xnext, x0))
    lam = np.array(lam, dtype=dt, order="C", copy=True)
    rho = np.array(np.reshape(rho, (B,)), dtype=dt, order="C", copy=True)
    ulo, uhi, sb, st = _bounds(ulo, uhi, B, T, nu, dt)
    getattr(lib(), "orc_dual_update_" + dtype)(
        B, T, nx, nu, _p(z), _p(xnext), _p(x0), _p(ulo), _p(uhi), C.c_long(sb), C.c_long(st),
        _p(lam), _p(rho))
    return lam, rho


def solve_lin(dtype, Qd, q, F, c, x0, ulo, uhi, z0, lam0=None, rho0=None, al_iter=2,
              max_newton=4, n_ls=20, exit_mode="fixed", solver="banded", trace_steps=0,
              save_factor=False):
    """Whole LinDx solve. Returns a dict (z, lam, rho, status, newton_per_al, trace...)."""
    dt = _np(dtype)
    B, T, n = Qd.shape
    nx = x0.shape[1]
    nu = n - nx
    M = T * nx + 2 * T * nu
    Qd, q, F, c, x0 = (_c(a, dt) for a in (Qd, q, F, c, x0))
    z = np.array(z0, dtype=dt, order="C", copy=True)
    lam = np.zeros((B, M), dt) if lam0 is None else np.array(lam0, dtype=dt, order="C", copy=True)
    rho = np.ones(B, dt) if rho0 is None else np.array(np.reshape(rho0, (B,)), dtype=dt, order="C", copy=True)
    ulo, uhi, sb, st = _bounds(ulo, uhi, B, T, nu, dt)
    status = np.ones(B, np.uint8)
    npa = np.zeros(max(al_iter, 1), np.int32)
    tr = None
    out = {}
    if trace_steps:
        S = trace_steps
        out["g"] = np.zeros((S, B, T, n), dt)
        out["d"] = np.zeros((S, B, T, n), dt)
        out["phi"] = np.zeros((S, n_ls, B), dt)
        out["phi_prev"] = np.zeros((S, B), dt)
        out["z_steps"] = np.zeros((S, B, T, n), dt)
        out["k"] = np.zeros((S, B), np.int32)
        out["accept"] = np.zeros((S, B), np.int32)
        out["Hd"] = np.zeros((al_iter, B, T, n, n), dt)
        out["Hs"] = np.zeros((al_iter, B, max(T - 1, 1), n, n), dt)
        tr = _Trace64(_p(out["g"]), _p(out["d"]), _p(out["phi"]), _p(out["phi_prev"]),
                      _p(out["z_steps"]), _p(out["k"]), _p(out["accept"]), _p(out["Hd"]),
                      _p(out["Hs"]) if T > 1 else None, S)
    Ls = np.zeros((B, T, n, n), dt) if save_factor else None
    zs = np.zeros((B, T, n), dt) if save_factor else None
    fn = getattr(lib(), "orc_solve_lin_" + dtype)
    fn.restype = C.c_int
    total = fn(B, T, nx, nu, al_iter, max_newton, n_ls,
               {"fixed": 0, "reference": 1}[exit_mode], {"banded": 0, "dense": 1}[solver],
               _p(Qd), _p(q), _p(F), _p(c), _p(x0), _p(ulo), _p(uhi), C.c_long(sb), C.c_long(st),
               _p(z), _p(lam), _p(rho), _p(status), _p(npa),
               C.byref(tr) if tr is not None else None, _p(Ls), _p(zs))
    out.update(z=z, lam=lam, rho=rho, status=status.astype(bool), newton_per_al=npa[:al_iter],
               total_steps=total)
    if trace_steps and T > 1:
        out["Hs"] = out["Hs"][:, :, : T - 1]
    if save_factor:
        out["L"] = Ls
        out["z_saved"] = zs
    return out


def backward(dtype, L, F, rho, z_saved, gbar):
    dt = _np(dtype)
    B, T, n, _ = L.shape
    nx = F.shape[2] if F.ndim == 4 and F.shape[1] > 0 else None
    if nx is None:
        raise ValueError("need F[B,T-1,nx,n]")
    nu = n - nx
    L, F, z_saved, gbar = (_c(a, dt) for a in (L, F, z_saved, gbar))
    rho = _c(np.reshape(rho, (B,)), dt)
    qg = np.empty((B, T, n), dt)
    Qg = np.empty((B, T, n), dt)
    getattr(lib(), "orc_backward_" + dtype)(
        B, T, nx, nu, _p(L), _p(F), _p(rho), _p(z_saved), _p(gbar), _p(qg), _p(Qg))
    return qg, Qg
