"""ctypes loader for the CPU wave emulation of the register-resident interior-point kernel
(tests/emu/ipm_g4_emu.cpp = deq-mpc-corl_amd/csrc/alqp_ipm_g4.hpp + tests/emu/wave_emu.hpp).

TEST INFRASTRUCTURE: the product path never imports this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_CSRC = os.path.join(_ROOT, "deq-mpc-corl_amd", "csrc")
_LIB = os.path.join(_HERE, "libipm_g4_emu.so")
_lib = None
INIT, RESID, STEP, LOOP, FINAL = 1, 2, 4, 8, 16


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ipm_g4_emu.cpp", "wave_emu.hpp")] + \
           [os.path.join(_CSRC, f) for f in ("alqp_ipm_g4.hpp", "alqp_ipm_args.hpp", "alqp_dims.hpp")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I" + _CSRC,
                               "-I" + os.path.join(_ROOT, "include"), os.path.join(_HERE, "ipm_g4_emu.cpp"), "-o", _LIB])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Solve:
    """One solve on batch-major data (Qd, p [B,T,n]; F [B,T-1,nx,n]; f [B,T-1,nx]; x0 [B,nx]; uhi, ulo [nu]):
    `launch(flags, ...)` = one kernel launch with those ALQP_IPM_* flags; the workspace persists between launches."""

    def __init__(self, dtype, Qd, p, F, f, x0, uhi, ulo, kkt_eps=1e-7):
        self.sfx = dtype
        dt = {"f64": np.float64, "f32": np.float32}[dtype]
        self.dt = dt
        B, T, n = Qd.shape
        nx = x0.shape[1]
        nu = n - nx
        self.dims = (B, T, nx, nu)
        a = lambda v: np.ascontiguousarray(v, dtype=dt)
        self.Qd, self.p, self.F, self.f, self.x0 = a(Qd), a(p), a(F), a(f), a(x0)
        self.uhi, self.ulo = a(np.broadcast_to(uhi, (nu,))), a(np.broadcast_to(ulo, (nu,)))
        fw = getattr(lib(), "emu_ipm_ws_words_" + dtype)
        fw.restype = C.c_size_t
        self.wsw = int(fw(nx, nu, T))
        assert self.wsw > 0, "no emulated instance for these sizes"
        self.ws = np.full((B, self.wsw), np.nan, dt)   # nothing may depend on what the workspace held before
        self.kkt_eps = kkt_eps
        self.out = dict(zhat=np.zeros((B, T * n), dt), nus=np.zeros((B, T * nx), dt), lams=np.zeros((B, 2 * T * nu), dt),
                        slacks=np.zeros((B, 2 * T * nu), dt), resid=np.zeros(B, dt), mu=np.zeros(B, dt),
                        iter_best=np.zeros(B, np.int32), improved=np.zeros(B, np.int32), info=np.zeros(B, np.int32))

    def launch(self, flags, max_iter=0, iter0=0, ry=None):
        B, T, nx, nu = self.dims
        n = nx + nu
        o = self.out
        fn = getattr(lib(), "emu_ipm_solve_" + self.sfx)
        fn.argtypes = [C.c_int] * 7 + [C.c_double] + [C.c_void_p] * 7 + [C.c_long] * 6 + [C.c_void_p] * 11
        fn.restype = C.c_int
        ry = None if ry is None else np.ascontiguousarray(ry, dtype=self.dt)
        rc = fn(B, T, nx, nu, flags, max_iter, iter0, self.kkt_eps, _p(self.Qd), _p(self.p), _p(self.F), _p(self.f),
                _p(self.x0), _p(self.uhi), _p(self.ulo), n, T * n, nx * n, (T - 1) * nx * n, nx, (T - 1) * nx, _p(self.ws),
                _p(ry), _p(o["zhat"]), _p(o["nus"]), _p(o["lams"]), _p(o["slacks"]), _p(o["resid"]), _p(o["mu"]),
                _p(o["iter_best"]), _p(o["improved"]), _p(o["info"]))
        assert rc == 0, rc
        return o

    def cur_x(self):
        B, T, nx, nu = self.dims
        return self.ws[:, :T * (nx + nu)].copy()


def forward(dtype, Qd, p, F, f, x0, uhi, ulo, exit_mode="fixed", eps=1e-12, not_improved_lim=3, max_iter=20, ry_fn=None):
    """The host loop of backend.ipm_solve on the emulated kernel."""
    s = Solve(dtype, Qd, p, F, f, x0, uhi, ulo)
    if exit_mode == "fixed" and ry_fn is None:
        o = s.launch(INIT | LOOP | FINAL, max_iter)
        o["iters"] = max_iter
        return o
    s.launch(INIT)
    n_not, done = 0, max_iter
    for it in range(max_iter):
        ry = ry_fn(s.cur_x()) if ry_fn is not None else None
        o = s.launch(RESID, 0, it, ry)
        if exit_mode == "reference":
            n_not = 0 if (it == 0 or o["improved"].max() > 0) else n_not + 1
            if n_not == not_improved_lim or o["resid"].max() < eps or o["mu"].min() > 1e32:
                done = it
                break
        s.launch(STEP, 0, it)
    o = s.launch(FINAL)
    o["iters"] = done
    return o


def backward(dtype, Qd, F, lams, slacks, g):
    dt = {"f64": np.float64, "f32": np.float32}[dtype]
    B, T, n = Qd.shape
    nx = F.shape[2]
    nu = n - nx
    a = lambda v: np.ascontiguousarray(v, dtype=dt)
    Qd, F, lams, slacks, g = a(Qd), a(F), a(lams), a(slacks), a(g)
    fw = getattr(lib(), "emu_ipm_ws_words_" + dtype)
    fw.restype = C.c_size_t
    ws = np.full((B, int(fw(nx, nu, T))), np.nan, dt)
    dx, dlam, dnu = np.zeros((B, T * n), dt), np.zeros((B, 2 * T * nu), dt), np.zeros((B, T * nx), dt)
    info = np.zeros(B, np.int32)
    fn = getattr(lib(), "emu_ipm_backward_" + dtype)
    fn.argtypes = [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_long] * 4 + [C.c_void_p] * 8
    rc = fn(B, T, nx, nu, _p(Qd), _p(F), n, T * n, nx * n, (T - 1) * nx * n, _p(lams), _p(slacks), _p(g), _p(ws), _p(dx),
            _p(dlam), _p(dnu), _p(info))
    assert rc == 0, rc
    return dx, dlam, dnu
