"""Device backend: torch tensors -> raw device pointers -> libmi_alqp.so.

PyTorch is plumbing here (allocator, streams); every number is produced by the
hand-written gfx950 kernels in csrc/. All methods enqueue on torch's current HIP
stream and never synchronise.
"""
from __future__ import annotations

import ctypes as C

import os

import torch

from . import _lib


def _dt(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError(f"mi_alqp: unsupported dtype {t.dtype}")


def _ptr(t, name, dtype=None, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError(f"mi_alqp: {name} is required")
    if not t.is_cuda:
        raise RuntimeError(
            f"mi_alqp: {name} lives on {t.device}; the solver only runs on a ROCm device "
            "(no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError(f"mi_alqp: {name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"mi_alqp: {name} has dtype {t.dtype}, expected {dtype}")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# experiments only (profiles/r02/experiments): bits 24-31 of AlqpParams.flags reach the quad kernel untouched
_DEBUG_FLAGS = ((int(os.environ.get("ALQP_DEBUG_STAGGER", "0")) & 0xff) << 24) | ((int(os.environ.get("ALQP_DEBUG_STAGGER_CU", "0")) & 0xf) << 20)


class HipBackend:
    """The product backend: thin argument marshalling over the C ABI."""

    name = "hip"
    supports_exit_in_kernel = True   # solve_lin(newton_counts=...): ALQP_EXIT_IN_KERNEL, cooperative launch
    default_variant = "auto"
    # "auto": the quad variant (16 instances per wavefront, factor streamed through HBM) wins once
    # the batch fills the chip; below that the team variant (one instance per lane team, factor in
    # LDS) has 2-2.4x lower latency (measured on MI355X at (13,4) T=20: B=128 0.83 vs 1.94 ms,
    # B=4096 2.11 vs 2.08 ms, B=16384 8.2 vs 3.4 ms; (8,2) T=10 crosses near B=5000)
    QUAD_MIN_BATCH = 4096

    def __init__(self):
        self.lib = _lib.load()
        self._ws = {}  # (device, dtype) -> scratch tensor for the quad variant (grown on demand)
        if os.environ.get("ALQP_QUAD_STAGGER") is not None:   # -1 auto (default), 0 off, > 0 units of ~1024 clocks
            self.set_quad_stagger(int(os.environ["ALQP_QUAD_STAGGER"]))

    quad_stagger = -1   # -1 automatic, 0 off, > 0 explicit units of ~1024 clocks (passed per call in AlqpParams.quad_stagger)

    def set_quad_stagger(self, mode):
        """Start offset between the four wavefronts of a CU in the quad solve (AlqpParams.quad_stagger, include/mi_alqp.h):
        -1 automatic, 0 off, > 0 explicit. Returns the previous mode. Timing only - results do not depend on it.
        State of THIS backend object; the library keeps none."""
        prev, self.quad_stagger = self.quad_stagger, (-1 if int(mode) < 0 else int(mode))
        return prev

    def workspace_bytes(self, B, T, nx, nu, dtype):
        d = _lib.AlqpDims(B, T, nx, nu)
        return int(self.lib.alqp_workspace_bytes(C.byref(d), int(dtype == torch.float64)))

    def _workspace(self, dims, like):
        """Device scratch for the quad variant, cached per (device, dtype). Contents need
        not survive between calls; stream order protects reuse on one stream."""
        need = self.workspace_bytes(*dims, like.dtype)
        key = (like.device, like.dtype)
        ws = self._ws.get(key)
        if ws is None or ws.numel() * ws.element_size() < need:
            ws = torch.empty(need // like.element_size() + 16, dtype=like.dtype, device=like.device)
            self._ws[key] = ws
        return ws, need

    # -- queries ---------------------------------------------------------------
    def supported(self, B, T, nx, nu, dtype):
        d = _lib.AlqpDims(B, T, nx, nu)
        return bool(self.lib.alqp_supported(C.byref(d), int(dtype == torch.float64)))

    def lds_bytes(self, B, T, nx, nu, dtype):
        d = _lib.AlqpDims(B, T, nx, nu)
        return int(self.lib.alqp_lds_bytes(C.byref(d), int(dtype == torch.float64)))

    def qps_per_wave(self, B, T, nx, nu, dtype):
        d = _lib.AlqpDims(B, T, nx, nu)
        return int(self.lib.alqp_qps_per_wave(C.byref(d), int(dtype == torch.float64)))

    # -- kernels -----------------------------------------------------------------
    def solve_lin(self, dims, Qd, q, F, c, x0, ulo, uhi, sb_u, st_u, z, lam, rho, phi,
                  rnorm2=None, info=None, status=None, factor=None, al_iter=2, max_newton=4,
                  n_ls=20, flags=_lib.ALQP_INIT_MERIT | _lib.ALQP_DUAL_UPDATE, rho_scale=10.0,
                  trace=None, variant=None, workspace=None, skip=None, newton_counts=None, exit_tol=1e-3):
        """variant: None/"auto" (quad unless a factor must be saved), "team", "quad".
        workspace: a dedicated scratch tensor for the quad variant (kept by the caller when
        the factor it holds afterwards is needed for `backward_ws`); default: a cached one.
        newton_counts (int32 [al_iter], device): with it the reference's batch-global exit test of the Newton loop
        runs INSIDE the launch (ALQP_EXIT_IN_KERNEL, one cooperative launch); returns False - nothing launched - when
        the grid cannot be co-resident (the caller then takes the launch-per-step route), True otherwise."""
        B, T, nx, nu = dims
        dt = z.dtype
        sfx = _dt(z)
        d = _lib.AlqpDims(B, T, nx, nu)
        if variant is None:
            variant = self.default_variant
        vnum = {"auto": 0, "team": 1, "quad": 2}[variant]
        if vnum == 0:
            # the fused solve stays on the team kernels THROUGH B = 4096 (exactly two fp32 / four fp64 full rounds of
            # teams; round 3: fp32 1.83 against 2.00 ms, fp64 4.35 against 4.71 ms there, quad ahead from B = 5120 on)
            # (whole-wavefront teams only - 2n + nx + 1 > 32 rows, where it was measured; smaller teams keep round 2's rule)
            if self.QUAD_MIN_BATCH != 4096:
                qmin = self.QUAD_MIN_BATCH
            elif 2 * (nx + nu) + nx + 1 > 32:
                qmin = 4097
            else:
                qmin = 4608 if dt == torch.float64 else 4096
            vnum = 1 if ((flags & _lib.ALQP_SAVE_FACTOR) or B < qmin) else 2
            if vnum == 1 and not (flags & _lib.ALQP_SAVE_FACTOR) and not self.lib.alqp_supported_variant(C.byref(d), int(dt == torch.float64), 1):
                vnum = 2   # horizon too long for the team's LDS image: the quad kernels run it at any batch
        ws, ws_bytes = (None, 0)
        if vnum == 2:
            if workspace is not None:
                ws, ws_bytes = workspace, self.workspace_bytes(*dims, z.dtype)
                if ws.numel() * ws.element_size() < ws_bytes:
                    raise ValueError("mi_alqp: workspace too small")
            else:
                ws, ws_bytes = self._workspace(dims, z)
        self.last_variant = "quad" if vnum == 2 else "team"
        skp = _ptr(skip, "skip", torch.float64, True)
        p = _lib.AlqpParams(al_iter, max_newton, n_ls, flags | _DEBUG_FLAGS, rho_scale, vnum, skp.value if skp is not None else None)
        p.quad_stagger = {-1: 0, 0: -1}.get(self.quad_stagger, self.quad_stagger)   # ABI: 0 automatic, < 0 off
        if newton_counts is not None:
            self._exit_in_kernel(p, B, z.device, newton_counts, exit_tol)
        tr = None
        if trace is not None:
            tr = _lib.AlqpTrace(*[
                (trace[k].data_ptr() if trace.get(k) is not None else None)
                for k in ("g", "d", "phi", "phi_prev", "k", "accept")])
        fn = getattr(self.lib, "alqp_solve_lin_" + sfx)
        rc = fn(C.byref(d), C.byref(p), _ptr(Qd, "Qd", dt), _ptr(q, "q", dt), _ptr(F, "F", dt),
                _ptr(c, "c", dt), _ptr(x0, "x0", dt), _ptr(ulo, "u_lower", dt), _ptr(uhi, "u_upper", dt),
                sb_u, st_u, _ptr(z, "z", dt), _ptr(lam, "lam", dt), _ptr(rho, "rho", dt),
                _ptr(phi, "phi", dt), _ptr(rnorm2, "rnorm2", dt, True),
                _ptr(info, "info", torch.int32, True), _ptr(status, "status", torch.uint8, True),
                _ptr(factor, "factor", dt, True), C.byref(tr) if tr is not None else None,
                _ptr(ws, "workspace", dt, True), ws_bytes, _stream())
        if rc == _lib.ALQP_E_COOP and newton_counts is not None:
            return False
        _lib.check(rc, "alqp_solve_lin_" + sfx)
        return True

    def _exit_in_kernel(self, p, B, device, newton_counts, exit_tol):
        """AlqpParams fields of ALQP_EXIT_IN_KERNEL: the cached scratch (arrival counter + 2 x B partial sums)."""
        key = ("exit", device)
        scr = self._ws.get(key)
        if scr is None or scr.numel() < 2 * B + 2:
            scr = torch.zeros(2 * B + 2, dtype=torch.float64, device=device)
            self._ws[key] = scr
        scr.zero_()   # arrival counter, partial sums, time-out flag
        p.flags |= _lib.ALQP_EXIT_IN_KERNEL
        p.exit_tol = float(exit_tol)
        p.newton_counts = _ptr(newton_counts, "newton_counts", torch.int32).value
        p.exit_scratch = scr.data_ptr()

    def solve_nonlin(self, dims, dyn_id, dyn_h, Qd, q, x0, ulo, uhi, sb_u, st_u, z, lam, rho, phi, rnorm2=None,
                     info=None, status=None, al_iter=2, max_newton=4,
                     flags=_lib.ALQP_INIT_MERIT | _lib.ALQP_DUAL_UPDATE, rho_scale=10.0, skip=None, workspace=None,
                     newton_counts=None, exit_tol=1e-3):
        """Nonlinear fused solve (alqp_solve_nonlin): the dynamics model `dyn_id` is inlined.
        workspace: a private one (new_workspace_nonlin) when the factor and linearisation it holds
        afterwards are needed by backward_ws; default: a cached one."""
        B, T, nx, nu = dims
        dt = z.dtype
        sfx = _dt(z)
        d = _lib.AlqpDims(B, T, nx, nu)
        need = int(self.lib.alqp_workspace_bytes_nonlin(C.byref(d), int(dt == torch.float64)))
        if need == 0:
            raise RuntimeError("mi_alqp: nonlinear fused solve not available for these sizes")
        if workspace is not None:
            ws = workspace
            if ws.numel() * ws.element_size() < need:
                raise ValueError("mi_alqp: workspace too small")
        else:
            key = ("nl", z.device, dt)
            ws = self._ws.get(key)
            if ws is None or ws.numel() * ws.element_size() < need:
                ws = torch.empty(need // z.element_size() + 16, dtype=dt, device=z.device)
                self._ws[key] = ws
        skp = _ptr(skip, "skip", torch.float64, True)
        p = _lib.AlqpParams(al_iter, max_newton, 20, flags, rho_scale, 2, skp.value if skp is not None else None)
        if newton_counts is not None:   # the reference's exit test inside one cooperative launch, see solve_lin
            self._exit_in_kernel(p, B, z.device, newton_counts, exit_tol)
        fn = getattr(self.lib, "alqp_solve_nonlin_" + sfx)
        rc = fn(C.byref(d), C.byref(p), int(dyn_id), float(dyn_h), _ptr(Qd, "Qd", dt), _ptr(q, "q", dt), _ptr(x0, "x0", dt),
                _ptr(ulo, "u_lower", dt), _ptr(uhi, "u_upper", dt), sb_u, st_u, _ptr(z, "z", dt), _ptr(lam, "lam", dt),
                _ptr(rho, "rho", dt), _ptr(phi, "phi", dt), _ptr(rnorm2, "rnorm2", dt, True),
                _ptr(info, "info", torch.int32, True), _ptr(status, "status", torch.uint8, True),
                _ptr(ws, "workspace", dt), need, _stream())
        if rc == _lib.ALQP_E_COOP and newton_counts is not None:
            return False
        _lib.check(rc, "alqp_solve_nonlin_" + sfx)
        self.last_variant = "quad"
        return True

    def dyn_pendulum1l(self, x, u, h, want_jac=True):
        """pendulum1l provider (alqp_dyn_pendulum1l): x [K,2], u [K,1], h float or [K(,1)] tensor
        -> xnext [K,2], F [K,2,3] or None."""
        K = x.shape[0]
        dt = x.dtype
        xn = torch.empty(K, 2, dtype=dt, device=x.device)
        F = torch.empty(K, 2, 3, dtype=dt, device=x.device) if want_jac else None
        hpt = h.to(dt).reshape(-1).contiguous() if torch.is_tensor(h) else None
        if hpt is not None and hpt.numel() != K:
            raise ValueError("mi_alqp: h must be a number or one value per point")
        fn = getattr(self.lib, "alqp_dyn_pendulum1l_" + _dt(x))
        rc = fn(K, _ptr(x, "x", dt), _ptr(u, "u", dt), 0.0 if hpt is not None else float(h), _ptr(hpt, "h", dt, True),
                _ptr(xn, "xnext", dt), _ptr(F, "F", dt, True), _stream())
        _lib.check(rc, "alqp_dyn_pendulum1l")
        return xn, F

    def dyn_cartpole1l(self, x, tau, h, want_jac=True, version=1):
        """cartpole1l provider (alqp_dyn_cartpole1l; version=2: alqp_dyn_cartpole1l_v2, the cartpole1l_v2 package's
        constants): x [K,4], tau [K,2], h float or per-point tensor -> xnext [K,4], J [K,4,6] (d xnext / d(q, qdot, tau))
        or None."""
        K = x.shape[0]
        dt = x.dtype
        xn = torch.empty(K, 4, dtype=dt, device=x.device)
        J = torch.empty(K, 4, 6, dtype=dt, device=x.device) if want_jac else None
        hpt = h.to(dt).reshape(-1).contiguous() if torch.is_tensor(h) else None
        if hpt is not None and hpt.numel() != K:
            raise ValueError("mi_alqp: h must be a number or one value per point")
        fn = getattr(self.lib, ("alqp_dyn_cartpole1l_v2_" if version == 2 else "alqp_dyn_cartpole1l_") + _dt(x))
        rc = fn(K, _ptr(x, "x", dt), _ptr(tau, "tau", dt), 0.0 if hpt is not None else float(h), _ptr(hpt, "h", dt, True),
                _ptr(xn, "xnext", dt), _ptr(J, "J", dt, True), _stream())
        _lib.check(rc, "alqp_dyn_cartpole1l")
        return xn, J

    def dyn_cartpole2l(self, x, tau, h, want_jac=True):
        """cartpole2l provider (alqp_dyn_cartpole2l): x [K,6], tau [K,3] -> xnext [K,6], J [K,6,9] or None."""
        K = x.shape[0]
        dt = x.dtype
        xn = torch.empty(K, 6, dtype=dt, device=x.device)
        J = torch.empty(K, 6, 9, dtype=dt, device=x.device) if want_jac else None
        hpt = h.to(dt).reshape(-1).contiguous() if torch.is_tensor(h) else None
        if hpt is not None and hpt.numel() != K:
            raise ValueError("mi_alqp: h must be a number or one value per point")
        fn = getattr(self.lib, "alqp_dyn_cartpole2l_" + _dt(x))
        rc = fn(K, _ptr(x, "x", dt), _ptr(tau, "tau", dt), 0.0 if hpt is not None else float(h), _ptr(hpt, "h", dt, True),
                _ptr(xn, "xnext", dt), _ptr(J, "J", dt, True), _stream())
        _lib.check(rc, "alqp_dyn_cartpole2l")
        return xn, J

    def dyn_rigid(self, model, params, x, u, h, want_jac=True):
        """Quadrotor / flying-cartpole provider (alqp_dyn_rexquadrotor / alqp_dyn_flyingcartpole): model "rex" (x [K,12])
        or "flycart" (x [K,14]), u [K,4], params an _lib.AlqpRigidParams -> xnext [K,nx], F = [A | B] [K,nx,nx+4] or None."""
        K, nx = x.shape
        dt = x.dtype
        xn = torch.empty(K, nx, dtype=dt, device=x.device)
        F = torch.empty(K, nx, nx + 4, dtype=dt, device=x.device) if want_jac else None
        name = {"rex": "alqp_dyn_rexquadrotor_", "flycart": "alqp_dyn_flyingcartpole_"}[model]
        if nx != {"rex": 12, "flycart": 14}[model] or u.shape != (K, 4):
            raise ValueError(f"mi_alqp: {model} dynamics take x [K,{ {'rex': 12, 'flycart': 14}[model] }] and u [K,4]")
        rc = getattr(self.lib, name + _dt(x))(K, C.byref(params), _ptr(x, "x", dt), _ptr(u, "u", dt), float(h),
                                             _ptr(xn, "xnext", dt), _ptr(F, "F", dt, True), _stream())
        _lib.check(rc, name)
        return xn, F

    def exit_test(self, sumsq, ctl, mode, tol=1e-3):
        """Device-side batch-global exit test (alqp_exit_test): sumsq 0-d/1-elem float64 tensor,
        ctl float64[3] = {done, steps, old_norm}; nothing is synchronised."""
        rc = self.lib.alqp_exit_test(_ptr(sumsq, "sumsq", torch.float64), _ptr(ctl, "ctl", torch.float64),
                                     int(mode), float(tol), _stream())
        _lib.check(rc, "alqp_exit_test")

    @staticmethod
    def _obs_struct(obs, dims, dt):
        """obs = (pos [B,T,nobs,3] tensor, radius float) -> AlqpObstacles (Obstacle_MPC rows)."""
        if isinstance(obs, str) and obs == "state_estimator":     # that variant's row set and gradient, no obstacle rows
            return _lib.AlqpObstacles(None, 0.0, 0, 1)
        pos, radius = obs
        B, T = dims[0], dims[1]
        if pos.dim() != 4 or pos.shape[0] != B or pos.shape[1] != T or pos.shape[3] != 3:
            raise ValueError(f"mi_alqp: obstacle centres must be [B,T,nobs,3], got {tuple(pos.shape)}")
        return _lib.AlqpObstacles(_ptr(pos, "obstacle centres", dt).value, float(radius), int(pos.shape[2]), 0)

    def newton_step(self, dims, z, xnext, F, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, d_out,
                    g_out=None, factor=None, info=None, obs=None, workspace=None):
        """workspace: a quad-variant workspace tensor -> alqp_newton_step_ws (16 instances per wavefront,
        the factor stays in the workspace records for backward_ws); None -> the team kernel
        (one instance per lane team, optional packed `factor`)."""
        B, T, nx, nu = dims
        dt = z.dtype
        sfx = _dt(z)
        d = _lib.AlqpDims(B, T, nx, nu)
        if workspace is not None:
            if factor is not None:
                raise ValueError("mi_alqp: the quad Newton step leaves its factor in the workspace records (no packed factor)")
            need = self.workspace_bytes(*dims, dt)
            if workspace.numel() * workspace.element_size() < need:
                raise ValueError("mi_alqp: workspace too small")
            head = (C.byref(d), _ptr(z, "z", dt), _ptr(xnext, "xnext", dt), _ptr(F, "F", dt), _ptr(x0, "x0", dt),
                    _ptr(lam, "lam", dt), _ptr(rho, "rho", dt), _ptr(Qd, "Qd", dt), _ptr(q, "q", dt),
                    _ptr(ulo, "u_lower", dt), _ptr(uhi, "u_upper", dt), sb_u, st_u)
            tail = (_ptr(workspace, "workspace", dt), need, _ptr(d_out, "d_out", dt), _ptr(g_out, "g_out", dt, True),
                    _ptr(info, "info", torch.int32, True), _stream())
            if obs is None:
                rc = getattr(self.lib, "alqp_newton_step_ws_" + sfx)(*head, *tail)
            else:   # obstacle rows / the state-estimator row set on the quad kernels
                o = self._obs_struct(obs, dims, dt)
                rc = getattr(self.lib, "alqp_newton_step_ws_obs_" + sfx)(*head, C.byref(o), *tail)
            _lib.check(rc, "alqp_newton_step_ws_" + sfx)
            self.last_step_kernel = "k_newton_step_quad"
            return
        self.last_step_kernel = "k_newton_step"
        head = (C.byref(d), _ptr(z, "z", dt), _ptr(xnext, "xnext", dt), _ptr(F, "F", dt),
                _ptr(x0, "x0", dt), _ptr(lam, "lam", dt), _ptr(rho, "rho", dt), _ptr(Qd, "Qd", dt),
                _ptr(q, "q", dt), _ptr(ulo, "u_lower", dt), _ptr(uhi, "u_upper", dt), sb_u, st_u)
        tail = (_ptr(d_out, "d_out", dt), _ptr(g_out, "g_out", dt, True),
                _ptr(factor, "factor", dt, True), _ptr(info, "info", torch.int32, True), _stream())
        if obs is None:
            rc = getattr(self.lib, "alqp_newton_step_" + sfx)(*head, *tail)
        else:
            o = self._obs_struct(obs, dims, dt)
            rc = getattr(self.lib, "alqp_newton_step_obs_" + sfx)(*head, C.byref(o), *tail)
        _lib.check(rc, "alqp_newton_step_" + sfx)

    def merit(self, dims, K, zc, xnext, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, phi, rnorm2=None, obs=None):
        B, T, nx, nu = dims
        dt = zc.dtype
        sfx = _dt(zc)
        d = _lib.AlqpDims(B, T, nx, nu)
        head = (C.byref(d), K, _ptr(zc, "zc", dt), _ptr(xnext, "xnext", dt), _ptr(x0, "x0", dt),
                _ptr(lam, "lam", dt), _ptr(rho, "rho", dt), _ptr(Qd, "Qd", dt), _ptr(q, "q", dt),
                _ptr(ulo, "u_lower", dt), _ptr(uhi, "u_upper", dt), sb_u, st_u)
        tail = (_ptr(phi, "phi", dt), _ptr(rnorm2, "rnorm2", dt, True), _stream())
        if obs is None:
            rc = getattr(self.lib, "alqp_merit_" + sfx)(*head, *tail)
        else:
            o = self._obs_struct(obs, dims, dt)
            rc = getattr(self.lib, "alqp_merit_obs_" + sfx)(*head, C.byref(o), *tail)
        _lib.check(rc, "alqp_merit_" + sfx)

    def merit_pick(self, dims, n_ls, d, xnext_all, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, z, phi_prev,
                   rnorm2=None, phi_all=None, k_out=None, accept_out=None, obs=None):
        """The line search of one Newton step in one launch (alqp_merit_pick): merits of z + 2^-k d from the
        caller's x_next of every candidate, decision, z and phi_prev updated in place."""
        B, T, nx, nu = dims
        dt = z.dtype
        sfx = _dt(z)
        dd = _lib.AlqpDims(B, T, nx, nu)
        o = self._obs_struct(obs, dims, dt) if obs is not None else None
        rc = getattr(self.lib, "alqp_merit_pick_" + sfx)(
            C.byref(dd), n_ls, _ptr(d, "d", dt), _ptr(xnext_all, "xnext_all", dt), _ptr(x0, "x0", dt),
            _ptr(lam, "lam", dt), _ptr(rho, "rho", dt), _ptr(Qd, "Qd", dt), _ptr(q, "q", dt),
            _ptr(ulo, "u_lower", dt), _ptr(uhi, "u_upper", dt), sb_u, st_u, C.byref(o) if o is not None else None,
            _ptr(z, "z", dt), _ptr(phi_prev, "phi_prev", dt), _ptr(rnorm2, "rnorm2", dt, True),
            _ptr(phi_all, "phi_all", dt, True), _ptr(k_out, "k_out", torch.int32, True),
            _ptr(accept_out, "accept_out", torch.int32, True), _stream())
        _lib.check(rc, "alqp_merit_pick_" + sfx)

    def linesearch_pick(self, dims, n_ls, phi, phi_prev, d, z, k_out=None, accept_out=None):
        B, T, nx, nu = dims
        dt = z.dtype
        sfx = _dt(z)
        dd = _lib.AlqpDims(B, T, nx, nu)
        fn = getattr(self.lib, "alqp_linesearch_pick_" + sfx)
        rc = fn(C.byref(dd), n_ls, _ptr(phi, "phi", dt), _ptr(phi_prev, "phi_prev", dt),
                _ptr(d, "d", dt), _ptr(z, "z", dt), _ptr(k_out, "k_out", torch.int32, True),
                _ptr(accept_out, "accept_out", torch.int32, True), _stream())
        _lib.check(rc, "alqp_linesearch_pick_" + sfx)

    def dual_update(self, dims, z, xnext, x0, ulo, uhi, sb_u, st_u, lam, rho, rho_scale=10.0, obs=None):
        B, T, nx, nu = dims
        dt = z.dtype
        sfx = _dt(z)
        d = _lib.AlqpDims(B, T, nx, nu)
        head = (C.byref(d), _ptr(z, "z", dt), _ptr(xnext, "xnext", dt), _ptr(x0, "x0", dt),
                _ptr(ulo, "u_lower", dt), _ptr(uhi, "u_upper", dt), sb_u, st_u)
        tail = (_ptr(lam, "lam", dt), _ptr(rho, "rho", dt), rho_scale, _stream())
        if obs is None:
            rc = getattr(self.lib, "alqp_dual_update_" + sfx)(*head, *tail)
        else:
            o = self._obs_struct(obs, dims, dt)
            rc = getattr(self.lib, "alqp_dual_update_obs_" + sfx)(*head, C.byref(o), *tail)
        _lib.check(rc, "alqp_dual_update_" + sfx)

    def backward(self, dims, factor, F, rho, z_final, gbar, q_grad, Qd_grad):
        B, T, nx, nu = dims
        dt = gbar.dtype
        sfx = _dt(gbar)
        d = _lib.AlqpDims(B, T, nx, nu)
        fn = getattr(self.lib, "alqp_backward_" + sfx)
        rc = fn(C.byref(d), _ptr(factor, "factor", dt), _ptr(F, "F", dt), _ptr(rho, "rho", dt),
                _ptr(z_final, "z_final", dt), _ptr(gbar, "gbar", dt), _ptr(q_grad, "q_grad", dt),
                _ptr(Qd_grad, "Qd_grad", dt), _stream())
        _lib.check(rc, "alqp_backward_" + sfx)


    def new_workspace_nonlin(self, dims, like):
        """A private workspace for nonlinear fused solves: [records | F linearisations]."""
        B, T, nx, nu = dims
        d = _lib.AlqpDims(B, T, nx, nu)
        need = int(self.lib.alqp_workspace_bytes_nonlin(C.byref(d), int(like.dtype == torch.float64)))
        return torch.empty(need // like.element_size() + 16, dtype=like.dtype, device=like.device)

    def nonlin_F_view(self, ws, dims):
        """The F region of a nonlinear workspace as a [B, T-1, nx, n] tensor (the linearisation of the
        last executed Newton step, next to its factor in the records)."""
        B, T, nx, nu = dims
        off = self.workspace_bytes(B, T, nx, nu, ws.dtype) // ws.element_size()
        return ws[off:off + B * (T - 1) * nx * (nx + nu)].view(B, T - 1, nx, nx + nu)

    def new_workspace(self, dims, like):
        """A private workspace for one quad solve whose factor will be used by backward_ws."""
        need = self.workspace_bytes(*dims, like.dtype)
        return torch.empty(need // like.element_size() + 16, dtype=like.dtype, device=like.device)

    def backward_ws(self, dims, workspace, F, rho, z_final, gbar, q_grad, Qd_grad):
        B, T, nx, nu = dims
        dt = gbar.dtype
        sfx = _dt(gbar)
        d = _lib.AlqpDims(B, T, nx, nu)
        fn = getattr(self.lib, "alqp_backward_ws_" + sfx)
        rc = fn(C.byref(d), _ptr(workspace, "workspace", dt), self.workspace_bytes(*dims, dt),
                _ptr(F, "F", dt), _ptr(rho, "rho", dt), _ptr(z_final, "z_final", dt),
                _ptr(gbar, "gbar", dt), _ptr(q_grad, "q_grad", dt), _ptr(Qd_grad, "Qd_grad", dt), _stream())
        _lib.check(rc, "alqp_backward_ws_" + sfx)


    # ---- interior-point path (csrc/alqp_ipm.hip) -------------------------------------------------
    def _ipm_ws(self, dims, like):
        d = _lib.AlqpDims(*dims)
        need = int(self.lib.alqp_ipm_workspace_bytes(C.byref(d), int(like.dtype == torch.float64)))
        if need == 0:
            raise RuntimeError(f"mi_alqp: no interior-point kernel instance for (nx={dims[2]}, nu={dims[3]})")
        key = ("ipm", like.device, like.dtype)
        ws = self._ws.get(key)
        if ws is None or ws.numel() * ws.element_size() < need:
            ws = torch.empty(need // like.element_size() + 16, dtype=like.dtype, device=like.device)
            self._ws[key] = ws
        return ws, need

    ipm_variant = "auto"   # default kernel of ipm_solve / ipm_backward: a key of _lib.IPM_VARIANTS (include/mi_alqp.h)

    def ipm_solve(self, dims, Cd, c, F, f, x0, uhi, ulo, exit_mode="reference", eps=1e-12, not_improved_lim=3,
                  max_iter=20, ry_fn=None, kkt_eps=1e-7, process_group=None, sharded=False, variant=None):
        """pdipm_b_LU.forward (batch_LU.py:29-197) on time-major data: Cd, c [T,B,n]; F [T-1,B,nx,n];
        f [T-1,B,nx]; x0 [B,nx]; uhi, ulo [nu]. ry_fn(z [B,T*n]) -> [B,T*nx]: equality residual of the TRUE
        dynamics (one launch per iteration, PyTorch call in between), or None for A z - b.
        exit_mode "fixed" (and no ry_fn): ONE launch for the whole solve. "reference": the reference's
        batch-global exit rule, one host read per iteration (the reference syncs there as well); with `sharded`
        the three batch-global quantities are max-reduced over the ranks of `process_group` first.
        variant: "auto" (the register/LDS-resident kernel when T <= 20, else the generic one), "generic_lds",
        "generic_ws", "resident" (AlqpIpmParams.variant); None: self.ipm_variant."""
        vnum = _lib.IPM_VARIANTS[variant or self.ipm_variant]
        B, T, nx, nu = dims
        n = nx + nu
        dt, dev = c.dtype, c.device
        sfx = _dt(c)
        d = _lib.AlqpDims(B, T, nx, nu)
        ws, need = self._ipm_ws(dims, c)
        out = {"zhat": torch.empty(B, T * n, dtype=dt, device=dev), "nus": torch.empty(B, T * nx, dtype=dt, device=dev),
               "lams": torch.empty(B, 2 * T * nu, dtype=dt, device=dev),
               "slacks": torch.empty(B, 2 * T * nu, dtype=dt, device=dev),
               "resid": torch.empty(B, dtype=dt, device=dev), "info": torch.zeros(B, dtype=torch.int32, device=dev)}
        mu = torch.empty(B, dtype=dt, device=dev)
        improved = torch.zeros(B, dtype=torch.int32, device=dev)
        fn = getattr(self.lib, "alqp_ipm_solve_" + sfx)
        ptrs = (_ptr(Cd, "Cd", dt), _ptr(c, "c", dt), _ptr(F, "F", dt), _ptr(f, "f", dt), _ptr(x0, "x0", dt),
                _ptr(uhi, "u_upper", dt), _ptr(ulo, "u_lower", dt), B * n, n, B * nx * n, nx * n, B * nx, nx)

        def launch(flags, iters=0, it0=0, ry=None):
            p = _lib.AlqpIpmParams(flags, iters, it0, kkt_eps, vnum)
            rc = fn(C.byref(d), C.byref(p), *ptrs, _ptr(ws, "workspace", dt), need, _ptr(ry, "ry", dt, True),
                    _ptr(out["zhat"], "zhat", dt), _ptr(out["nus"], "nus", dt), _ptr(out["lams"], "lams", dt),
                    _ptr(out["slacks"], "slacks", dt), _ptr(out["resid"], "resid", dt), _ptr(mu, "mu", dt), None,
                    _ptr(improved, "improved", torch.int32), _ptr(out["info"], "info", torch.int32), _stream())
            _lib.check(rc, "alqp_ipm_solve_" + sfx)

        if exit_mode == "fixed" and ry_fn is None:
            launch(_lib.ALQP_IPM_INIT | _lib.ALQP_IPM_LOOP | _lib.ALQP_IPM_FINAL, max_iter)
            out["iters"] = max_iter
            return out
        launch(_lib.ALQP_IPM_INIT)
        ws_words = need // c.element_size() // B
        cur_x = ws[:B * ws_words].view(B, ws_words)[:, :T * n]      # the iterate's x block of every instance
        n_not_improved, done = 0, max_iter
        for it in range(max_iter):
            ry = ry_fn(cur_x.contiguous()).to(dt).contiguous() if ry_fn is not None else None
            launch(_lib.ALQP_IPM_RESID, 0, it, ry)
            if exit_mode == "reference":
                # batch_LU.py:120-151: nNotImproved counts iterations in which NO instance improved
                glob = torch.stack((improved.max().to(torch.float64), out["resid"].max().to(torch.float64),
                                    -mu.min().to(torch.float64)))
                if sharded:
                    torch.distributed.all_reduce(glob, op=torch.distributed.ReduceOp.MAX, group=process_group)
                any_imp, best_max, mu_min = glob.tolist()
                mu_min = -mu_min
                n_not_improved = 0 if (it == 0 or any_imp > 0) else n_not_improved + 1
                if n_not_improved == not_improved_lim or best_max < eps or mu_min > 1e32:
                    done = it
                    break
            launch(_lib.ALQP_IPM_STEP, 0, it)
        launch(_lib.ALQP_IPM_FINAL)
        out["iters"] = done
        return out

    def ipm_backward(self, dims, Cd, F, lams, slacks, g, variant=None):
        """DenseQPFunction.backward's KKT solve (qp.py:243-252) -> dx [B,T*n], dlam [B,2*T*nu], dnu [B,T*nx]."""
        vnum = _lib.IPM_VARIANTS[variant or self.ipm_variant]
        B, T, nx, nu = dims
        n = nx + nu
        dt, dev = g.dtype, g.device
        sfx = _dt(g)
        d = _lib.AlqpDims(B, T, nx, nu)
        ws, need = self._ipm_ws(dims, g)
        dx = torch.empty(B, T * n, dtype=dt, device=dev)
        dlam = torch.empty(B, 2 * T * nu, dtype=dt, device=dev)
        dnu = torch.empty(B, T * nx, dtype=dt, device=dev)
        fn = getattr(self.lib, "alqp_ipm_backward_" + sfx)
        rc = fn(C.byref(d), _ptr(Cd, "Cd", dt), _ptr(F, "F", dt), B * n, n, B * nx * n, nx * n, _ptr(lams, "lams", dt),
                _ptr(slacks, "slacks", dt), _ptr(g, "gbar", dt), _ptr(ws, "workspace", dt), need, _ptr(dx, "dx", dt),
                _ptr(dlam, "dlam", dt), _ptr(dnu, "dnu", dt), None, vnum, _stream())
        _lib.check(rc, "alqp_ipm_backward_" + sfx)
        return dx, dlam, dnu


_default = None


def default_backend():
    global _default
    if _default is None:
        _default = HipBackend()
    return _default
