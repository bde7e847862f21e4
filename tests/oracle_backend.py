"""TEST-ONLY backend: the solver's backend interface implemented on CPU tensors with
the oracle's building blocks (oracle/alqp_oracle.c). It lets the host logic of the
drop-in MPC class (state carry, exit modes, dispatch, sharding) run in the CPU
container. It is never importable from the product package."""
import numpy as np
import torch

from oracle import oracle_py as orc

INIT_MERIT, DUAL_UPDATE, SAVE_FACTOR = 1, 2, 4


def _n(t):
    return t.detach().cpu().numpy()


def _sfx(t):
    return "f64" if t.dtype == torch.float64 else "f32"


def _bounds(ulo, uhi, sb, st, B, T, nu):
    lo, hi = _n(ulo), _n(uhi)
    if sb == 0 and st == 0:
        return lo.reshape(nu), hi.reshape(nu)
    return lo.reshape(B, T, nu), hi.reshape(B, T, nu)


class OracleBackend:
    name = "oracle-test"

    def supported(self, B, T, nx, nu, dtype):
        return True

    def exit_test(self, sumsq, ctl, mode, tol=1e-3):
        """CPU twin of alqp_exit_test (include/mi_alqp.h)."""
        import math
        nw = math.sqrt(float(sumsq[0]))
        if mode == 0:
            ctl[0], ctl[1], ctl[2] = 0.0, 0.0, nw
        elif float(ctl[0]) == 0.0:
            ctl[1] += 1.0
            old = float(ctl[2])
            if nw < tol or (nw == nw and abs(old - nw) / nw < tol if nw != 0 else False):
                ctl[0] = 1.0
            else:
                ctl[2] = nw

    def solve_lin(self, dims, Qd, q, F, c, x0, ulo, uhi, sb_u, st_u, z, lam, rho, phi, rnorm2=None,
                  info=None, status=None, factor=None, al_iter=2, max_newton=4, n_ls=20, flags=3,
                  rho_scale=10.0, trace=None, variant=None, skip=None):
        if skip is not None and float(skip[0]) != 0.0:
            return
        B, T, nx, nu = dims
        s = _sfx(z)
        npdt = np.float64 if s == "f64" else np.float32
        lo, hi = _bounds(ulo, uhi, sb_u, st_u, B, T, nu)
        Qd_, q_, F_, c_, x0_ = _n(Qd), _n(q), _n(F), _n(c), _n(x0)
        zz, ll, rr, ph = _n(z).copy(), _n(lam).copy(), _n(rho).copy(), _n(phi).copy()

        def xnext(v):
            return (np.einsum("btij,btj->bti", F_, v[:, :-1]) + c_).astype(npdt)

        L = None
        inf_acc = np.zeros(B, np.int32)
        for _ in range(al_iter):
            if flags & INIT_MERIT:
                ph, _ = orc.merit(s, zz, xnext(zz), x0_, ll, rr, Qd_, q_, lo, hi)
            for _ in range(max_newton):
                g, Hd, Hs = orc.grad_hess(s, zz, xnext(zz), F_, x0_, ll, rr, Qd_, q_, lo, hi)
                d, inf, L, _ = orc.newton_dir(s, g, Hd, Hs, nx, want_factor=True)
                inf_acc = np.where(inf_acc == 0, inf, inf_acc)
                phis = []
                for k in range(n_ls):
                    zc = (zz + npdt(2.0 ** -k) * d).astype(npdt)
                    phis.append(orc.merit(s, zc, xnext(zc), x0_, ll, rr, Qd_, q_, lo, hi)[0])
                kk, acc, pm = orc.linesearch_pick(s, np.stack(phis), ph)
                alpha = np.where(acc > 0, 2.0 ** -kk.astype(np.float64), 0.0).astype(npdt)
                zz = np.where((acc > 0)[:, None, None], zz + alpha[:, None, None] * d, zz).astype(npdt)
                ph = pm
            if flags & DUAL_UPDATE:
                ll, rr = orc.dual_update(s, zz, xnext(zz), x0_, lo, hi, ll, rr)
                if rho_scale != 10.0:
                    rr = rr / 10.0 * rho_scale
        _, rp2 = orc.merit(s, zz, xnext(zz), x0_, ll, rr, Qd_, q_, lo, hi)
        z.copy_(torch.from_numpy(zz)); lam.copy_(torch.from_numpy(ll)); rho.copy_(torch.from_numpy(rr))
        phi.copy_(torch.from_numpy(ph))
        if rnorm2 is not None:
            rnorm2.copy_(torch.from_numpy(rp2))
        if info is not None:   # sticky like the kernels: the first failure of the solve stays
            cur = _n(info)
            info.copy_(torch.from_numpy(np.where(cur == 0, inf_acc, cur).astype(np.int32)))
        if status is not None:
            status.copy_(torch.from_numpy(np.isfinite(zz).all(axis=(1, 2)).astype(np.uint8)))
        if factor is not None and (flags & SAVE_FACTOR) and L is not None:
            factor.copy_(torch.from_numpy(self._pack_X(L)))

    @staticmethod
    def _pack_X(L):
        """L[B,T,n,n] -> packed upper rows of X = L^{-T} (the device factor layout)."""
        B, T, n, _ = L.shape
        X = np.linalg.inv(L.astype(np.float64)).transpose(0, 1, 3, 2)
        out = np.zeros((B, T, n * (n + 1) // 2), L.dtype)
        off = 0
        for i in range(n):
            out[..., off:off + n - i] = X[..., i, i:]
            off += n - i
        return out

    @staticmethod
    def _unpack_L(fac, n):
        B, T, _ = fac.shape
        X = np.zeros((B, T, n, n), np.float64)
        off = 0
        for i in range(n):
            X[..., i, i:] = fac[..., off:off + n - i]
            off += n - i
        return np.linalg.inv(X.transpose(0, 1, 3, 2))

    @staticmethod
    def _obs(s, obs):
        """Oracle context for the obstacle rows of Obstacle_MPC (obs = (centres [B,T,nobs,3], radius))."""
        if isinstance(obs, str) and obs == "state_estimator":
            return orc.state_estimator(s)
        return orc.obstacles(s, None if obs is None else _n(obs[0]), 0.0 if obs is None else obs[1])

    def newton_step(self, dims, z, xnext, F, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, d_out, g_out=None,
                    factor=None, info=None, obs=None, workspace=None):
        B, T, nx, nu = dims
        s = _sfx(z)
        lo, hi = _bounds(ulo, uhi, sb_u, st_u, B, T, nu)
        with self._obs(s, obs):
            g, Hd, Hs = orc.grad_hess(s, _n(z), _n(xnext), _n(F), _n(x0), _n(lam), _n(rho), _n(Qd), _n(q), lo, hi)
        d, inf, L, _ = orc.newton_dir(s, g, Hd, Hs, nx, want_factor=True)
        d_out.copy_(torch.from_numpy(d))
        if g_out is not None:
            g_out.copy_(torch.from_numpy(g))
        if factor is not None:
            factor.copy_(torch.from_numpy(self._pack_X(L)))
        if info is not None:
            cur = _n(info)
            info.copy_(torch.from_numpy(np.where(cur == 0, inf, cur).astype(np.int32)))

    def merit(self, dims, K, zc, xnext, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, phi, rnorm2=None, obs=None):
        B, T, nx, nu = dims
        s = _sfx(zc)
        lo, hi = _bounds(ulo, uhi, sb_u, st_u, B, T, nu)
        zc_, xn_ = _n(zc).reshape(K, B, T, nx + nu), _n(xnext).reshape(K, B, T - 1, nx)
        for k in range(K):
            with self._obs(s, obs):
                p, r2 = orc.merit(s, zc_[k], xn_[k], _n(x0), _n(lam), _n(rho), _n(Qd), _n(q), lo, hi)
            phi.view(K, B)[k].copy_(torch.from_numpy(p))
            if rnorm2 is not None:
                rnorm2.view(K, B)[k].copy_(torch.from_numpy(r2))

    def merit_pick(self, dims, n_ls, d, xnext_all, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, z, phi_prev,
                   rnorm2=None, phi_all=None, k_out=None, accept_out=None, obs=None):
        """CPU twin of alqp_merit_pick: merits of z + 2^-k d, then linesearch_pick."""
        B, T, nx, nu = dims
        alphas = (2.0 ** -torch.arange(n_ls, dtype=z.dtype)).view(n_ls, 1, 1, 1)
        zc = (z.unsqueeze(0) + alphas * d.unsqueeze(0)).contiguous()
        phis = torch.empty(n_ls, B, dtype=z.dtype)
        rn2s = torch.empty(n_ls, B, dtype=z.dtype)
        self.merit(dims, n_ls, zc, xnext_all, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, phis, rn2s, obs=obs)
        kk = torch.zeros(B, dtype=torch.int32)
        acc = torch.zeros(B, dtype=torch.int32)
        self.linesearch_pick(dims, n_ls, phis, phi_prev, d, z, kk, acc)
        if rnorm2 is not None:
            rnorm2.copy_(torch.where(acc.bool(), rn2s.gather(0, kk.long().unsqueeze(0)).squeeze(0), rnorm2))
        if phi_all is not None:
            phi_all.copy_(phis)
        if k_out is not None:
            k_out.copy_(kk)
        if accept_out is not None:
            accept_out.copy_(acc)

    def linesearch_pick(self, dims, n_ls, phi, phi_prev, d, z, k_out=None, accept_out=None):
        s = _sfx(z)
        kk, acc, pm = orc.linesearch_pick(s, _n(phi), _n(phi_prev))
        alpha = torch.from_numpy(np.where(acc > 0, 2.0 ** -kk.astype(np.float64), 0.0)).to(z.dtype)
        z.copy_(torch.where(torch.from_numpy(acc > 0)[:, None, None], z + alpha[:, None, None] * d, z))
        phi_prev.copy_(torch.from_numpy(pm))
        if k_out is not None:
            k_out.copy_(torch.from_numpy(kk))
        if accept_out is not None:
            accept_out.copy_(torch.from_numpy(acc))

    def dual_update(self, dims, z, xnext, x0, ulo, uhi, sb_u, st_u, lam, rho, rho_scale=10.0, obs=None):
        B, T, nx, nu = dims
        s = _sfx(z)
        lo, hi = _bounds(ulo, uhi, sb_u, st_u, B, T, nu)
        with self._obs(s, obs):
            ll, rr = orc.dual_update(s, _n(z), _n(xnext), _n(x0), lo, hi, _n(lam), _n(rho))
        lam.copy_(torch.from_numpy(ll))
        rho.copy_(torch.from_numpy(rr))

    def backward(self, dims, factor, F, rho, z_final, gbar, q_grad, Qd_grad):
        B, T, nx, nu = dims
        s = _sfx(gbar)
        L = self._unpack_L(_n(factor), nx + nu).astype(_n(gbar).dtype)
        qg, Qg = orc.backward(s, L, _n(F), _n(rho), _n(z_final), _n(gbar))
        q_grad.copy_(torch.from_numpy(qg))
        Qd_grad.copy_(torch.from_numpy(Qg))


    # ---- interior-point path (oracle/ipm_oracle.c): CPU twin of HipBackend.ipm_solve / ipm_backward ----
    def ipm_solve(self, dims, Cd, c, F, f, x0, uhi, ulo, exit_mode="reference", eps=1e-12, not_improved_lim=3,
                  max_iter=20, ry_fn=None, process_group=None, sharded=False, variant=None):
        if sharded and exit_mode == "reference":
            raise NotImplementedError("the oracle's exit rule runs inside its C loop: sharded batches need the product backend")
        # (exit mode "fixed" takes no batch-global decision inside the interior-point iteration: a shard is a batch)
        from oracle import ipm_py
        B, T, nx, nu = dims
        s = _sfx(c)
        tm = lambda a: np.ascontiguousarray(np.swapaxes(_n(a), 0, 1))   # time-major -> batch-major
        cb = None
        if ry_fn is not None:
            cb = lambda x: _n(ry_fn(torch.from_numpy(x).to(c.dtype)))
        o = ipm_py.forward(s, tm(Cd), tm(c), tm(F), tm(f), _n(x0), _n(uhi), _n(ulo), solver=0,
                           exit_mode=0 if exit_mode == "reference" else 1, eps=eps,
                           not_improved_lim=not_improved_lim, max_iter=max_iter, ry_fn=cb)
        out = {k: torch.from_numpy(o[k]) for k in ("zhat", "nus", "lams", "slacks", "resid", "info")}
        out["iters"] = o["iters"]
        return out

    def ipm_backward(self, dims, Cd, F, lams, slacks, g, variant=None):
        from oracle import ipm_py
        s = _sfx(g)
        tm = lambda a: np.ascontiguousarray(np.swapaxes(_n(a), 0, 1))
        dx, dlam, dnu = ipm_py.backward(s, tm(Cd), tm(F), _n(lams), _n(slacks), _n(g), solver=0)
        return torch.from_numpy(dx), torch.from_numpy(dlam), torch.from_numpy(dnu)
