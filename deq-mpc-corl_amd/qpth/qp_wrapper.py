"""Drop-in for the reference's ``qpth.qp_wrapper`` module (the interior-point MPC behind
``--solver_type ip``), MI355X-native.

Surface kept (callers: deqmpc/policies.py:5, 1217-1234, 1265-1283): ``MPC`` with the reference's
constructor keywords (qpth/qp_wrapper.py:121-147) and ``forward(x0, cost, dx, dx_jac, dx_true=None)
-> (x [T,B,nx], u [T,B,nu])`` (:210-293), ``QuadCost(C, c)``, ``LinDx(F, f)``, ``GradMethods``.
Everything is TIME-major ``[T, B, .]`` as in the reference.

What runs where
  * this file: the reference's host logic - initial rollout, (re)linearisation of the dynamics through
    the caller's ``dx`` / ``dx_jac`` (:466-500), the rollout line search (:402-421), the SQP-style outer
    loop (:345-387). Dynamics are arbitrary Python callables, so these steps are PyTorch calls between
    kernel launches, like the nonlinear-caller mode of the AL path;
  * the QP itself - what the reference hands to ``qp.DenseQPFunction`` (qp.py:187-270), i.e. the batched
    primal-dual interior-point method ``pdipm_b_LU.forward`` (solvers/pdipm/batch_LU.py:29-197) with its
    dense ``(nz + 2 nineq + neq)^2`` LU per iteration - runs in the HIP kernels behind
    ``backend.ipm_solve`` / ``ipm_backward`` (csrc/alqp_ipm.hpp): the same regularised KKT system and
    refinement step, solved by eliminating the slack/multiplier rows and a block-tridiagonal Cholesky of
    the Schur complement on the equality multipliers. No dense Q / G / A is ever assembled.
There is no CPU path: without the built extension and a ROCm device it raises.

Scope: diagonal ``C_t`` (what ``policies.Tracking_MPC`` builds, policies.py:1172, 1265); a ``C`` with
off-diagonal entries raises NotImplementedError, as do ``slew_rate_penalty``, ``add_goal_constraint``,
``delta_u`` and ``u_zero_I`` (not reachable from the DEQ-MPC loop).

Exit modes of the interior-point iteration:
  ``"reference"`` the reference's batch-global rule (batch_LU.py:147-151: stop when NO instance of the
                  batch improved three times in a row, or every best residual < eps): one RESID and one
                  STEP launch per iteration with a host read in between - the reference syncs there too;
  ``"fixed"``     all ``maxIter`` (20) iterations in ONE launch, best iterate kept per instance: results do
                  not depend on who else is in the batch (so sharding the batch changes nothing).
"""
from __future__ import annotations

import os
import sys
from collections import namedtuple
from enum import Enum

import torch
from torch.nn import Module

try:
    import deq_mpc_corl_amd  # noqa: F401
except ImportError:  # pragma: no cover
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import deq_mpc_corl_amd  # noqa: F401

QuadCost = namedtuple("QuadCost", "C c")
LinDx = namedtuple("LinDx", "F f")
QuadCost.__new__.__defaults__ = (None,) * len(QuadCost._fields)
LinDx.__new__.__defaults__ = (None,) * len(LinDx._fields)

IPM_EPS = 1e-12            # qp.py:203 (DenseQPFunction defaults)
IPM_NOT_IMPROVED_LIM = 3
IPM_MAX_ITER = 20


class GradMethods(Enum):
    AUTO_DIFF = 1
    FINITE_DIFF = 2
    ANALYTIC = 3
    ANALYTIC_CHECK = 4


def _detach_maybe(t):
    if t is None or not torch.is_tensor(t):
        return t
    return t.detach() if t.requires_grad else t


class _IPQP(torch.autograd.Function):
    """One interior-point QP solve as a differentiable node: qp.DenseQPFunction (qp.py:187-270).

    forward : zhat = argmin 1/2 z'Qz + p'z  s.t. bounds on u, linear(ised) dynamics  -> [B, T*n]
    backward: one KKT solve at the returned iterate (qp.py:238-252), then the reference's gradient
              formulas (:254-268) written on the structured data: dp = dx, dQ = sym(dx zhat'),
              dA = dnu zhat' + nu dx' (only the F entries of A are parameters), db = -dnu
              (b = [-f ; x0]). The bounds were detached by the constructor (qp_wrapper.py:160-164).
    """

    @staticmethod
    def forward(ctx, C, c, F, f, x0, mpc, ry_fn):
        T, B, n = c.shape
        nx = x0.shape[1]
        Cd = C.diagonal(dim1=-2, dim2=-1).contiguous()
        with torch.no_grad():
            out = mpc.backend.ipm_solve((B, T, nx, n - nx), Cd, c.contiguous(), F.contiguous(), f.contiguous(),
                                        x0.contiguous(), mpc._uhi(c), mpc._ulo(c), exit_mode=mpc.exit_mode,
                                        eps=IPM_EPS, not_improved_lim=IPM_NOT_IMPROVED_LIM,
                                        max_iter=IPM_MAX_ITER, ry_fn=ry_fn, **mpc._pg_kwargs())
        mpc.last_ipm = {k: out[k] for k in ("iters", "resid", "info") if k in out}
        ctx.mpc = mpc
        ctx.dims = (B, T, nx, n - nx)
        ctx.save_for_backward(out["zhat"], out["nus"], out["lams"], out["slacks"], Cd, F)
        return out["zhat"]

    @staticmethod
    def backward(ctx, g):
        zhat, nus, lams, slacks, Cd, F = ctx.saved_tensors
        B, T, nx, nu = ctx.dims
        n = nx + nu
        dx, dlam, dnu = ctx.mpc.backend.ipm_backward(ctx.dims, Cd, F, lams, slacks, g.contiguous())
        zt = zhat.view(B, T, n).transpose(0, 1)          # [T,B,n]
        dxt = dx.view(B, T, n).transpose(0, 1)
        dC = 0.5 * (dxt.unsqueeze(-1) * zt.unsqueeze(-2) + zt.unsqueeze(-1) * dxt.unsqueeze(-2))
        dc = dxt.contiguous()
        nu_dyn = nus[:, :(T - 1) * nx].view(B, T - 1, nx).transpose(0, 1)     # [T-1,B,nx]
        dnu_dyn = dnu[:, :(T - 1) * nx].view(B, T - 1, nx).transpose(0, 1)
        dF = dnu_dyn.unsqueeze(-1) * zt[:-1].unsqueeze(-2) + nu_dyn.unsqueeze(-1) * dxt[:-1].unsqueeze(-2)
        df = dnu_dyn.contiguous()                         # b_dyn = -f, db = -dnu
        dx0 = -dnu[:, (T - 1) * nx:]
        return dC, dc, dF, df, dx0, None, None


class MPC(Module):
    """Box-constrained MPC by sequential interior-point QPs (problem statement: qp_wrapper.py:56-66)

        min_{x,u} sum_t 1/2 tau_t' C_t tau_t + c_t' tau_t     tau_t = [x_t; u_t]
        s.t.      x_{t+1} = f(x_t, u_t),  x_0 = x_init,  u_lower <= u <= u_upper
    """

    def __init__(self, n_state, n_ctrl, T, u_lower=None, u_upper=None, u_zero_I=None, u_init=None, x_init=None,
                 qp_iter=10, grad_method=GradMethods.ANALYTIC, delta_u=None, verbose=0, eps=1e-7, back_eps=1e-7,
                 n_batch=None, linesearch_decay=0.2, max_linesearch_iter=10, exit_unconverged=True,
                 detach_unconverged=True, backprop=True, slew_rate_penalty=None, prev_ctrl=None,
                 not_improved_lim=5, best_cost_eps=1e-4, solver_type="dense", single_qp_solve=False,
                 add_goal_constraint=False, x_goal=None, exit_mode="reference", backend=None, process_group=None):
        super().__init__()
        assert (u_lower is None) == (u_upper is None)
        assert max_linesearch_iter > 0
        if u_lower is None:
            raise NotImplementedError("qp_wrapper.MPC without control bounds is not built (Tracking_MPC always passes them)")
        if slew_rate_penalty is not None or add_goal_constraint or delta_u is not None or u_zero_I is not None:
            raise NotImplementedError("slew_rate_penalty / add_goal_constraint / delta_u / u_zero_I are not "
                                      "reachable from the DEQ-MPC loop and are not built")
        if grad_method not in (GradMethods.ANALYTIC, GradMethods.AUTO_DIFF):
            raise NotImplementedError("grad_method: ANALYTIC or AUTO_DIFF (both call dx_jac, qp_wrapper.py:489, 530)")
        if solver_type != "dense":
            raise NotImplementedError("solver_type must be 'dense' (the only one the reference implements, :307)")
        if exit_mode not in ("reference", "fixed"):
            raise ValueError("exit_mode must be 'reference' or 'fixed'")
        self.n_state, self.n_ctrl, self.T = n_state, n_ctrl, T
        self.u_lower = _detach_maybe(u_lower)
        self.u_upper = _detach_maybe(u_upper)
        self.u_init = _detach_maybe(u_init)
        self.x_init = _detach_maybe(x_init)
        self.qp_iter = qp_iter
        self.grad_method = grad_method
        self.verbose = verbose
        self.eps = eps
        self.back_eps = back_eps
        self.n_batch = n_batch
        self.linesearch_decay = linesearch_decay
        self.max_linesearch_iter = max_linesearch_iter
        self.exit_unconverged = exit_unconverged
        self.detach_unconverged = detach_unconverged
        self.backprop = backprop
        self.not_improved_lim = not_improved_lim
        self.best_cost_eps = best_cost_eps
        self.prev_ctrl = prev_ctrl
        self.solver_type = solver_type
        self.single_qp_solve = single_qp_solve
        self.add_goal_constraint = False
        self.exit_mode = exit_mode
        self._backend = backend
        # a batch sharded over ranks (one process per GPU): the reference's batch-global decisions - the interior-point
        # exit rule (batch_LU.py:120-151) and the SQP line search's `.all()` (qp_wrapper.py:440) - are taken over ALL
        # ranks (one small all-reduce each), so every rank does what the un-sharded reference would
        self.process_group = process_group
        self.last_ipm = None
        self.last_alpha = None

    @property
    def backend(self):
        if self._backend is None:
            from deq_mpc_corl_amd.backend import default_backend
            self._backend = default_backend()
        return self._backend

    def _sharded(self):
        return self.process_group is not None or (
            torch.distributed.is_available() and torch.distributed.is_initialized()
            and getattr(self, "sync_global_exit", False))

    def _pg_kwargs(self):
        return {"process_group": self.process_group, "sharded": True} if self._sharded() else {}

    def _global_all(self, flags):
        """`flags.all()` over the whole batch, i.e. over all ranks of a sharded batch."""
        ok = flags.all().to(torch.int32).reshape(1)
        if self._sharded():
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=self.process_group)
        return bool(ok.item())

    def _global_norm(self, v):
        """`v.norm()` over the whole batch: the squared norm is summed over the ranks of a sharded batch (like
        AL_mpc.MPC._global_norm), so that every rank takes the same decision on it."""
        sq = (v.to(torch.float64) ** 2).sum().reshape(1)
        if self._sharded():
            torch.distributed.all_reduce(sq, op=torch.distributed.ReduceOp.SUM, group=self.process_group)
        return float(sq.sqrt().item())

    def _bound(self, v, like):
        v = torch.as_tensor(v, dtype=like.dtype, device=like.device)
        if v.dim() > 1:
            raise NotImplementedError("per-(t, b) control bounds: the reference's compute_Gh_dense tiles a [n_ctrl] "
                                      "vector over the horizon (qp_wrapper.py:651-652); pass that")
        return v.reshape(-1).expand(self.n_ctrl).contiguous()

    def _uhi(self, like):
        return self._bound(self.u_upper, like)

    def _ulo(self, like):
        return self._bound(self.u_lower, like)

    # -- reference surface -------------------------------------------------------------------------
    def forward(self, x0, cost, dx, dx_jac, dx_true=None):
        self.dx_true = dx if dx_true is None else dx_true
        if not isinstance(cost, QuadCost) and not (hasattr(cost, "C") and hasattr(cost, "c")):
            raise NotImplementedError("only QuadCost costs are built (non-quadratic costs need approximate_cost, "
                                      "which no caller in the repository uses)")
        if self.n_batch is not None:
            B = self.n_batch
        elif cost.C.ndimension() == 4:
            B = cost.C.size(1)
        else:
            raise ValueError("MPC Error: Could not infer batch size, pass in as n_batch")
        n = self.n_state + self.n_ctrl
        C, c = cost.C, cost.c
        if C.ndimension() == 2:
            C = C.unsqueeze(0).unsqueeze(0).expand(self.T, B, n, -1)
        elif C.ndimension() == 3:
            C = C.unsqueeze(1).expand(self.T, B, n, -1)
        if c.ndimension() == 1:
            c = c.unsqueeze(0).unsqueeze(0).expand(self.T, B, -1)
        elif c.ndimension() == 2:
            c = c.unsqueeze(1).expand(self.T, B, -1)
        if C.ndimension() != 4 or c.ndimension() != 3:
            raise ValueError("MPC Error: Unexpected QuadCost shape.")
        cost = QuadCost(C, c)
        assert x0.ndimension() == 2 and x0.size(0) == B
        self.n_batch_run = B

        if self.u_init is None:
            u = torch.zeros(self.T, B, self.n_ctrl, dtype=x0.dtype, device=x0.device)
        else:
            u = self.u_init
            if u.ndimension() == 2:
                u = u.unsqueeze(1).expand(self.T, B, -1).clone()
        u = u.type_as(x0.data)
        if self.x_init is None:
            x = self.rollout(x0, u, dx)
        else:
            x = self.x_init
            if x.ndimension() == 2:
                x = x.unsqueeze(1).expand(self.T, B, -1).clone()
        x = x.type_as(x0.data)

        if self.single_qp_solve:
            x, u, _ = self.single_qp_ls(x, u, dx, dx_jac, x0, cost)
        else:
            x, u, _ = self.solve_nonlin(x, u, dx, dx_jac, x0, cost)
        return x, u

    def _check_diag(self, C):
        off = C - torch.diag_embed(C.diagonal(dim1=-2, dim2=-1))
        if bool((off != 0).any()):
            raise NotImplementedError("qp_wrapper.MPC: only diagonal C_t is built (policies.Tracking_MPC passes "
                                      "torch.diag(Q).repeat(...), policies.py:1172)")

    def single_qp(self, x, u, dx, dx_jac, x0, cost):
        """Linearise, solve the QP, return the step to its solution (qp_wrapper.py:295-321)."""
        B, T, nx, nu = x.shape[1], self.T, self.n_state, self.n_ctrl
        if isinstance(dx, LinDx) or (hasattr(dx, "F") and hasattr(dx, "f") and torch.is_tensor(getattr(dx, "F"))
                                     and not callable(dx)):
            F, f = dx.F, dx.f
            if f is None:
                f = torch.zeros(T - 1, B, nx, dtype=x0.dtype, device=x0.device)
        else:
            F, f = self.linearize_dynamics(x, _detach_maybe(u), dx, dx_jac, diff=False)
        self._check_diag(cost.C)
        ry_fn = None
        if not self._is_lin(self.dx_true):
            # the IPM's equality residual is the residual of the TRUE dynamics at its current iterate
            # (dyn_res callback, qp_wrapper.py:306, 323-342; batch_LU.py:95), not A z - b
            ry_fn = lambda zflat: self.dyn_res(zflat, self.dx_true, x0)
        zhat = _IPQP.apply(cost.C.to(x0.dtype), cost.c.to(x0.dtype), F.to(x0.dtype), f.to(x0.dtype), x0, self, ry_fn)
        zhat = zhat.reshape(B, T, nx + nu)
        x_hat = zhat[:, :, :nx].transpose(0, 1)
        u_hat = zhat[:, :, nx:].transpose(0, 1)
        cost_total = self.compute_cost(zhat, cost)
        return x_hat - x, u_hat - u, cost_total

    @staticmethod
    def _is_lin(dx):
        return isinstance(dx, LinDx) or (hasattr(dx, "F") and hasattr(dx, "f") and not callable(dx))

    def dyn_res(self, z, dx, x0):
        """Equality residual rows in the reference's order: dynamics t = 0..T-2, then x_0 - x0
        (qp_wrapper.py:323-342). z: [B, T*n] (or [B,T,n])."""
        B, T, nx, nu = z.shape[0], self.T, self.n_state, self.n_ctrl
        z = z.reshape(B, T, nx + nu)
        x, u = z[:, :, :nx], z[:, :, nx:]
        if self._is_lin(dx):
            x_next = (dx.F.permute(1, 0, 2, 3) * z[:, :-1, None, :]).sum(-1) + dx.f.permute(1, 0, 2)
        else:
            # the reference evaluates the dynamics on all T stages and drops the last (:333)
            x_next = dx(x.reshape(-1, nx), u.reshape(-1, nu)).reshape(B, T, nx)[:, :-1]
        res = (x_next - x[:, 1:]).reshape(B, -1)
        return torch.cat((res, (x[:, 0] - x0).reshape(B, -1)), dim=1)

    def single_qp_ls(self, x, u, dx, dx_jac, x0, cost):
        """One QP, then the rollout line search decides how far to go along its step (:389-399)."""
        delta_x, delta_u, _ = self.single_qp(x, u, dx, dx_jac, x0, cost)
        with torch.no_grad():
            _, _, alpha, cost_total = self.line_search(x, u, delta_x, delta_u, dx, x0, cost)
        self.last_alpha = alpha
        return x + delta_x * alpha, u + delta_u * alpha, cost_total

    def solve_nonlin(self, x, u, dx, dx_jac, x0, cost):
        """qp_iter x [QP ; line search], best iterate per instance, then one differentiable QP
        from the best point (:345-387). `n_not_improved` is never incremented in the reference
        (:347, 371), so only the step-norm test can end the loop early."""
        best = None
        B = x.shape[1]
        self.last_sqp_iters = 0
        with torch.no_grad():
            for _ in range(self.qp_iter):
                u_prev = u.clone()
                delta_x, delta_u, _ = self.single_qp(x, u, dx, dx_jac, x0, cost)
                x, u, alpha, cost_total = self.line_search(x, u, delta_x, delta_u, dx, x0, cost)
                # the reference's step-norm test runs over the WHOLE batch (qp_wrapper.py:355, 372): a rank-local norm would
                # let the rank whose shard converges first leave the loop and pair its later all-reduces with the wrong ones
                full_du_norm = self._global_norm(u - u_prev)
                if best is None:
                    best = {"x": x.clone(), "u": u.clone(), "costs": cost_total.clone()}
                else:
                    better = cost_total <= best["costs"] + self.best_cost_eps
                    best["x"] = torch.where(better[None, :, None], x, best["x"])
                    best["u"] = torch.where(better[None, :, None], u, best["u"])
                    best["costs"] = torch.where(better, cost_total, best["costs"])
                self.last_sqp_iters = getattr(self, "last_sqp_iters", 0) + 1
                if full_du_norm < self.eps:
                    break
        x, u = best["x"], best["u"]
        delta_x, delta_u, _ = self.single_qp(x, u, dx, dx_jac, x0, cost)
        with torch.no_grad():
            _, _, alpha, cost_total = self.line_search(x, u, delta_x, delta_u, dx, x0, cost)
        self.last_alpha = alpha
        return x + delta_x * alpha, u + delta_u * alpha, cost_total

    def line_search(self, x, u, delta_x, delta_u, dx, x0, cost):
        """Backtracking on the ROLLOUT cost of u + alpha du (:402-421): per-instance alpha shrinks by
        `linesearch_decay` while that instance has not improved; the loop ends when every instance has
        (a batch-global test that only saves work: finished instances are not touched again)."""
        B = x.shape[1]
        alpha = torch.ones(1, B, 1, dtype=x0.dtype, device=x0.device)
        cost_total = self.compute_cost(torch.cat((x, u), dim=2).transpose(0, 1), cost)
        for _ in range(self.max_linesearch_iter):
            u_new = u + delta_u * alpha
            x_new = self.rollout(x0, u_new, dx)
            cost_new = self.compute_cost(torch.cat((x_new, u_new), dim=2).transpose(0, 1), cost)
            if self._global_all(cost_new < cost_total):
                break
            mask = (cost_new >= cost_total).to(x0.dtype)[None, :, None]
            alpha = alpha * self.linesearch_decay * mask + (1 - mask) * alpha
        return x_new, u_new, alpha, cost_new

    def linearize_dynamics(self, x, u, dynamics, dx_jac, diff):
        """F_t = [A_t B_t] and the offset f_t = f(x_t,u_t) - A_t x_t - B_t u_t at every stage (:466-500)."""
        T, B, nx, nu = self.T, x.shape[1], self.n_state, self.n_ctrl
        with torch.enable_grad():
            _u = u[:-1].reshape(-1, nu).detach().requires_grad_(True)
            _x = x[:-1].contiguous().reshape(-1, nx).detach().requires_grad_(True)
            new_x = dynamics(_x, _u)
            R, S = dx_jac(_x, _u)[1]
        new_x, R, S, _x, _u = new_x.detach(), R.detach(), S.detach(), _x.detach(), _u.detach()
        f = new_x - torch.einsum("kij,kj->ki", R, _x) - torch.einsum("kij,kj->ki", S, _u)
        F = torch.cat((R.reshape(T - 1, B, nx, nx), S.reshape(T - 1, B, nx, nu)), dim=3)
        return F.contiguous(), f.reshape(T - 1, B, nx).contiguous()

    def rollout(self, x, actions, dynamics):
        xs = [x]
        for t in range(self.T - 1):
            xt, ut = xs[t], actions[t]
            if self._is_lin(dynamics):
                nxt = torch.einsum("bij,bj->bi", dynamics.F[t], torch.cat([xt, ut], dim=-1)) + dynamics.f[t]
            else:
                nxt = dynamics(xt, ut)
            xs.append(nxt)
        return torch.stack(xs, 0)

    def compute_cost(self, xu, cost):
        """xu [B,T,n] -> [B]  (:656-659); diagonal C."""
        Cd = cost.C.diagonal(dim1=-2, dim2=-1).transpose(0, 1)
        c = cost.c.transpose(0, 1)
        return (0.5 * (xu * Cd * xu).sum(-1) + (xu * c).sum(-1)).sum(-1)
