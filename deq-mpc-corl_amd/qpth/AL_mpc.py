"""Drop-in for the reference's ``qpth.AL_mpc`` module, MI355X-native.

``MPC`` keeps the surface the DEQ-MPC loop touches (callers are
``deqmpc/policies.py:1181-1216, 1236-1315`` of the reference):

  * ctor ``MPC(n_state, n_ctrl, T, u_lower, u_upper, ..., n_batch, dtype)``
    (qpth/AL_mpc.py:118-142; the kwargs the reference ignores are accepted and
    ignored here too),
  * ``__call__(x0, cost, dx, dx_jac, compute_Qq=None, u_init=None, x_init=None)
    -> (x[B,T,nx] f32, u[B,T,nu] f32, status)`` (qpth/AL_mpc.py:207-258),
  * ``reinitialize(x, mask)`` (must precede the first call, :569-579),
    ``warm_start_initialize(x, u, args)`` (:581-592),
  * attributes ``al_iter, x_init, u_init, lamda_prev, rho_prev`` and
    ``get_xu() / get_cost()`` (:560-567).

What runs where: this file is host logic only (state carry, dispatch, the
batch-global exit test that needs a host decision). All arithmetic of the AL
inner iteration - gradient, block-tridiagonal Cholesky, Newton step, 20-point
line search, dual update - is in the HIP kernels behind ``backend`` (csrc/).
There is no CPU path: without the built extension and a ROCm device it raises.

Two exit modes for the Newton loop:
  ``"reference"``  reproduces the reference's batch-global early exit
                   (al_utils.py:486,552,560-564): one kernel launch per Newton
                   step and a host read of sum_b |r+|^2 in between, exactly the
                   host syncs the reference has (its ``.item()`` calls);
  ``"fixed"``      always 4 Newton steps per AL iteration: the whole solve is one
                   kernel launch, no host sync, results do not depend on who else
                   is in the batch (so sharding the batch changes nothing).
"""
from __future__ import annotations

import math
import os
import sys

import torch
from torch.nn import Module

try:  # the package may be reached as top-level `qpth` (drop-in) or via the alias
    import deq_mpc_corl_amd  # noqa: F401
except ImportError:  # pragma: no cover
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import deq_mpc_corl_amd  # noqa: F401

from deq_mpc_corl_amd import _lib as _abi
from deq_mpc_corl_amd.qpth import al_utils
from deq_mpc_corl_amd.qpth.al_utils import LinDx, QuadCost  # noqa: F401  (re-exported)

MAX_NEWTON = 4   # al_utils.py:485
N_LS = 20        # al_utils.py:619
RHO_SCALE = 10.0  # AL_mpc.py:325
SE_NO_BOUND = 1e20  # state estimator: |u| bound that is never active (rho_max * SE_NO_BOUND is finite in fp32)


def _detach_maybe(t):
    if t is None:
        return None
    return t.detach() if t.requires_grad else t


class _SolveState:
    """Everything the solve needs that autograd must not see."""

    __slots__ = ("mpc", "x0", "z", "lam", "rho", "dx", "dx_jac", "lin", "stream_mode",
                 "status_flag", "per_instance_status", "newton_per_al")


class _ALSolve(torch.autograd.Function):
    """The whole AL solve as one differentiable node.

    Like the reference, only the LAST AL iteration is differentiated (earlier ones
    are cut by ``.detach().clone()``, AL_mpc.py:299) and only w.r.t. q and diag(Q)
    (NewtonAL.backward, al_utils.py:578-615): w = -H^{-1} gbar with the factor saved
    at the last executed Newton step, q_grad = w, Q_grad = w * z_final.
    """

    @staticmethod
    def forward(ctx, Qd, q, st):
        mpc = st.mpc
        need_grad = Qd.requires_grad or q.requires_grad
        with torch.no_grad():
            saved = mpc._run(st, Qd.detach().contiguous(), q.detach().contiguous(), need_grad)
        ctx.mpc = mpc
        ctx.has_factor = saved is not None
        if saved is not None:
            kind, factor, F_last, rho_last = saved
            ctx.factor_kind = kind  # "packed" (team kernels) or "workspace" (quad solve's workspace)
            ctx.save_for_backward(factor, F_last, rho_last, st.z)
        ctx.dims = (st.z.shape[0], mpc.T, mpc.n_state, mpc.n_ctrl)
        return st.z.clone()

    @staticmethod
    def backward(ctx, gz):
        if not ctx.has_factor:
            return None, None, None
        factor, F_last, rho_last, z_final = ctx.saved_tensors
        gbar = gz.to(z_final.dtype).contiguous()
        q_grad = torch.empty_like(z_final)
        Qd_grad = torch.empty_like(z_final)
        if ctx.factor_kind == "workspace":
            ctx.mpc.backend.backward_ws(ctx.dims, factor, F_last, rho_last, z_final, gbar, q_grad, Qd_grad)
        else:
            ctx.mpc.backend.backward(ctx.dims, factor, F_last, rho_last, z_final, gbar, q_grad, Qd_grad)
        return Qd_grad, q_grad, None


class MPC(Module):
    """Batched box-constrained MPC solved by an augmented-Lagrangian method
    (same problem statement as qpth/AL_mpc.py:53-63):

        min_{x,u} sum_t 1/2 tau_t' C_t tau_t + c_t' tau_t     tau_t = [x_t; u_t]
        s.t.      x_{t+1} = f(x_t, u_t),  x_0 = x_init,  u_lower <= u <= u_upper
    """

    def __init__(self, n_state, n_ctrl, T, u_lower=None, u_upper=None, u_init=None, x_init=None,
                 al_iter=2, verbose=0, eps=1e-7, back_eps=1e-7, n_batch=None,
                 linesearch_decay=0.2, max_linesearch_iter=10, exit_unconverged=True,
                 detach_unconverged=True, backprop=True, slew_rate_penalty=None,
                 solver_type="dense", add_goal_constraint=False, x_goal=None, diag_cost=True,
                 ineqG=None, ineqh=None, state_estimator=False, dtype=torch.float64,
                 exit_mode="reference", backend=None, process_group=None, prefer_fused=False,
                 check_numerics=None, exit_in_kernel="auto"):
        super().__init__()
        if (u_lower is None) != (u_upper is None) or u_lower is None:
            raise ValueError("MPC: u_lower and u_upper are both required (AL_mpc.py:145,152)")
        if add_goal_constraint or ineqG is not None or not diag_cost:
            raise NotImplementedError("goal constraints / general inequalities / dense cost are "
                                      "not reachable from Tracking_MPC and are not built")
        if exit_mode not in ("reference", "fixed"):
            raise ValueError("exit_mode must be 'reference' or 'fixed'")
        self.dtype = dtype
        self.n_state, self.n_ctrl, self.T = n_state, n_ctrl, T
        self.u_lower = _detach_maybe(torch.as_tensor(u_lower).to(dtype))
        self.u_upper = _detach_maybe(torch.as_tensor(u_upper).to(dtype))
        self.u_init = _detach_maybe(u_init)
        self.x_init = _detach_maybe(x_init)
        self.al_iter = al_iter
        self.verbose = verbose
        self.n_batch = n_batch
        self.diag_cost = True
        self.linearize_once = False
        self.recompute_Qq = False
        # state_estimator=True (AL_mpc.py:179-199 -> qpth/al_utils_se.py): the controls are GIVEN and only the
        # states move; T-1 dynamics row blocks, no initial-state rows, no bound rows (lamda is [B, nx (T-1)]).
        self.state_estimator = bool(state_estimator)
        self.neq = n_state * (T - 1) if self.state_estimator else n_state * T
        self.nineq = 0 if self.state_estimator else 2 * n_ctrl * T
        self.rho_prev = 1.0
        self.rho_max = 1e8
        self.dyn_res_prev = 1000000
        self.exit_mode = exit_mode
        # None: numerical trouble is only recorded (last_info / last_status, no host read-back in the call);
        # "warn" / "raise": one read-back per call, warnings.warn / FloatingPointError when an instance met a
        # non-positive pivot (info) or holds a non-finite iterate (status). The reference has no such report:
        # its cholesky_ex `info` is dropped (al_utils.py:510).
        if check_numerics not in (None, "warn", "raise"):
            raise ValueError("check_numerics must be None, 'warn' or 'raise'")
        self.check_numerics = check_numerics
        # reference exit rule on an un-sharded batch: take the batch-global test inside one cooperative launch (True), or
        # between one-step launches (False; what a sharded batch always does - it needs the all-reduce in between).
        # "auto": inside the launch while the call is launch-bound, i.e. at most 512 wavefronts (half the SIMDs) - measured
        # at (20,13,4): bsz 200 (the reference's) 1.15 against 1.28 ms per call in fp32, 1.60 / 1.73 in fp64; B = 512
        # 1.29 / 1.39; from B = 1024 on the launches win (1.61 / 1.51; B = 2048: 2.41 / 1.68; B = 16384: 4.9 / 4.4): a grid
        # barrier per Newton step makes every wavefront wait for the slowest and spin next to working ones
        if exit_in_kernel not in (True, False, "auto"):
            raise ValueError("exit_in_kernel must be True, False or 'auto'")
        self.exit_in_kernel = exit_in_kernel
        self.prefer_fused = bool(prefer_fused)  # take the compiled-in dynamics model even where it is not the default
        self.process_group = process_group
        self._backend = backend
        self.warm_starting = None  # set by reinitialize(); forward() insists on it
        self.cost_hist_stream = [[], []]
        self.lamda_prev = None
        if n_batch is not None:
            self.lamda_prev = torch.zeros(n_batch, self.neq + self.nineq, dtype=dtype,
                                          device=self.u_upper.device)
        self.mask = None
        self.last_status = None
        self.last_info = None
        self.last_newton_per_al = None

    # -- backend -----------------------------------------------------------------
    @property
    def backend(self):
        if self._backend is None:
            from deq_mpc_corl_amd.backend import default_backend
            self._backend = default_backend()
        return self._backend

    # -- reference surface ---------------------------------------------------------
    def reinitialize(self, x, mask):
        """rho <- 1, lam <- 0, warm starts dropped (AL_mpc.py:569-579)."""
        self.u_init = None
        self.x_init = None
        self.n_batch = x.size(0)
        self.rho_prev = torch.ones((self.n_batch, 1), device=x.device, dtype=x.dtype)
        self.lamda_prev = torch.zeros(self.n_batch, self.neq + self.nineq, device=x.device, dtype=x.dtype)
        self.cost_hist_stream = [[], []]
        self.dyn_res_prev = 1000000
        self.just_initialized = True
        self.warm_starting = False
        self.mask = mask

    def warm_start_initialize(self, x, u, args):
        """Streaming warm start (AL_mpc.py:581-592): inits <- given, multipliers are
        shifted one stage and then zeroed (the reference multiplies by 0, :589),
        rho <- min(rho, args.rho_init_max)."""
        self.u_init = u
        self.x_init = x
        self.lamda_prev = torch.zeros_like(self.lamda_prev)
        self.rho_prev = torch.clamp(self.rho_prev, max=args.rho_init_max)
        self.just_initialized = True
        self.warm_starting = True

    @property
    def last_status(self):
        """Per-instance finiteness flag of the last solve (bool tensor), None before the first."""
        raw = getattr(self, "_status_raw", None)
        return None if raw is None else raw.bool()

    @last_status.setter
    def last_status(self, v):
        self._status_raw = v

    @property
    def dyn_res_prev(self):
        """||r+|| per instance after the last solve (AL_mpc.py's `dyn_res_prev`); the reference's initial 1000000 before."""
        raw = getattr(self, "_rn2_raw", None)
        return self._dyn_res_init if raw is None else raw.sqrt()

    @dyn_res_prev.setter
    def dyn_res_prev(self, v):
        self._dyn_res_init = v
        self._rn2_raw = None

    def get_xu(self):
        return torch.cat((self.x_init, self.u_init), dim=2)

    def get_cost(self, cost):
        xu = self.get_xu()
        Qd = cost.C.diagonal(dim1=-2, dim2=-1)
        f = cost.f.sum(dim=-1) if cost.f is not None else 0.0
        return (0.5 * (xu * Qd * xu).sum(-1) + (cost.c * xu).sum(-1)).sum(dim=-1) + f

    def rollout(self, x, actions, dynamics):
        """x_{t+1} = f(x_t, u_t) from x_0 (AL_mpc.py:521-534)."""
        xs = [x]
        lin = self._as_lindx(dynamics, x.shape[0])
        for t in range(self.T - 1):
            xt, ut = xs[t], actions[:, t]
            if lin is not None:
                F, c = lin
                nxt = torch.einsum("bij,bj->bi", F[:, t].to(xt.dtype), torch.cat([xt, ut], -1)) + c[:, t].to(xt.dtype)
            else:
                nxt = dynamics(xt, ut)
            xs.append(nxt)
        return torch.stack(xs, 1)

    def forward(self, x0, cost, dx, dx_jac, compute_Qq=None, u_init=None, x_init=None):
        if self.warm_starting is None:
            raise RuntimeError("MPC.reinitialize(x, mask) must be called before the first solve "
                               "(the reference creates `warm_starting` there, AL_mpc.py:578)")
        self.compute_Qq = compute_Qq
        B = self.n_batch if self.n_batch is not None else cost.C.size(0)
        assert cost.C.ndimension() == 4
        assert x0.ndimension() == 2 and x0.size(0) == B

        def expand(v):
            return v.unsqueeze(0).expand(B, self.T, -1).clone() if v.ndimension() == 2 else v

        if u_init is not None:
            u = expand(u_init)
        elif self.u_init is None:
            u = torch.zeros(B, self.T, self.n_ctrl, dtype=x0.dtype, device=x0.device)
        else:
            u = expand(self.u_init)
        u = u.type_as(x0.data)
        if x_init is not None:
            x = expand(x_init)
        elif self.x_init is None:
            x = self.rollout(x0, u, dx)
        else:
            x = expand(self.x_init)
        x = x.type_as(x0.data)

        Qd = cost.C.diagonal(dim1=-2, dim2=-1)
        x, u, status = self._al_solve(x, u, dx, dx_jac, x0, Qd, cost.c, bool(self.warm_starting))
        self.x_init = x.detach().clone()
        self.u_init = u.detach().clone()
        return x, u, status

    # the reference exposes both names; both end in the same machinery here
    def al_solve(self, x, u, dx, dx_jac, x0, cost, lamda_init=None, rho_init=None):
        if lamda_init is not None:
            self.lamda_prev = lamda_init
        if rho_init is not None:
            self.rho_prev = rho_init
        return self._al_solve(x, u, dx, dx_jac, x0, cost.C, cost.c, False)

    def al_solve_stream(self, x, u, dx, dx_jac, x0, cost, lamda_init=None, rho_init=None):
        if lamda_init is not None:
            self.lamda_prev = lamda_init
        if rho_init is not None:
            self.rho_prev = rho_init
        return self._al_solve(x, u, dx, dx_jac, x0, cost.C, cost.c, True)

    # -- internals -------------------------------------------------------------------
    def _obs_kwargs(self, dtype, device):
        """Extra backend arguments describing the constraint-row set: obstacle rows (Obstacle_MPC), or the
        state-estimator variant's set without initial-state rows; the plain MPC has none."""
        return {"obs": "state_estimator"} if self.state_estimator else {}

    def _has_extra_rows(self):
        """True when the constraint-row set differs from the plain MPC's (obstacle rows, state estimator): a predicate
        that moves no tensor (the former `bool(_obs_kwargs(dtype, "cpu"))` copied the obstacle centres to the host and
        synchronised on every solve)."""
        return bool(self.state_estimator)

    def _as_lindx(self, dx, B):
        """(F[B,T-1,nx,n], c[B,T-1,nx]) if `dx` carries affine data, else None."""
        if self._has_extra_rows():
            return None   # obstacle rows exist only in the nonlinear-caller building blocks
        F = getattr(dx, "F", None)
        c = getattr(dx, "f", None)
        if F is None or c is None or not torch.is_tensor(F):
            return None
        n = self.n_state + self.n_ctrl
        if tuple(F.shape) != (B, self.T - 1, self.n_state, n):
            raise ValueError(f"LinDx.F must be [B,T-1,nx,n]={B, self.T - 1, self.n_state, n}, got {tuple(F.shape)}")
        return F, c

    def _bounds(self, B, dtype, device):
        nu = self.n_ctrl
        if self.state_estimator:
            # no bound rows in this variant (AL_mpc.py:198): bounds no control reaches keep the kernels' bound
            # rows at residual 0 and multiplier 0 for every rho up to rho_max
            big = torch.full((nu,), SE_NO_BOUND, dtype=dtype, device=device)
            return -big, big, 0, 0
        lo = self.u_lower.to(device=device, dtype=dtype)
        hi = self.u_upper.to(device=device, dtype=dtype)
        if lo.dim() <= 1 and hi.dim() <= 1:
            lo = lo.reshape(-1).expand(nu).contiguous()
            hi = hi.reshape(-1).expand(nu).contiguous()
            return lo, hi, 0, 0
        lo = lo.expand(B, self.T, nu).contiguous()
        hi = hi.expand(B, self.T, nu).contiguous()
        return lo, hi, self.T * nu, nu

    def _rho_tensor(self, B, dtype, device):
        r = self.rho_prev
        if not torch.is_tensor(r):
            return torch.full((B,), float(r), dtype=dtype, device=device)
        return r.to(device=device, dtype=dtype).reshape(B).contiguous().clone()

    @staticmethod
    def _exit_terms(rn2, info):
        """Per-instance terms of the batch norm the Newton loop exits on. An instance whose factorisation hit a
        non-positive pivot (sticky info != 0) counts as +inf: the exit then never fires in this call. The reference's
        norm (torch.norm(dyn_res), al_utils.py:552) holds the UNDEFINED residuals of such instances (half-finished
        Cholesky factor, :510-515) and kept both recorded batches at the full 4 Newton steps
        (tests/golden/fail_*.npz: newton_per_al = [4]); without its garbage the healthy instances alone would stop
        after 3 and end 0.03 / 0.06 (fp64 / fp32) away from the reference's controls."""
        if info is None:
            return rn2
        return torch.where(info != 0, torch.full_like(rn2, float("inf")), rn2)

    def _global_sumsq(self, rn2, info=None):
        """sum_b sum_rows r+^2 as a 1-element float64 device tensor (all-reduced over the ranks
        of a sharded batch); nothing is synchronised with the host."""
        s = self._exit_terms(rn2, info).sum(dtype=torch.float64).reshape(1)
        if self.process_group is not None or (
                torch.distributed.is_available() and torch.distributed.is_initialized()
                and getattr(self, "sync_global_exit", False)):
            torch.distributed.all_reduce(s, group=self.process_group)
        return s

    def _global_norm(self, rn2, info=None):
        """sqrt(sum_b sum_rows r+^2): the batch-global quantity the reference exits on
        (torch.norm(dyn_res).item(), al_utils.py:486,552). With a sharded batch the
        partial sums are all-reduced (8 bytes over RCCL) so that every rank takes the
        same decision the un-sharded reference would."""
        s = self._exit_terms(rn2, info).sum(dtype=torch.float64)
        if self.process_group is not None or (
                torch.distributed.is_available() and torch.distributed.is_initialized()
                and getattr(self, "sync_global_exit", False)):
            torch.distributed.all_reduce(s, group=self.process_group)
        return math.sqrt(float(s.item()))

    def _sharded(self):
        return self.process_group is not None or (
            torch.distributed.is_available() and torch.distributed.is_initialized()
            and getattr(self, "sync_global_exit", False))

    def _global_mean(self, v):
        """Batch mean of a per-instance quantity (the stream loop's break test,
        `dyn_res_clamp.mean().item()`, AL_mpc.py:406-408); over ALL ranks of a sharded batch, so
        that every rank leaves the loop in the same iteration."""
        s = torch.stack((v.sum(dtype=torch.float64), torch.tensor(float(v.numel()), dtype=torch.float64, device=v.device)))
        if self._sharded():
            torch.distributed.all_reduce(s, group=self.process_group)
        s = s.tolist()
        return s[0] / s[1]

    def _global_max(self, v):
        """`rho.max().item()` (AL_mpc.py:412, 420), over all ranks of a sharded batch."""
        m = v.max().to(torch.float64).reshape(1)
        if self._sharded():
            torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MAX, group=self.process_group)
        return float(m.item())

    def _al_solve(self, x, u, dx, dx_jac, x0, Qd, q, stream_mode):
        B = x.shape[0]
        dt = self.dtype
        dev = x0.device
        if self.lamda_prev is None:
            self.lamda_prev = torch.zeros(B, self.neq + self.nineq, dtype=dt, device=dev)
        nrows = self.n_state * self.T + 2 * self.n_ctrl * self.T   # the kernels' row layout
        st = _SolveState()
        st.mpc = self
        st.x0 = x0.detach().to(dt).contiguous()
        st.z = torch.cat((x, u), dim=2).detach().to(dt).contiguous()   # (cat made a fresh tensor: no clone)
        st.lam = self.lamda_prev.detach().to(device=dev, dtype=dt).contiguous().clone()
        if self.state_estimator:   # [B, nx (T-1)] -> the kernels' layout; the init and bound rows stay 0
            st.lam = torch.cat((st.lam, st.lam.new_zeros(B, nrows - self.neq)), dim=1).contiguous()
        st.rho = self._rho_tensor(B, dt, dev)
        st.dx, st.dx_jac = dx, dx_jac
        st.lin = self._as_lindx(dx, B)
        st.stream_mode = stream_mode
        st.status_flag = False
        z = _ALSolve.apply(Qd.to(dt), q.to(dt), st)
        self.lamda_prev = st.lam[:, :self.neq].contiguous() if self.state_estimator else st.lam
        self.rho_prev = st.rho.reshape(B, 1)
        self.just_initialized = False
        self.last_newton_per_al = st.newton_per_al
        nx = self.n_state
        return z[..., :nx].float(), z[..., nx:].float(), st.status_flag

    # one NewtonAL.forward worth of work on affine data, host-driven exit
    def _newton_al_lin(self, st, Qd, q, F, c, bnd, ws, need_factor):
        be = self.backend
        dims = (st.z.shape[0], self.T, self.n_state, self.n_ctrl)
        lo, hi, sb, stt = bnd
        common = dict(rnorm2=ws["rn2"], info=ws["info"], status=ws["status"], n_ls=N_LS,
                      rho_scale=RHO_SCALE)
        if ws.get("qws") is not None:   # quad solve on a private workspace: it IS the saved factor
            common["workspace"] = ws["qws"]
            common["variant"] = "quad"
            need_factor = False
        fl_save = _abi.ALQP_SAVE_FACTOR if need_factor else 0
        wsx = common.get("workspace")
        if wsx is None and hasattr(be, "_workspace"):
            wsx = be._workspace(dims, st.z)[0]   # the cached scratch a quad launch will use
        # ALQP_WS_PRIMED bookkeeping: ws["primed"] is the workspace whose records hold THIS
        # solve's current z/lam (written by the previous quad launch), else None
        def pflag():
            return _abi.ALQP_WS_PRIMED if (wsx is not None and ws.get("primed") is wsx) else 0
        def after():
            ws["primed"] = wsx if getattr(be, "last_variant", None) == "quad" else None
        if self.exit_mode == "fixed":
            be.solve_lin(dims, Qd, q, F, c, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho, ws["phi"],
                         factor=ws.get("factor"), al_iter=1, max_newton=MAX_NEWTON,
                         flags=_abi.ALQP_INIT_MERIT | fl_save | (0 if fl_save else pflag()), **common)
            after()
            return MAX_NEWTON
        be.solve_lin(dims, Qd, q, F, c, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho, ws["phi"],
                     al_iter=1, max_newton=0, flags=_abi.ALQP_INIT_MERIT | pflag(), **common)
        after()
        # The reference's batch-global exit test (al_utils.py:551-564) is taken on the device: all
        # MAX_NEWTON launches are enqueued, each one a no-op once ctl[0] is set, and the number of
        # executed steps (ctl[1]) is read back once per solve. No host round trip per Newton step.
        ctl = torch.zeros(3, dtype=torch.float64, device=st.z.device)
        be.exit_test(self._global_sumsq(ws["rn2"], ws["info"]), ctl, 0)
        for _ in range(MAX_NEWTON):
            # same workspace as the launch before, nothing touched in between: no copy-in pass
            be.solve_lin(dims, Qd, q, F, c, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho, ws["phi"],
                         factor=ws.get("factor"), al_iter=1, max_newton=1,
                         flags=fl_save | (0 if fl_save else pflag()), skip=ctl, **common)
            after()
            be.exit_test(self._global_sumsq(ws["rn2"], ws["info"]), ctl, 1)
        return ctl

    def _newton_al_fused_nl(self, st, Qd, q, bnd, ws):
        """NewtonAL.forward for a dynamics model compiled into the library, reference exit rule:
        alqp_solve_nonlin once per Newton step, alqp_exit_test in between (no host round trip)."""
        be = self.backend
        dims = (st.z.shape[0], self.T, self.n_state, self.n_ctrl)
        lo, hi, sb, stt = bnd
        common = dict(rnorm2=ws["rn2"], info=ws["info"], status=ws["status"], rho_scale=RHO_SCALE,
                      workspace=ws.get("nlws"))
        args = (dims, st.dx.fused_id, st.dx.dt, Qd, q, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho, ws["phi"])
        be.solve_nonlin(*args, al_iter=1, max_newton=0, flags=_abi.ALQP_INIT_MERIT, **common)
        ctl = torch.zeros(3, dtype=torch.float64, device=st.z.device)
        be.exit_test(self._global_sumsq(ws["rn2"], ws["info"]), ctl, 0)
        for _ in range(MAX_NEWTON):
            be.solve_nonlin(*args, al_iter=1, max_newton=1, flags=0, skip=ctl, **common)
            be.exit_test(self._global_sumsq(ws["rn2"], ws["info"]), ctl, 1)
        return ctl

    def _linearize(self, st, z):
        """dx_jac at every (x_t, u_t), t < T-1 -> (f(z) [B,T-1,nx], F = [A|B] [B,T-1,nx,n]).
        Called the way the reference does (al_utils.py:501-503, 233-248): grad mode on and the
        iterate requiring grad, because the torch-coded environments differentiate through their own
        RK4 step inside `dx_jac` (rex_quadrotor.py:136-144)."""
        B, T, nx, nu = z.shape[0], self.T, self.n_state, self.n_ctrl
        with torch.enable_grad():
            zz = z.detach().requires_grad_(True)
            xn_j, (A, Bm) = st.dx_jac(zz[:, :-1, :nx].reshape(-1, nx), zz[:, :-1, nx:].reshape(-1, nu))
        Bm = Bm.detach().reshape(B, T - 1, nx, nu)
        if self.state_estimator:
            Bm = Bm * 0.0   # the controls do not move (al_utils_se.py:151)
        F = torch.cat((A.detach().reshape(B, T - 1, nx, nx), Bm), dim=-1).to(z.dtype).contiguous()
        return xn_j.detach().reshape(B, T - 1, nx).to(z.dtype).contiguous(), F

    def _newton_al_nonlin(self, st, Qd, q, bnd, ws, need_factor):
        """NewtonAL.forward (al_utils.py:451-576) with `dx`/`dx_jac` as PyTorch calls
        between kernel launches."""
        be = self.backend
        B, T, nx, nu = st.z.shape[0], self.T, self.n_state, self.n_ctrl
        n = nx + nu
        dims = (B, T, nx, nu)
        lo, hi, sb, stt = bnd
        z = st.z
        dt = z.dtype

        def dyn(zz):  # zz [..., T, n] -> f(x_t,u_t) [..., T-1, nx]
            lead = zz.shape[:-2]
            xn = st.dx(zz[..., :-1, :nx].reshape(-1, nx), zz[..., :-1, nx:].reshape(-1, nu))
            return xn.reshape(*lead, T - 1, nx).to(dt).contiguous()

        okw = self._obs_kwargs(dt, z.device)   # {} or {"obs": (centres, radius)} (Obstacle_MPC)
        xn = dyn(z)
        be.merit(dims, 1, z, xn, st.x0, st.lam, st.rho, Qd, q, lo, hi, sb, stt, ws["phi"], ws["rn2"], **okw)
        old = self._global_norm(ws["rn2"], ws["info"]) if self.exit_mode == "reference" else None
        alphas = (2.0 ** -torch.arange(N_LS, device=z.device, dtype=dt)).view(N_LS, 1, 1, 1)
        steps = 0
        while steps < MAX_NEWTON:
            steps += 1
            xn, F = self._linearize(st, z)
            # from B = QUAD_MIN_BATCH on the direction comes from the quad kernels,
            # whose factor stays in the workspace records: a private one (ws["qws"]) when backward follows
            qws = ws.get("qws")
            if qws is None and hasattr(be, "_workspace") and B >= getattr(be, "QUAD_MIN_BATCH", 1 << 62):
                qws = be._workspace(dims, z)[0]
            if qws is not None:   # (obstacle rows / the state-estimator row set included: alqp_newton_step_ws_obs)
                be.newton_step(dims, z, xn, F, st.x0, st.lam, st.rho, Qd, q, lo, hi, sb, stt, ws["d"],
                               info=ws["info"], workspace=qws, **okw)
            else:
                be.newton_step(dims, z, xn, F, st.x0, st.lam, st.rho, Qd, q, lo, hi, sb, stt, ws["d"],
                               factor=ws.get("factor") if need_factor else None, info=ws["info"], **okw)
            if need_factor:
                ws["F_last"] = F
            zc = (z.unsqueeze(0) + alphas * ws["d"].unsqueeze(0)).contiguous()
            xnc = dyn(zc)
            # 20 merits + decision + update in one launch; rn2 <- the chosen candidate's when accepted
            be.merit_pick(dims, N_LS, ws["d"], xnc, st.x0, st.lam, st.rho, Qd, q, lo, hi, sb, stt, z, ws["phi"],
                          rnorm2=ws["rn2"], k_out=ws["k"], accept_out=ws["acc"], **okw)
            if self.exit_mode == "reference":
                new = self._global_norm(ws["rn2"], ws["info"])
                if new < 1e-3 or (math.isfinite(new) and abs(old - new) / new < 1e-3):   # inf: a tripped instance, no exit
                    break
                old = new
        return steps

    def _run(self, st, Qd, q, need_grad):
        """The AL outer loop (AL_mpc.py:260-339 / :342-423). Mutates st.z/lam/rho.
        Returns (factor, F_last, rho_last) when a backward pass may follow."""
        be = self.backend
        B, T, nx, nu = st.z.shape[0], self.T, self.n_state, self.n_ctrl
        n = nx + nu
        dt, dev = st.z.dtype, st.z.device
        dims = (B, T, nx, nu)
        if not be.supported(B, T, nx, nu, dt):
            raise RuntimeError(f"mi_alqp: no kernel instance for (nx={nx}, nu={nu}, T={T}, {dt}); "
                               "add it to ALQP_FOR_EACH_DIMS in csrc/alqp_kernels.hip")
        bnd = self._bounds(B, dt, dev)
        lo, hi, sb, stt = bnd
        if self.state_estimator:
            # cost gradient on the states only (al_utils_se.py:300-310) while the Hessian keeps diag(Q) on the
            # controls (:66-68): the kernels' `state_estimator` flag. Their merit still counts the controls' cost
            # terms, the same constant for every line-search candidate since du = 0 exactly (:31 leaves them out).
            if torch.is_tensor(getattr(st.dx, "F", None)) or self.linearize_once:
                raise NotImplementedError("state_estimator: only the callable dx / dx_jac route exists "
                                          "(al_utils_se.py has no LinDx or frozen-linearisation branch)")
        # phi, rn2 and info out of ONE zeroed allocation (one fill kernel instead of three)
        esz = torch.empty((), dtype=dt).element_size()
        zb = torch.zeros(B * (2 * esz + 4), dtype=torch.uint8, device=dev)
        ws = {"phi": zb[:B * esz].view(dt), "rn2": zb[B * esz:2 * B * esz].view(dt),
              "info": zb[2 * B * esz:].view(torch.int32),
              "status": torch.ones(B, dtype=torch.uint8, device=dev)}
        lin = st.lin
        if need_grad and self.linearize_once and st.stream_mode:
            # al_utils_lin.NewtonAL.backward returns 14 gradients for 15 inputs (al_utils_lin.py:444-459):
            # autograd rejects it, so the reference cannot differentiate this route. Same error type.
            raise RuntimeError("MPC: the linearize_once streaming route is not differentiable (the reference's "
                               "al_utils_lin.NewtonAL.backward returns an incorrect number of gradients)")
        has_obs = self._has_extra_rows()
        # (round 3: the quad step kernel takes the obstacle / state-estimator rows too, so their factor can stay in its records)
        use_qws = need_grad and hasattr(be, "backward_ws") and B >= getattr(be, "QUAD_MIN_BATCH", 0) and not (
            bool(self.linearize_once) and st.stream_mode)
        if use_qws:
            ws["qws"] = be.new_workspace(dims, st.z)
        elif need_grad:
            ws["factor"] = torch.empty(B, T, n * (n + 1) // 2, dtype=dt, device=dev)
        stream = st.stream_mode
        if self.linearize_once and not stream:
            # The reference cannot run this combination either: al_solve (AL_mpc.py:292-306) hands the
            # frozen-linearisation dict to al_utils.merit_grad_hessian, which calls it
            # ("TypeError: 'dict' object is not callable", al_utils.py:237). Same error type here.
            raise TypeError("MPC: linearize_once is only defined for the streaming route (after "
                            "warm_start_initialize); the reference's al_solve raises TypeError on it as well")
        linearize_once = bool(self.linearize_once) and stream
        if has_obs and linearize_once:
            raise NotImplementedError("Obstacle_MPC with linearize_once: the reference's frozen-linearisation "
                                      "module knows no obstacle rows (AL_mpc_custom.py:68, 75, 83)")
        npa = []
        rho_last = None
        F_last = None

        if lin is None or linearize_once:
            ws.update(d=torch.empty(B, T, n, dtype=dt, device=dev),
                      phis=torch.empty(N_LS, B, dtype=dt, device=dev),
                      rn2s=torch.empty(N_LS, B, dtype=dt, device=dev),
                      k=torch.zeros(B, dtype=torch.int32, device=dev),
                      acc=torch.zeros(B, dtype=torch.int32, device=dev))

        def true_next(z):
            xn = st.dx(z[:, :-1, :nx].reshape(-1, nx), z[:, :-1, nx:].reshape(-1, nu))
            return xn.reshape(B, T - 1, nx).to(dt).contiguous()

        F = c = None
        if lin is not None and not linearize_once:
            F = lin[0].detach().to(device=dev, dtype=dt).contiguous()
            c = lin[1].detach().to(device=dev, dtype=dt).contiguous()
        elif linearize_once:
            # frozen linearisation captured once per call (al_utils_lin.py:140-169)
            # note the offset: x_{t+1} of the WARM START minus F_t z_t, not f(z_t) - F_t z_t (:154)
            z = st.z
            _, F = self._linearize(st, z)
            c = (z[:, 1:, :nx] - torch.einsum("btij,btj->bti", F, z[:, :-1])).contiguous()

        # ---- reference exit rule, un-sharded batch, affine dynamics: the batch-global test of the Newton loop is taken
        # INSIDE one cooperative launch (grid-wide barrier + ordered sum per Newton step, ALQP_EXIT_IN_KERNEL) instead
        # of a launch per Newton step with alqp_exit_test in between. Falls through when the grid cannot be co-resident.
        done_in_kernel = False
        coop = (F is not None and not linearize_once and self.exit_mode == "reference" and not self._sharded()
                and getattr(be, "supports_exit_in_kernel", False)
                and (self.exit_in_kernel is True
                     or (self.exit_in_kernel == "auto" and B < getattr(be, "QUAD_MIN_BATCH", 0)
                         and -(-B // max(1, be.qps_per_wave(B, T, nx, nu, dt))) <= 512)))
        if coop and not stream:
            save = need_grad and not use_qws
            flags = _abi.ALQP_INIT_MERIT | _abi.ALQP_DUAL_UPDATE | (_abi.ALQP_SAVE_FACTOR if save else 0)
            extra = dict(workspace=ws["qws"], variant="quad") if use_qws else {}
            counts = torch.zeros(self.al_iter, dtype=torch.int32, device=dev)
            done_in_kernel = be.solve_lin(dims, Qd, q, F, c, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho, ws["phi"],
                                          rnorm2=ws["rn2"], info=ws["info"], status=ws["status"],
                                          factor=ws.get("factor"), al_iter=self.al_iter, max_newton=MAX_NEWTON,
                                          n_ls=N_LS, flags=flags, rho_scale=RHO_SCALE, newton_counts=counts, **extra)
            if done_in_kernel:
                npa = list(counts.unbind())
                rho_last = st.rho / RHO_SCALE
                F_last = F

        # ---- fast path: the whole solve in ONE launch -----------------------------------
        if done_in_kernel:
            pass
        elif F is not None and not stream and self.exit_mode == "fixed":
            save = need_grad and not use_qws
            flags = _abi.ALQP_INIT_MERIT | _abi.ALQP_DUAL_UPDATE | (_abi.ALQP_SAVE_FACTOR if save else 0)
            extra = dict(workspace=ws["qws"], variant="quad") if use_qws else {}
            be.solve_lin(dims, Qd, q, F, c, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho, ws["phi"],
                         rnorm2=ws["rn2"], info=ws["info"], status=ws["status"],
                         factor=ws.get("factor"), al_iter=self.al_iter, max_newton=MAX_NEWTON,
                         n_ls=N_LS, flags=flags, rho_scale=RHO_SCALE, **extra)
            npa = [MAX_NEWTON] * self.al_iter
            rho_last = st.rho / RHO_SCALE if need_grad else None
            F_last = F
        elif (F is None and not stream and self.exit_mode == "fixed" and not has_obs
              and getattr(st.dx, "fused_id", None) is not None and hasattr(be, "solve_nonlin")
              and (getattr(st.dx, "fused_default", True) or self.prefer_fused)
              and (getattr(st.dx, "nx", None), getattr(st.dx, "nu", None)) == (nx, nu)):
            # ---- nonlinear dynamics whose model is compiled into the library (dynamics.py): the
            # whole nonlinear solve in ONE launch, no PyTorch round trip between Newton steps
            if need_grad:   # private workspace: its records and F region are the saved factor
                ws["nlws"] = be.new_workspace_nonlin(dims, st.z)
            be.solve_nonlin(dims, st.dx.fused_id, st.dx.dt, Qd, q, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho,
                            ws["phi"], rnorm2=ws["rn2"], info=ws["info"], status=ws["status"],
                            al_iter=self.al_iter, max_newton=MAX_NEWTON,
                            flags=_abi.ALQP_INIT_MERIT | _abi.ALQP_DUAL_UPDATE, rho_scale=RHO_SCALE,
                            workspace=ws.get("nlws"))
            npa = [MAX_NEWTON] * self.al_iter
            rho_last = st.rho / RHO_SCALE
            if need_grad:
                F_last = be.nonlin_F_view(ws["nlws"], dims)
        else:
            num_iters = 100 if linearize_once else self.al_iter
            prev_mean = None
            if linearize_once:  # dyn_res_clamp_prev starts at the residual of the warm start (:358-369)
                xn = true_next(st.z)
                be.merit(dims, 1, st.z, xn, st.x0, st.lam, st.rho, Qd, q, lo, hi, sb, stt,
                         ws["phi"], ws["rn2"])
                prev_mean = self._global_mean(ws["rn2"].sqrt())
            fused_nl = (F is None and not stream and self.exit_mode == "reference" and not has_obs
                        and getattr(st.dx, "fused_id", None) is not None and hasattr(be, "solve_nonlin")
                        and (getattr(st.dx, "fused_default", True) or self.prefer_fused)
                        and (getattr(st.dx, "nx", None), getattr(st.dx, "nu", None)) == (nx, nu))
            if fused_nl and not self._sharded() and getattr(be, "supports_exit_in_kernel", False) and (
                    self.exit_in_kernel is True or (self.exit_in_kernel == "auto" and -(-B // 16) <= 512)):
                # launch-bound batch: the whole nonlinear solve with the reference's exit rule in ONE cooperative launch
                if need_grad and "nlws" not in ws:
                    ws["nlws"] = be.new_workspace_nonlin(dims, st.z)
                counts = torch.zeros(self.al_iter, dtype=torch.int32, device=dev)
                if be.solve_nonlin(dims, st.dx.fused_id, st.dx.dt, Qd, q, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho,
                                   ws["phi"], rnorm2=ws["rn2"], info=ws["info"], status=ws["status"],
                                   al_iter=self.al_iter, max_newton=MAX_NEWTON,
                                   flags=_abi.ALQP_INIT_MERIT | _abi.ALQP_DUAL_UPDATE, rho_scale=RHO_SCALE,
                                   workspace=ws.get("nlws"), newton_counts=counts):
                    npa = list(counts.unbind())
                    rho_last = st.rho / RHO_SCALE
                    if need_grad:
                        F_last = be.nonlin_F_view(ws["nlws"], dims)
                    num_iters = 0
            for _ in range(num_iters):
                rho_last = st.rho.clone()
                if fused_nl:
                    # compiled-in model, reference exit: one launch per Newton step (the model inlined),
                    # the batch-global exit test on the device, then the dual update launch
                    if need_grad and "nlws" not in ws:
                        ws["nlws"] = be.new_workspace_nonlin(dims, st.z)
                    npa.append(self._newton_al_fused_nl(st, Qd, q, bnd, ws))
                    be.solve_nonlin(dims, st.dx.fused_id, st.dx.dt, Qd, q, st.x0, lo, hi, sb, stt, st.z, st.lam,
                                    st.rho, ws["phi"], rnorm2=ws["rn2"], info=None, status=ws["status"],
                                    al_iter=1, max_newton=0, flags=_abi.ALQP_DUAL_UPDATE, rho_scale=RHO_SCALE,
                                    workspace=ws.get("nlws"))
                    if need_grad:   # L and F of the last executed Newton step are still in the workspace
                        F_last = be.nonlin_F_view(ws["nlws"], dims)
                    continue
                if coop and ws.get("coop_ok", True):
                    # one AL iteration (starting merit, Newton loop with the batch-global exit, dual update) per launch
                    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
                    save = need_grad and not use_qws
                    extra = dict(workspace=ws["qws"], variant="quad") if use_qws else {}
                    wsx = extra.get("workspace")
                    if wsx is None and hasattr(be, "_workspace"):
                        wsx = be._workspace(dims, st.z)[0]
                    pf = _abi.ALQP_WS_PRIMED if (not save and wsx is not None and ws.get("primed") is wsx) else 0
                    ok = be.solve_lin(dims, Qd, q, F, c, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho, ws["phi"],
                                      rnorm2=ws["rn2"], info=ws["info"], status=ws["status"], factor=ws.get("factor"),
                                      al_iter=1, max_newton=MAX_NEWTON, n_ls=N_LS, rho_scale=RHO_SCALE, newton_counts=cnt,
                                      flags=_abi.ALQP_INIT_MERIT | _abi.ALQP_DUAL_UPDATE | (_abi.ALQP_SAVE_FACTOR if save else 0) | pf,
                                      **extra)
                    ws["coop_ok"] = ok
                    if ok:
                        ws["primed"] = wsx if getattr(be, "last_variant", None) == "quad" else None
                        npa.append(cnt[0])
                        F_last = F
                        if stream and self._global_max(st.rho) > self.rho_max:
                            break
                        continue
                if F is not None:
                    npa.append(self._newton_al_lin(st, Qd, q, F, c, bnd, ws, need_grad))
                    F_last = F
                else:
                    npa.append(self._newton_al_nonlin(st, Qd, q, bnd, ws, need_grad))
                    F_last = ws.get("F_last")
                # dual update with the TRUE dynamics (AL_mpc.py:315-317 / :397-399)
                if F is not None and not linearize_once:
                    # (never on the private workspace: its y/r/s slots may be rewritten, its L not,
                    #  but keep the saved factor's workspace out of later launches altogether)
                    wsc = be._workspace(dims, st.z)[0] if hasattr(be, "_workspace") else None
                    pf = _abi.ALQP_WS_PRIMED if (wsc is not None and ws.get("primed") is wsc) else 0
                    be.solve_lin(dims, Qd, q, F, c, st.x0, lo, hi, sb, stt, st.z, st.lam, st.rho,
                                 ws["phi"], rnorm2=ws["rn2"], info=None, status=ws["status"],
                                 al_iter=1, max_newton=0, n_ls=N_LS, flags=_abi.ALQP_DUAL_UPDATE | pf,
                                 rho_scale=RHO_SCALE)
                    ws["primed"] = wsc if getattr(be, "last_variant", None) == "quad" else None
                else:
                    xn = true_next(st.z)
                    okw = self._obs_kwargs(dt, dev)
                    be.merit(dims, 1, st.z, xn, st.x0, st.lam, st.rho, Qd, q, lo, hi, sb, stt,
                             ws["phi"], ws["rn2"], **okw)
                    be.dual_update(dims, st.z, xn, st.x0, lo, hi, sb, stt, st.lam, st.rho, RHO_SCALE, **okw)
                    ws["primed"] = None   # lam/rho changed behind the workspace records' back
                if stream:
                    if linearize_once:
                        mean = self._global_mean(ws["rn2"].sqrt())
                        if prev_mean is not None and not mean < prev_mean:
                            break
                        prev_mean = mean
                    if self._global_max(st.rho) > self.rho_max:
                        break
            if stream and self._global_max(st.rho) > self.rho_max:
                st.status_flag = True
        # device-side exit counters (one read-back for the whole solve)
        if any(torch.is_tensor(v) for v in npa):
            vals = torch.stack([(v[1] if v.numel() == 3 else v.to(torch.float64).reshape(())) if torch.is_tensor(v)
                                else torch.tensor(float(v), dtype=torch.float64, device=st.z.device)
                                for v in npa]).tolist()
            npa = [int(round(v)) for v in vals]
            if min(npa) < 0:
                raise RuntimeError("mi_alqp: a grid barrier of the in-kernel exit test timed out (ALQP_EXIT_IN_KERNEL); "
                                   "construct the MPC with exit_in_kernel=False")
        st.newton_per_al = npa
        # (kept raw: `last_status` / `dyn_res_prev` are formed when read - two device kernels per call that a
        #  solve whose caller never looks at them does not pay for; at the reference's batch size a call is ~0.6 ms)
        self._status_raw = ws["status"]
        self.last_info = ws["info"]
        self._rn2_raw = ws["rn2"]
        if self.check_numerics is not None:
            n_piv = int((ws["info"] != 0).sum().item())
            n_bad = int((ws["status"] == 0).sum().item())
            if n_piv or n_bad:
                msg = (f"mi_alqp: {n_piv} of {B} instances met a non-positive pivot (penalty x conditioning beyond "
                       f"{dt}; modified-Cholesky step taken, see MPC.last_info), {n_bad} hold a non-finite iterate "
                       "(MPC.last_status)")
                if self.check_numerics == "raise":
                    raise FloatingPointError(msg)
                import warnings
                warnings.warn(msg, RuntimeWarning, stacklevel=3)
        if need_grad and F_last is not None:
            if "nlws" in ws:
                return "workspace", ws["nlws"], F_last, rho_last
            if use_qws:
                return "workspace", ws["qws"], F_last, rho_last
            return "packed", ws["factor"], F_last, rho_last
        return None
