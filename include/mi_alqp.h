/*
 * mi_alqp.h - C ABI of the MI355X-native batched augmented-Lagrangian MPC/QP solver.
 *
 * The reference (anonymous-author-918/deq-mpc-corl) has no native interface for this
 * path: its AL solver is eager PyTorch (qpth/AL_mpc.py, qpth/al_utils.py). The entry
 * points below are what a binding for that path would call; each one names the
 * reference code it replaces. INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. PyTorch's caching
 *     allocator); the library never allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*), so ordering is
 *     stream order; every call is re-entrant (no global state);
 *   - return value: 0 on success, <0 on bad arguments / unsupported dims / launch
 *     failure (ALQP_E_*). Numerical trouble is reported per instance in info[] / status[];
 *   - dtype suffix _f32 / _f64 is the arithmetic AND storage type of all real arrays;
 *   - batch-first row-major layouts, as the reference uses them:
 *       z   [B][T][n]        n = nx+nu, state first        (xu in al_utils.py)
 *       Qd,q[B][T][n]        diagonal of QuadCost.C and QuadCost.c (AL_mpc.py:250)
 *       F   [B][T-1][nx][n]  [A_t B_t]  (dynamics Jacobian, al_utils.py:242-248)
 *       c   [B][T-1][nx]     affine offset of LinDx-style dynamics
 *       xnext[B][T-1][nx]    f(x_t,u_t) evaluated by the caller (nonlinear mode)
 *       x0  [B][nx]
 *       lam [B][M]           M = T*nx + 2*T*nu; eq rows t*nx+i (t<T-1 dynamics, t=T-1 the
 *                            initial-state row), then ineq rows T*nx + t*2nu + j
 *                            (j<nu upper, j>=nu lower)  (al_utils.py:218-225, 293)
 *       rho [B]              (the reference's [B,1])
 *       u_lo,u_hi            element (b,t,j) at b*sb_u + t*st_u + j  (0 strides broadcast)
 *       factor[B][T][n(n+1)/2]  packed rows of X_t = L_tt^{-T} of the block-tridiagonal
 *                            Cholesky factor (row i holds X[i][i..n-1])
 */
#ifndef MI_ALQP_H
#define MI_ALQP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct AlqpDims {
    int B;   /* instances */
    int T;   /* horizon, >= 2 */
    int nx;  /* state dim */
    int nu;  /* control dim */
} AlqpDims;

/* obstacle rows of Obstacle_MPC (documented with alqp_newton_step_obs_* below) */
typedef struct AlqpObstacles {
    const void *pos;   /* DEVICE [B][T][nobs][3], the real type of the call */
    double radius;
    int nobs;          /* reference: 4 nearest of 40 (AL_mpc_custom.py:52-54, 112-115); 0: no obstacle rows */
    int state_estimator; /* 1: the state-estimator variant (qpth/al_utils_se.py): no initial-state rows (row block
                          T-1 of lam is ignored, x_0 gets no E'E term: :92-105, 186-200) and a cost gradient that is
                          zero on the given controls (:300-310) while H keeps diag(Q) there (:66-68), so du = 0 exactly.
                          The rest of that variant is data the caller prepares: B columns of F zeroed (:151), bounds
                          out of reach (the variant has no bound rows, AL_mpc.py:198) */
} AlqpObstacles;

/* flags for alqp_solve_lin_* */
#define ALQP_INIT_MERIT   1  /* evaluate merit(z) at the start of every AL iteration (al_utils.py:481) */
#define ALQP_DUAL_UPDATE  2  /* lam += rho*r, clamp, rho *= rho_scale after the Newton steps (AL_mpc.py:315-325) */
#define ALQP_SAVE_FACTOR  4  /* write the factor of the last executed Newton step to factor_out */
#define ALQP_WS_PRIMED    8  /* quad variant: the workspace still holds this solve's records from the previous
                               launch on it (same z, lam, Qd, q, c, bounds as the arrays passed now): skip the
                               copy-in pass. Only for back-to-back launches of one solve, see INTEGRATION.md */

#define ALQP_EXIT_IN_KERNEL 16 /* alqp_solve_lin: the reference's batch-global exit test of the Newton loop
                               (al_utils.py:486, 551-564: stop when ||r+|| over the WHOLE batch is below exit_tol or moved
                               by less than exit_tol relative) is taken INSIDE the launch: one cooperative launch does what
                               a merit launch + up to max_newton one-step launches + alqp_exit_test in between do, with a
                               grid-wide barrier and an ordered sum per Newton step. Needs exit_scratch; no trace, no
                               skip_flag. Returns ALQP_E_COOP when the grid cannot be co-resident (the caller then uses
                               the launch-per-step route). Single-GPU batches only: a batch sharded over ranks needs the
                               all-reduce between the steps. */

typedef struct AlqpParams {
    int al_iter;      /* AL outer iterations done by this call (AL_mpc.py:290) */
    int max_newton;   /* Newton steps per AL iteration, reference: 4 (al_utils.py:485) */
    int n_ls;         /* line-search candidates 2^-k, reference: 20 (al_utils.py:619); <= 20 */
    int flags;        /* ALQP_* */
    double rho_scale; /* reference: 10 (AL_mpc.py:325) */
    int variant;      /* kernel variant of alqp_solve_lin: 0 auto, 1 team (factor in LDS, no
                         workspace), 2 quad (4 lanes per instance, factor streamed through the
                         HBM workspace; no ALQP_SAVE_FACTOR) */
    const double *skip_flag; /* nullable DEVICE pointer: when *skip_flag != 0 at launch time the call
                         does nothing at all (ctl[0] of alqp_exit_test: the batch-global exit of the
                         reference's Newton loop taken on the device, no host round trip) */
    /* ALQP_EXIT_IN_KERNEL only (ignored otherwise): */
    double exit_tol;      /* reference: 1e-3 (al_utils.py:552, 560) */
    int *newton_counts;   /* nullable DEVICE [al_iter]: executed Newton steps per AL iteration (-1: barrier time-out) */
    double *exit_scratch; /* DEVICE, at least 2 * B + 2 doubles, zeroed before every call (arrival counter, partial sums,
                             time-out flag: a grid barrier that waited ~1 s gives up, newton_counts then holds -1) */
    int quad_stagger;     /* quad solve (variant 2 / auto at B >= 4096): start offset between the four wavefronts of a CU.
                             All wavefronts run the same sweeps; started together they queue on the CU's vector-memory
                             pipeline in lock step. 0 (default): automatic - a fifth of a sweep between neighbouring SIMDs
                             when the grid fills the SIMDs, nx + nu >= 12 and the launch holds enough Newton steps to
                             amortise the delay; < 0: off; > 0: that many units of ~1024 clocks. Timing only: results do
                             not depend on it. No counterpart in the reference (a scheduling control of this library). */
} AlqpParams;

/* optional per-step trace (all nullable, for tests): S = al_iter*max_newton steps */
typedef struct AlqpTrace {
    void *g;        /* [S][B][T][n] merit gradient */
    void *d;        /* [S][B][T][n] Newton update */
    void *phi;      /* [S][n_ls][B] candidate merits */
    void *phi_prev; /* [S][B] */
    int *k;         /* [S][B] chosen candidate */
    int *accept;    /* [S][B] */
} AlqpTrace;

#define ALQP_E_BADARG     (-1)
#define ALQP_E_UNSUPPORTED (-2)  /* (nx,nu) not instantiated or LDS budget exceeded */
#define ALQP_E_LAUNCH     (-3)
#define ALQP_E_COOP       (-4)  /* ALQP_EXIT_IN_KERNEL: the grid is too large for a cooperative launch */

/* 1 if (nx,nu) has a compiled kernel instance (the quad variant then runs any horizon). */
int alqp_supported(const AlqpDims *dims, int is_f64);
/* Per variant (AlqpParams.variant): 1 team - additionally the horizon's factor must fit the LDS image;
 * 2 quad - needs alqp_workspace_bytes() of workspace, no LDS. */
int alqp_supported_variant(const AlqpDims *dims, int is_f64, int variant);
/* LDS bytes one workgroup of the fused kernel uses (0 if unsupported). */
size_t alqp_lds_bytes(const AlqpDims *dims, int is_f64);
/* QP instances one 64-lane wavefront solves concurrently (team variant). */
int alqp_qps_per_wave(const AlqpDims *dims, int is_f64);
/* Bytes of device workspace the quad variant of alqp_solve_lin needs for these dims
 * (0 if (nx,nu) is not instantiated). The caller allocates; contents need not be kept
 * between calls. */
size_t alqp_workspace_bytes(const AlqpDims *dims, int is_f64);

/*
 * Fused solve on affine ("LinDx") dynamics x_{t+1} = F_t [x_t;u_t] + c_t.
 * Replaces MPC.al_solve (qpth/AL_mpc.py:260-339) + NewtonAL.forward
 * (qpth/al_utils.py:451-576) + line_search_newton (:618-642) for one batch:
 * al_iter x [merit init; max_newton x (gradient, block-tridiagonal Cholesky,
 * Newton step, 20-point line search); dual update + projection; rho *= rho_scale].
 *   in/out: z (start iterate -> solution), lam, rho, phi (merit carried between
 *           calls when ALQP_INIT_MERIT is off)
 *   out   : rnorm2[B] = sum r+(z)^2 (what the reference's batch-global exit test
 *           sums, al_utils.py:552), info[B] (0 or stage*n+pivot+1 of the first
 *           non-positive pivot), status[B] (1 = iterate finite, al_utils.py:545-549),
 *           factor_out (nullable), trace (nullable)
 *   workspace: device scratch of >= alqp_workspace_bytes() for the quad variant (nullable
 *           -> team variant)
 */
int alqp_solve_lin_f32(const AlqpDims *dims, const AlqpParams *prm, const void *Qd, const void *q,
                       const void *F, const void *c, const void *x0, const void *u_lo,
                       const void *u_hi, long sb_u, long st_u, void *z, void *lam, void *rho,
                       void *phi, void *rnorm2, int *info, unsigned char *status,
                       void *factor_out, const AlqpTrace *trace, void *workspace, size_t ws_bytes,
                       void *stream);
int alqp_solve_lin_f64(const AlqpDims *dims, const AlqpParams *prm, const void *Qd, const void *q,
                       const void *F, const void *c, const void *x0, const void *u_lo,
                       const void *u_hi, long sb_u, long st_u, void *z, void *lam, void *rho,
                       void *phi, void *rnorm2, int *info, unsigned char *status,
                       void *factor_out, const AlqpTrace *trace, void *workspace, size_t ws_bytes,
                       void *stream);

/*
 * One Newton direction for the nonlinear-caller mode: the caller evaluated
 * dx_jac(x,u) -> (xnext, F) in PyTorch (al_utils.py:233-248). Replaces
 * merit_grad_hessian (:80-123) + cholesky_ex/cholesky_solve (:510-515).
 *   out: d[B][T][n] = -H^{-1} g, g_out (nullable), factor_out (nullable), info[B]
 */
int alqp_newton_step_f32(const AlqpDims *dims, const void *z, const void *xnext, const void *F,
                         const void *x0, const void *lam, const void *rho, const void *Qd,
                         const void *q, const void *u_lo, const void *u_hi, long sb_u, long st_u,
                         void *d_out, void *g_out, void *factor_out, int *info, void *stream);
int alqp_newton_step_f64(const AlqpDims *dims, const void *z, const void *xnext, const void *F,
                         const void *x0, const void *lam, const void *rho, const void *Qd,
                         const void *q, const void *u_lo, const void *u_hi, long sb_u, long st_u,
                         void *d_out, void *g_out, void *factor_out, int *info, void *stream);

/*
 * The same Newton direction by the quad kernels (16 instances per wavefront; use it once the batch fills the
 * chip, B >= 4096): the per-stage factor is streamed through - and left in - `workspace`
 * (alqp_workspace_bytes()), where alqp_backward_ws_* finds it for NewtonAL.backward. No packed factor_out.
 */
int alqp_newton_step_ws_f32(const AlqpDims *dims, const void *z, const void *xnext, const void *F,
                            const void *x0, const void *lam, const void *rho, const void *Qd, const void *q,
                            const void *u_lo, const void *u_hi, long sb_u, long st_u, void *workspace,
                            size_t ws_bytes, void *d_out, void *g_out, int *info, void *stream);
int alqp_newton_step_ws_f64(const AlqpDims *dims, const void *z, const void *xnext, const void *F,
                            const void *x0, const void *lam, const void *rho, const void *Qd, const void *q,
                            const void *u_lo, const void *u_hi, long sb_u, long st_u, void *workspace,
                            size_t ws_bytes, void *d_out, void *g_out, int *info, void *stream);

/*
 * Merit of K stacked candidates (al_utils.py:52-77 with the [K,B,T,n] broadcast of
 * :56-70): zc[K][B][T][n], xnext[K][B][T-1][nx] -> phi[K][B], rnorm2[K][B] (nullable).
 */
int alqp_merit_f32(const AlqpDims *dims, int K, const void *zc, const void *xnext, const void *x0,
                   const void *lam, const void *rho, const void *Qd, const void *q,
                   const void *u_lo, const void *u_hi, long sb_u, long st_u, void *phi,
                   void *rnorm2, void *stream);
int alqp_merit_f64(const AlqpDims *dims, int K, const void *zc, const void *xnext, const void *x0,
                   const void *lam, const void *rho, const void *Qd, const void *q,
                   const void *u_lo, const void *u_hi, long sb_u, long st_u, void *phi,
                   void *rnorm2, void *stream);

/*
 * Line-search decision + update (al_utils.py:634-641): first argmin over phi[n_ls][B],
 * accept iff strictly below phi_prev; z <- accept ? z + 2^-k d : z (in place);
 * phi_prev <- phi_min regardless (al_utils.py:569). k_out/accept_out nullable.
 */
int alqp_linesearch_pick_f32(const AlqpDims *dims, int n_ls, const void *phi, void *phi_prev,
                             const void *d, void *z, int *k_out, int *accept_out, void *stream);
int alqp_linesearch_pick_f64(const AlqpDims *dims, int n_ls, const void *phi, void *phi_prev,
                             const void *d, void *z, int *k_out, int *accept_out, void *stream);

/*
 * The whole line search of a Newton step in one launch (nonlinear-caller mode at scale): merits of the
 * n_ls candidates z + 2^-k d, given x_next of every candidate from the caller's dynamics
 * (xnext_all[n_ls][B][T-1][nx], what the reference's 20-fold replicated dx call returns,
 * al_utils.py:56-70, 629-633), then the decision and update of alqp_linesearch_pick. z, d are read
 * once; the stacked candidates the reference materialises are never formed for the merit.
 *   in/out: z, phi_prev; rnorm2[B] (nullable) <- sum r+^2 of the chosen candidate when accepted
 *   out   : phi_all[n_ls][B] (nullable), k_out, accept_out (nullable); obs nullable (Obstacle_MPC rows)
 */
int alqp_merit_pick_f32(const AlqpDims *dims, int n_ls, const void *d, const void *xnext_all, const void *x0,
                        const void *lam, const void *rho, const void *Qd, const void *q, const void *u_lo,
                        const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs, void *z,
                        void *phi_prev, void *rnorm2, void *phi_all, int *k_out, int *accept_out, void *stream);
int alqp_merit_pick_f64(const AlqpDims *dims, int n_ls, const void *d, const void *xnext_all, const void *x0,
                        const void *lam, const void *rho, const void *Qd, const void *q, const void *u_lo,
                        const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs, void *z,
                        void *phi_prev, void *rnorm2, void *phi_all, int *k_out, int *accept_out, void *stream);

/*
 * Dual update + projection + penalty growth (AL_mpc.py:315-317, 325) given
 * xnext = f(x_t,u_t) at the final iterate; lam, rho updated in place.
 */
int alqp_dual_update_f32(const AlqpDims *dims, const void *z, const void *xnext, const void *x0,
                         const void *u_lo, const void *u_hi, long sb_u, long st_u, void *lam,
                         void *rho, double rho_scale, void *stream);
int alqp_dual_update_f64(const AlqpDims *dims, const void *z, const void *xnext, const void *x0,
                         const void *u_lo, const void *u_hi, long sb_u, long st_u, void *lam,
                         void *rho, double rho_scale, void *stream);

/*
 * Backward of the implicit layer (NewtonAL.backward, al_utils.py:578-615):
 * w = -H^{-1} gbar with the saved factor; q_grad = w, Qd_grad = w * z_final.
 * rho is the penalty the factor was built with.
 */
int alqp_backward_f32(const AlqpDims *dims, const void *factor, const void *F, const void *rho,
                      const void *z_final, const void *gbar, void *q_grad, void *Qd_grad,
                      void *stream);
int alqp_backward_f64(const AlqpDims *dims, const void *factor, const void *F, const void *rho,
                      const void *z_final, const void *gbar, void *q_grad, void *Qd_grad,
                      void *stream);

/*
 * Same backward pass, with the factor taken from the workspace a quad-variant
 * alqp_solve_lin_* call left behind (the caller must have kept that workspace untouched:
 * give such solves a dedicated workspace). The workspace's y/d slots are overwritten.
 */
int alqp_backward_ws_f32(const AlqpDims *dims, void *workspace, size_t ws_bytes, const void *F,
                         const void *rho, const void *z_final, const void *gbar, void *q_grad,
                         void *Qd_grad, void *stream);
int alqp_backward_ws_f64(const AlqpDims *dims, void *workspace, size_t ws_bytes, const void *F,
                         const void *rho, const void *z_final, const void *gbar, void *q_grad,
                         void *Qd_grad, void *stream);

/* Library/ABI version, bumped when a signature changes. */
/*
 * The reference leaves its Newton loop on a BATCH-GLOBAL test (al_utils.py:486,551-564):
 * new = ||r_+||_F over the whole batch; stop when new < tol or |old - new| / new < tol.
 * This entry point takes that decision on the device so that the host can enqueue all Newton-step
 * launches of an AL iteration without synchronising: sumsq[0] = sum_b rnorm2[b] (the caller
 * reduces, and all-reduces over ranks when the batch is sharded), ctl = {done, steps, old_norm}.
 *   mode 0: ctl <- {0, 0, sqrt(sumsq)}                       (before the first step)
 *   mode 1: if (!done) { steps++; new = sqrt(sumsq); done = the test above; else old = new }
 * Pass ctl as AlqpParams.skip_flag of the following launches.
 */
int alqp_exit_test(const double *sumsq, double *ctl, int mode, double tol, void *stream);

/*
 * Dynamics provider for the nonlinear-caller mode, pendulum1l: replaces the reference's
 * CasADi-generated pair dynamics(q, qdot, tau, h) / derivatives(...) (deqmpc/my_envs/pendulum1l/
 * src/dynamics.cpp:13-47, generated_dynamics.c:55-140, generated_derivatives.c:52-222) with one
 * kernel that emits the MPC's packed operands: one RK4 step of theta'' = 4 tau - 19.62 sin(theta)
 * and its exact Jacobian.
 *   in : x[K][2] = (theta, omega), u[K][1] = tau, step length h, or h_pt[K] per point when not
 *        NULL (the reference passes a [bsz,1] tensor filled with dt, my_envs/dynamics.py:58)
 *   out: xnext[K][2] (nullable), F[K][2][3] = [A | B] (nullable: dynamics only)
 */
int alqp_dyn_pendulum1l_f32(long K, const void *x, const void *u, double h, const void *h_pt, void *xnext, void *F, void *stream);
int alqp_dyn_pendulum1l_f64(long K, const void *x, const void *u, double h, const void *h_pt, void *xnext, void *F, void *stream);

/*
 * Dynamics provider, cartpole1l (deqmpc/my_envs/cartpole1l/src/dynamics.cpp:13-47 and the two
 * generated files): one RK4 step of  M(th) q'' = tau - (sin(th) th'^2, 0) + (0, 9.81 sin(th)),
 * M = [[11, -cos th], [-cos th, 2]], q = (cart x, pole angle th; th = 0 upright), and its Jacobian.
 *   in : x[K][4] = (q, qdot), tau[K][2], h or h_pt[K]
 *   out: xnext[K][4] (nullable), J[K][4][6] = d xnext / d(q, qdot, tau) (nullable)
 */
int alqp_dyn_cartpole1l_f32(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream);
int alqp_dyn_cartpole1l_f64(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream);
/* cartpole1l_v2 (deqmpc/my_envs/cartpole1l_v2/src: a package the reference ships but does not import,
 * my_envs/cartpole.py:35): the same model with M = [[0.7, -0.1 cos th], [-0.1 cos th, 0.05]], Coriolis 0.1 sin(th) th'^2,
 * gravity 0.981 sin(th); same arrays as alqp_dyn_cartpole1l. */
int alqp_dyn_cartpole1l_v2_f32(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream);
int alqp_dyn_cartpole1l_v2_f64(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream);

/* cartpole2l (deqmpc/my_envs/cartpole2l/src): q = (cart x, th1, th2 relative to link 1); x[K][6], tau[K][3],
 * xnext[K][6] (nullable), J[K][6][9] = d xnext / d(q, qdot, tau) (nullable). Model: DESIGN.md section 9. */
int alqp_dyn_cartpole2l_f32(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream);
int alqp_dyn_cartpole2l_f64(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream);

/*
 * Nonlinear fused solve: the whole AL solve of MPC.al_solve (AL_mpc.py:260-339) with NONLINEAR
 * dynamics in ONE launch, for the robots whose dynamics model is compiled into the library
 * (dyn_id 1 = pendulum1l (nx=2, nu=1), 2 = cartpole1l (nx=4, nu=1, tau = (u, 0)), 3 = cartpole2l (nx=6, nu=1,
 * tau = (u, 0, 0)), 4 = cartpole1l_v2 (as 2, the second package's constants); dyn_h = step length).
 * Replaces the PyTorch round trips of NewtonAL.forward (al_utils.py:451-576): every Newton step
 * re-linearises on the device (dx_jac, :503), the 20 line-search candidates and the dual update
 * are evaluated with the true dynamics (:623-633, AL_mpc.py:315). Arguments as alqp_solve_lin
 * without F, c; n_ls must be 20; workspace of alqp_workspace_bytes_nonlin() bytes; no
 * ALQP_SAVE_FACTOR / ALQP_WS_PRIMED. Exit mode "fixed" (max_newton steps per AL iteration).
 */
size_t alqp_workspace_bytes_nonlin(const AlqpDims *dims, int is_f64);
int alqp_solve_nonlin_f32(const AlqpDims *dims, const AlqpParams *prm, int dyn_id, double dyn_h, const void *Qd,
                          const void *q, const void *x0, const void *u_lo, const void *u_hi, long sb_u, long st_u,
                          void *z, void *lam, void *rho, void *phi, void *rnorm2, int *info, unsigned char *status,
                          void *workspace, size_t ws_bytes, void *stream);
int alqp_solve_nonlin_f64(const AlqpDims *dims, const AlqpParams *prm, int dyn_id, double dyn_h, const void *Qd,
                          const void *q, const void *x0, const void *u_lo, const void *u_hi, long sb_u, long st_u,
                          void *z, void *lam, void *rho, void *phi, void *rnorm2, int *info, unsigned char *status,
                          void *workspace, size_t ws_bytes, void *stream);

/*
 * ---- Obstacle inequalities (Obstacle_MPC, qpth/AL_mpc_custom.py:22-135; al_utils.py:313-323, 351-388) ----
 * `nobs` spheres per stage add the rows  c_k = radius^2 - |x_t[0:3] - pos_k|^2 <= 0  behind the stage's
 * 2 nu bound rows: lam [B][M], M = T*nx + T*(2*nu + nobs), inequality row T*nx + t*(2*nu + nobs) + j with
 * j < nu upper, j < 2 nu lower, then obstacle j - 2 nu (the reference's stage-major order, al_utils.py:376).
 * The Newton step gets the rank-<=nobs Gauss-Newton update 4 rho (p - o_k)(p - o_k)' of the position corner
 * of H_tt for the rows with c_k >= 0 and the gradient term (lam_k + rho max(c_k, 0)) (-2)(p - o_k).
 * These are the nonlinear-caller building blocks with obstacles (the reference only reaches Obstacle_MPC
 * with PyTorch-coded dynamics); arguments as their plain twins. nx >= 3 required.
 */

/* The quad-variant Newton step (alqp_newton_step_ws: 16 instances per wavefront, factor left in the workspace records for
 * alqp_backward_ws) with the same extra rows: what Obstacle_MPC and the state-estimator variant run at B >= 4096. */
int alqp_newton_step_ws_obs_f32(const AlqpDims *dims, const void *z, const void *xnext, const void *F,
                                const void *x0, const void *lam, const void *rho, const void *Qd, const void *q,
                                const void *u_lo, const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs,
                                void *workspace, size_t ws_bytes, void *d_out, void *g_out, int *info, void *stream);
int alqp_newton_step_ws_obs_f64(const AlqpDims *dims, const void *z, const void *xnext, const void *F,
                                const void *x0, const void *lam, const void *rho, const void *Qd, const void *q,
                                const void *u_lo, const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs,
                                void *workspace, size_t ws_bytes, void *d_out, void *g_out, int *info, void *stream);
int alqp_newton_step_obs_f32(const AlqpDims *dims, const void *z, const void *xnext, const void *F,
                             const void *x0, const void *lam, const void *rho, const void *Qd, const void *q,
                             const void *u_lo, const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs,
                             void *d_out, void *g_out, void *factor_out, int *info, void *stream);
int alqp_newton_step_obs_f64(const AlqpDims *dims, const void *z, const void *xnext, const void *F,
                             const void *x0, const void *lam, const void *rho, const void *Qd, const void *q,
                             const void *u_lo, const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs,
                             void *d_out, void *g_out, void *factor_out, int *info, void *stream);
int alqp_merit_obs_f32(const AlqpDims *dims, int K, const void *zc, const void *xnext, const void *x0,
                       const void *lam, const void *rho, const void *Qd, const void *q, const void *u_lo,
                       const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs, void *phi, void *rnorm2,
                       void *stream);
int alqp_merit_obs_f64(const AlqpDims *dims, int K, const void *zc, const void *xnext, const void *x0,
                       const void *lam, const void *rho, const void *Qd, const void *q, const void *u_lo,
                       const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs, void *phi, void *rnorm2,
                       void *stream);
int alqp_dual_update_obs_f32(const AlqpDims *dims, const void *z, const void *xnext, const void *x0,
                             const void *u_lo, const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs,
                             void *lam, void *rho, double rho_scale, void *stream);
int alqp_dual_update_obs_f64(const AlqpDims *dims, const void *z, const void *xnext, const void *x0,
                             const void *u_lo, const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs,
                             void *lam, void *rho, double rho_scale, void *stream);

/*
 * ---- Interior-point QP solve (the `--solver_type ip` path) -------------------------------------------
 * Replaces qp.DenseQPFunction (qpth/qp.py:187-270) = pdipm_b_LU.forward / solve_kkt
 * (qpth/solvers/pdipm/batch_LU.py:29-244) on the QP that qp_wrapper.MPC.single_qp assembles
 * (qpth/qp_wrapper.py:295-321, 612-653):
 *     min 1/2 z'Qz + p'z  s.t.  u_lower <= u_t <= u_upper,  F_t tau_t + f_t = x_{t+1},  x_0 = x0
 * with DIAGONAL Q (what policies.Tracking_MPC builds). Same regularised KKT system (KKTeps = 1e-7), same
 * refinement step, same predictor-corrector iteration; no dense matrix is formed (csrc/alqp_ipm.hip).
 *   Cd, c  diag(C_t) and c_t, element (t, b, k) at t*sC_t + b*sC_b + k      (time-major [T,B,n]: sC_t = B*n, sC_b = n)
 *   F      [.][nx][n] blocks at t*sF_t + b*sF_b;  f [.][nx] at t*sf_t + b*sf_b
 *   x0 [B][nx]; u_hi, u_lo [nu]
 *   zhat [B][T*n]; nus [B][T*nx] (dynamics rows t*nx+i, then the initial-state rows);
 *   lams, slacks [B][2*T*nu] (upper-bound rows t*nu+j, then lower-bound rows T*nu + t*nu+j)   - the
 *   reference's orderings (qp_wrapper.py:194-207)
 *   ry_ext nullable [B][T*nx]: the caller's equality residual at the current iterate (the reference
 *   evaluates the TRUE dynamics there: dyn_res callback, qp_wrapper.py:306, batch_LU.py:95); NULL: A z - b
 *   workspace: alqp_ipm_workspace_bytes(); it carries the iterate between launches of one solve.
 * flags: ALQP_IPM_INIT  initial point (batch_LU.py:44-81)
 *        ALQP_IPM_LOOP  max_iter x [residuals + best iterate ; predictor-corrector step] in this launch
 *                       (exit mode "fixed": no batch-global exit rule, best iterate per instance)
 *        ALQP_IPM_RESID residuals + best-iterate update of iteration iter0 (improved[b], resid[b] = best
 *                       residual, mu[b] for the reference's batch-global exit rule, :147-151, taken by the host)
 *        ALQP_IPM_STEP  one predictor-corrector step (:153-197)
 *        ALQP_IPM_FINAL copy the best iterate to zhat / nus / lams / slacks
 * info[b]: 0, or block*nx + pivot + 1 of the first non-positive pivot of the Schur complement (sticky).
 */
#define ALQP_IPM_INIT  1
#define ALQP_IPM_RESID 2
#define ALQP_IPM_STEP  4
#define ALQP_IPM_LOOP  8
#define ALQP_IPM_FINAL 16

typedef struct AlqpIpmParams {
    int flags;       /* ALQP_IPM_* */
    int max_iter;    /* iterations of an ALQP_IPM_LOOP launch (reference: maxIter = 20, qp.py:203) */
    int iter0;       /* index of the (first) iteration this launch works on */
    double kkt_eps;  /* reference: 1e-7 (batch_LU.py:43) */
    int variant;     /* ALQP_IPM_VARIANT_*: which kernel. Results agree to rounding; a per-call argument, no process state */
} AlqpIpmParams;
/* 0: automatic - the register/LDS-resident kernel (one QP per wavefront, iterate and directions in registers, F and
 *    the Schur factor in LDS; HBM traffic = the inputs once + the best iterate when it improves) whenever the horizon
 *    fits its register slots (T <= 20), else the size-generic kernel with automatic factor placement;
 * 1 / 2: the size-generic kernel (vectors in the workspace) with the Schur factor in LDS / in the workspace;
 * 3: the register/LDS-resident kernel, ALQP_E_UNSUPPORTED when the problem does not fit it. */
#define ALQP_IPM_VARIANT_AUTO 0
#define ALQP_IPM_VARIANT_GENERIC_LDS 1
#define ALQP_IPM_VARIANT_GENERIC_WS 2
#define ALQP_IPM_VARIANT_RESIDENT 3

size_t alqp_ipm_workspace_bytes(const AlqpDims *dims, int is_f64);
int alqp_ipm_solve_f32(const AlqpDims *dims, const AlqpIpmParams *prm, const void *Cd, const void *c,
                       const void *F, const void *f, const void *x0, const void *u_hi, const void *u_lo,
                       long sC_t, long sC_b, long sF_t, long sF_b, long sf_t, long sf_b, void *workspace,
                       size_t ws_bytes, const void *ry_ext, void *zhat, void *nus, void *lams, void *slacks,
                       void *resid, void *mu, int *iter_best, int *improved, int *info, void *stream);
int alqp_ipm_solve_f64(const AlqpDims *dims, const AlqpIpmParams *prm, const void *Cd, const void *c,
                       const void *F, const void *f, const void *x0, const void *u_hi, const void *u_lo,
                       long sC_t, long sC_b, long sF_t, long sF_b, long sf_t, long sf_b, void *workspace,
                       size_t ws_bytes, const void *ry_ext, void *zhat, void *nus, void *lams, void *slacks,
                       void *resid, void *mu, int *iter_best, int *improved, int *info, void *stream);
/*
 * Backward of the QP layer (DenseQPFunction.backward, qp.py:238-252): one KKT solve, without
 * regularisation, at the returned lams / slacks: K (dx, ds, dlam, dnu) = -(gbar, 0, 0, 0).
 * The caller forms the reference's gradient formulas (:254-268) from dx, dlam, dnu. variant: ALQP_IPM_VARIANT_*.
 */
int alqp_ipm_backward_f32(const AlqpDims *dims, const void *Cd, const void *F, long sC_t, long sC_b, long sF_t,
                          long sF_b, const void *lams, const void *slacks, const void *gbar, void *workspace,
                          size_t ws_bytes, void *dx, void *dlam, void *dnu, int *info, int variant, void *stream);
int alqp_ipm_backward_f64(const AlqpDims *dims, const void *Cd, const void *F, long sC_t, long sC_b, long sF_t,
                          long sF_b, const void *lams, const void *slacks, const void *gbar, void *workspace,
                          size_t ws_bytes, void *dx, void *dlam, void *dnu, int *info, int variant, void *stream);

/*
 * ---- Dynamics + Jacobian providers of the reference's torch-coded robots -----------
 * RexQuadrotor (deqmpc/rex_quadrotor.py:98-144; x = (r, MRP, body velocity, body rate) [12], u [4]) and FlyingCartpole
 * (deqmpc/flying_cartpole2d.py:81-148; x = (r, MRP, theta, v, w, theta') [14], u [4]): one RK4 step of length h,
 * x+ [K][nx] and F = [A | B] = d x+ / d (x, u) [K][nx][nx+nu] (either output may be NULL). They replace the
 * reference's `dynamics` TorchScript module and its replicate-nx-times autograd Jacobian (rex_quadrotor.py:136-144).
 * PARITY UNPINNED: the reference files import `rexquad_utils`, which is not in the tree; mrp2quat, quatrot and
 * w2pdotkinematics_mrp are restated from their standard definitions (csrc/alqp_dyn_rigid.hip, oracle/rigid_py.py).
 * The struct carries the constructor values of the reference classes (the Python side fills it, rounding through
 * float32 where the reference builds float32 tensors).
 */
typedef struct AlqpRigidParams {
    double mass;        /* quadrotor: mass; flying cartpole: mass_q + mass_p */
    double J[9], Jinv[9];
    double g[3];
    double motor_dist, kf, bf, km;
    double act_scale;   /* 100 (quadrotor) / 10 (flying cartpole) */
    double u_hover;     /* flying cartpole: u <- act_scale (u + u_hover); 0 for the quadrotor */
    double pend_L;      /* flying cartpole: pendulum length */
    double ss[12];      /* motor arm directions [4][3], normalised */
    double bf_force;    /* quadrotor: Bf_z = 4 bf (rex_quadrotor.py:31-32) */
} AlqpRigidParams;
int alqp_dyn_rexquadrotor_f32(long K, const AlqpRigidParams *p, const void *x, const void *u, double h, void *xnext, void *F, void *stream);
int alqp_dyn_rexquadrotor_f64(long K, const AlqpRigidParams *p, const void *x, const void *u, double h, void *xnext, void *F, void *stream);
int alqp_dyn_flyingcartpole_f32(long K, const AlqpRigidParams *p, const void *x, const void *u, double h, void *xnext, void *F, void *stream);
int alqp_dyn_flyingcartpole_f64(long K, const AlqpRigidParams *p, const void *x, const void *u, double h, void *xnext, void *F, void *stream);

int alqp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_ALQP_H */
