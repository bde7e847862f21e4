/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 *
 * CPU restatement (plain C) of the reference's augmented-Lagrangian MPC/QP
 * solve, `qpth.AL_mpc.MPC.al_solve` + `qpth.al_utils.NewtonAL`, for the case the
 * hot path covers: diagonal quadratic cost, box bounds on the controls and a
 * dynamics provider given either as affine data (F, c) ("LinDx" mode) or as
 * pre-evaluated x_next / Jacobians (nonlinear-caller mode building blocks).
 *
 * Included twice by alqp_oracle.c with REAL = double / float and SFX = f64 / f32.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * Parity pin: tests/test_oracle_golden.py checks every function below against the
 * fixtures in tests/golden/ that tools/gen_golden.py produced by running the
 * reference itself.
 *
 * Conventions: z[B][T][n] (n = nx+nu, x first), lam[B][M] with M = T*nx + 2*T*nu:
 * equality rows t*nx+i (t < T-1: dynamics row of stage t; t = T-1: the initial
 * state row) then inequality rows neq + t*2nu + j (j < nu upper, j >= nu lower)
 * - qpth/al_utils.py:218-225 (eq) and :293,395 (ineq).
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

/* ---- obstacle rows (Obstacle_MPC, qpth/AL_mpc_custom.py:22-135; al_utils.py:313-323, 351-388) ------
 * Optional: after orc_set_obstacles(pos[B][T][nobs][3], nobs, radius) every stage carries `nobs` more
 * inequality rows  c_k = radius^2 - |x_t[0:3] - pos_k|^2 <= 0  behind its 2 nu bound rows (stage-major
 * row order: upper, lower, obstacles - al_utils.py:376), with Jacobian -2 (x_t[0:3] - pos_k)' on the
 * first three states. nobs = 0 (default) is the plain problem. Test infrastructure: process-global. */
static const REAL *FN(g_obs) = 0;
static int FN(g_nobs) = 0;
static REAL FN(g_obs_r2) = 0;
void FN(orc_set_obstacles)(const REAL *pos, int nobs, double radius)
{
    FN(g_obs) = pos; FN(g_nobs) = pos ? nobs : 0; FN(g_obs_r2) = (REAL)(radius * radius);
}
/* State-estimator variant (qpth/al_utils_se.py:92-105, 186-200, 300-310): no initial-state rows - with the flag
 * set the row block T-1 is identically zero (residual, multiplier update) and x_0 gets no E'E term - and the cost
 * gradient on the (given) controls is zero while the Hessian keeps diag(Q) there (:66-68). */
static int FN(g_no_init) = 0;
void FN(orc_set_state_estimator)(int flag) { FN(g_no_init) = flag; }
#define NOBS (FN(g_nobs))
#define NINEQ_T (2 * nu + NOBS)                      /* inequality rows per stage */
#define OBS_OF(b) (FN(g_obs) ? FN(g_obs) + (long)(b) * T * NOBS * 3 : (const REAL *)0)

/* ---- residuals ------------------------------------------------------------ */

/* Equality + inequality residuals of one instance, given x_next = f(x_t,u_t).
 * qpth/al_utils.py:209-226 (dyn_res_eq), :288-326 (dyn_res_ineq). */
static void FN(residual_one)(int T, int nx, int nu, const REAL *z, const REAL *xnext,
                             const REAL *x0, const REAL *ulo, const REAL *uhi, long st_u,
                             REAL *res /*[M]*/, const REAL *obs /*[T][nobs][3] or NULL*/)
{
    int n = nx + nu, neq = T * nx;
    for (int t = 0; t < T - 1; ++t)
        for (int i = 0; i < nx; ++i)
            res[t * nx + i] = z[(t + 1) * n + i] - xnext[t * nx + i];
    for (int i = 0; i < nx; ++i)
        res[(T - 1) * nx + i] = FN(g_no_init) ? (REAL)0 : z[i] - x0[i];
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < nu; ++j) {
            REAL u = z[t * n + nx + j];
            res[neq + t * NINEQ_T + j] = u - uhi[t * st_u + j];
            res[neq + t * NINEQ_T + nu + j] = -u + ulo[t * st_u + j];
        }
    if (obs)
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < NOBS; ++k) {
                REAL d2 = 0;
                for (int a = 0; a < 3; ++a) {
                    REAL dv = z[t * n + a] - obs[(t * NOBS + k) * 3 + a];
                    d2 += dv * dv;
                }
                res[neq + t * NINEQ_T + 2 * nu + k] = FN(g_obs_r2) - d2;
            }
}

/* x_next for affine dynamics: F_t [x_t;u_t] + c_t  (the synthetic `dx`). */
static void FN(affine_next_one)(int T, int nx, int nu, const REAL *z, const REAL *F,
                                const REAL *c, REAL *xnext)
{
    int n = nx + nu;
    for (int t = 0; t < T - 1; ++t)
        for (int i = 0; i < nx; ++i) {
            REAL s = 0;
            for (int k = 0; k < n; ++k)
                s += F[(t * nx + i) * n + k] * z[t * n + k];
            xnext[t * nx + i] = s + c[t * nx + i];
        }
}

/* merit = cost + lam.r + rho/2 |r+|^2 ; qpth/al_utils.py:73-77 (constant f omitted
 * because compute_cost is called without it, :73). Returns sum r+^2 via *rp2. */
static REAL FN(merit_one)(int T, int nx, int nu, const REAL *z, const REAL *res,
                          const REAL *lam, REAL rho, const REAL *Qd, const REAL *q, REAL *rp2)
{
    int n = nx + nu, neq = T * nx, M = neq + T * NINEQ_T;
    REAL cost = 0;
    for (int t = 0; t < T; ++t) {
        REAL a = 0, b = 0;
        for (int k = 0; k < n; ++k) {
            REAL v = z[t * n + k];
            a += v * Qd[t * n + k] * v;
            b += q[t * n + k] * v;
        }
        cost += (REAL)0.5 * a + b;
    }
    REAL lr = 0, sq = 0;
    for (int r = 0; r < M; ++r) {
        REAL v = res[r];
        REAL vc = (r < neq) ? v : (v > 0 ? v : 0);
        sq += vc * vc;
        lr += lam[r] * v;
    }
    if (rp2) *rp2 = sq;
    return cost + (REAL)0.5 * rho * sq + lr;
}

/* ---- gradient + block-tridiagonal Hessian --------------------------------- */

/* SURVEY.md 8a closed form of qpth/al_utils.py:80-123 (merit_grad_hessian) with the
 * Jacobian structure of :269-284 (eq) and :390-404 (ineq):
 *   H_tt    = diag(Q_t) + rho (E'E + [t<T-1] F_t'F_t + diag(0, a_t^+ + a_t^-))
 *   H_t+1,t = -rho E'F_t
 *   g       = Q z + q + J'lam + rho J+' r+        (lam multiplies the UNMASKED J, :115)
 * active  <=> res >= 0 (:397), not lam + rho res >= 0.
 * Hd[T][n][n] (full symmetric blocks), Hs[T-1][n][n] (block (t+1,t); rows >= nx are 0). */
static void FN(grad_hess_one)(int T, int nx, int nu, const REAL *z, const REAL *res,
                              const REAL *F, const REAL *lam, REAL rho, const REAL *Qd,
                              const REAL *q, REAL *g, REAL *Hd, REAL *Hs, const REAL *obs)
{
    int n = nx + nu, neq = T * nx;
    for (int t = 0; t < T; ++t) {
        REAL *H = Hd + (long)t * n * n;
        for (int i = 0; i < n * n; ++i) H[i] = 0;
        for (int k = 0; k < n; ++k) {
            H[k * n + k] = Qd[t * n + k];
            g[t * n + k] = Qd[t * n + k] * z[t * n + k] + q[t * n + k];
        }
        /* E'E : x_t enters row block t-1 (t>=1) or the init rows (t=0) with identity */
        for (int i = 0; i < nx; ++i) {
            if (t == 0 && FN(g_no_init)) break;
            H[i * n + i] += rho;
            int row = (t == 0) ? (T - 1) * nx + i : (t - 1) * nx + i;
            g[t * n + i] += lam[row] + rho * res[row];
        }
        if (t < T - 1) {
            const REAL *Ft = F + (long)t * nx * n;
            for (int a = 0; a < n; ++a)
                for (int b = 0; b < n; ++b) {
                    REAL s = 0;
                    for (int r = 0; r < nx; ++r) s += Ft[r * n + a] * Ft[r * n + b];
                    H[a * n + b] += rho * s;
                }
            for (int a = 0; a < n; ++a) {
                REAL s = 0;
                for (int r = 0; r < nx; ++r)
                    s += Ft[r * n + a] * (lam[t * nx + r] + rho * res[t * nx + r]);
                g[t * n + a] -= s;
            }
            REAL *S = Hs + (long)t * n * n;
            for (int i = 0; i < n * n; ++i) S[i] = 0;
            for (int r = 0; r < nx; ++r)
                for (int b = 0; b < n; ++b) S[r * n + b] = -rho * Ft[r * n + b];
        }
        for (int j = 0; j < nu; ++j) {
            int ru = neq + t * NINEQ_T + j, rl = ru + nu;
            REAL vu = res[ru], vl = res[rl];
            REAL au = vu >= 0 ? 1 : 0, al = vl >= 0 ? 1 : 0;
            H[(nx + j) * n + nx + j] += rho * (au + al);
            g[t * n + nx + j] += (lam[ru] + rho * (vu > 0 ? vu : 0))
                               - (lam[rl] + rho * (vl > 0 ? vl : 0));
            if (FN(g_no_init)) g[t * n + nx + j] = 0;   /* given controls: no gradient (al_utils_se.py:300-310) */
        }
        /* obstacle rows: J_k = -2 (p - o_k)' on the first three states; g += (lam_k + rho c_k+) J_k',
         * H += rho J_k'J_k for the rows with c_k >= 0 (al_utils.py:373-386, 113-120) */
        if (obs)
            for (int k = 0; k < NOBS; ++k) {
                int rk = neq + t * NINEQ_T + 2 * nu + k;
                REAL ck = res[rk], dv[3];
                for (int a = 0; a < 3; ++a) dv[a] = z[t * n + a] - obs[(t * NOBS + k) * 3 + a];
                REAL coef = lam[rk] + rho * (ck > 0 ? ck : 0);
                for (int a = 0; a < 3; ++a) g[t * n + a] += coef * (-2 * dv[a]);
                if (ck >= 0)
                    for (int a = 0; a < 3; ++a)
                        for (int c = 0; c < 3; ++c) H[a * n + c] += rho * 4 * dv[a] * dv[c];
            }
    }
}

/* ---- block-tridiagonal Cholesky (the algorithm the HIP kernel implements) -- */

/* In-place lower Cholesky of an n x n block; returns index+1 of the first
 * non-positive pivot (0 = ok), like cholesky_ex's info.
 * Failure policy ("non-positive pivots"): the reference's cholesky_ex stops at
 * such a pivot and its cholesky_solve then uses the half-finished factor (al_utils.py:510-515; the
 * linalg.solve fallback only fires on NaN/Inf, :517) - an undefined direction that only the line
 * search's strict-decrease test guards. There is nothing to restate; the kernels and this oracle
 * take |p| for a non-positive pivot p (a modified Cholesky: the factor of H + E, E >= 0 diagonal),
 * report the instance in info[] and leave the guarding to the same line search. */
static int FN(chol_block)(int n, REAL *A)
{
    int info = 0;
    for (int j = 0; j < n; ++j) {
        REAL d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0) && !info) info = j + 1;
        REAL l = SQRT(d < 0 ? -d : d);
        A[j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            REAL s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / l;
        }
        for (int k = j + 1; k < n; ++k) A[j * n + k] = 0;
    }
    return info;
}

/* Factor: L[T][n][n] (lower), W[T-1][nx][n] with W_t = H_{t+1,t}[0:nx] L_tt^{-T};
 * H_{t+1,t+1}[xx] -= W_t W_t'. Returns first failing (stage*n + pivot + 1) or 0. */
static int FN(bt_factor_one)(int T, int nx, int nu, const REAL *Hd, const REAL *Hs,
                             REAL *L, REAL *W)
{
    int n = nx + nu, info = 0;
    for (long i = 0; i < (long)T * n * n; ++i) L[i] = Hd[i];
    for (int t = 0; t < T; ++t) {
        REAL *Lt = L + (long)t * n * n;
        int fi = FN(chol_block)(n, Lt);
        if (fi && !info) info = t * n + fi;
        if (t == T - 1) break;
        REAL *Wt = W + (long)t * nx * n;
        const REAL *S = Hs + (long)t * n * n;
        for (int r = 0; r < nx; ++r)
            for (int j = 0; j < n; ++j) {
                REAL s = S[r * n + j];
                for (int k = 0; k < j; ++k) s -= Wt[r * n + k] * Lt[j * n + k];
                Wt[r * n + j] = s / Lt[j * n + j];
            }
        REAL *Ln = Lt + n * n;
        for (int a = 0; a < nx; ++a)
            for (int b = 0; b < nx; ++b) {
                REAL s = 0;
                for (int k = 0; k < n; ++k) s += Wt[a * n + k] * Wt[b * n + k];
                Ln[a * n + b] -= s;
            }
    }
    return info;
}

/* Solve H x = rhs with the factor; rhs/x [T][n]. */
static void FN(bt_solve_one)(int T, int nx, int nu, const REAL *L, const REAL *W,
                             const REAL *rhs, REAL *x)
{
    int n = nx + nu;
    for (int t = 0; t < T; ++t) {
        const REAL *Lt = L + (long)t * n * n;
        for (int j = 0; j < n; ++j) {
            REAL s = rhs[t * n + j];
            if (t > 0 && j < nx) {
                const REAL *Wp = W + (long)(t - 1) * nx * n;
                for (int k = 0; k < n; ++k) s -= Wp[j * n + k] * x[(t - 1) * n + k];
            }
            for (int k = 0; k < j; ++k) s -= Lt[j * n + k] * x[t * n + k];
            x[t * n + j] = s / Lt[j * n + j];
        }
    }
    for (int t = T - 1; t >= 0; --t) {
        const REAL *Lt = L + (long)t * n * n;
        for (int j = n - 1; j >= 0; --j) {
            REAL s = x[t * n + j];
            if (t < T - 1) {
                const REAL *Wt = W + (long)t * nx * n;
                for (int a = 0; a < nx; ++a) s -= Wt[a * n + j] * x[(t + 1) * n + a];
            }
            for (int i = j + 1; i < n; ++i) s -= Lt[i * n + j] * x[t * n + i];
            x[t * n + j] = s / Lt[j * n + j];
        }
    }
}

/* ---- dense path: what the reference literally does (al_utils.py:510-515) ---- */

static void FN(dense_from_band)(int T, int n, const REAL *Hd, const REAL *Hs, REAL *H)
{
    long N = (long)T * n;
    for (long i = 0; i < N * N; ++i) H[i] = 0;
    for (int t = 0; t < T; ++t)
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                H[((long)t * n + a) * N + t * n + b] = Hd[((long)t * n + a) * n + b];
                if (t < T - 1) {
                    REAL v = Hs[((long)t * n + a) * n + b];
                    H[((long)(t + 1) * n + a) * N + t * n + b] = v;
                    H[((long)t * n + b) * N + (t + 1) * n + a] = v;
                }
            }
}

/* Dense N x N Cholesky + solve of H x = rhs; H is destroyed. Returns info. */
static int FN(dense_chol_solve)(long N, REAL *H, const REAL *rhs, REAL *x)
{
    int info = 0;
    for (long j = 0; j < N; ++j) {
        REAL d = H[j * N + j];
        for (long k = 0; k < j; ++k) d -= H[j * N + k] * H[j * N + k];
        if (!(d > 0) && !info) info = (int)j + 1;
        REAL l = SQRT(d);
        H[j * N + j] = l;
        for (long i = j + 1; i < N; ++i) {
            REAL s = H[i * N + j];
            for (long k = 0; k < j; ++k) s -= H[i * N + k] * H[j * N + k];
            H[i * N + j] = s / l;
        }
    }
    for (long i = 0; i < N; ++i) {
        REAL s = rhs[i];
        for (long k = 0; k < i; ++k) s -= H[i * N + k] * x[k];
        x[i] = s / H[i * N + i];
    }
    for (long i = N - 1; i >= 0; --i) {
        REAL s = x[i];
        for (long k = i + 1; k < N; ++k) s -= H[k * N + i] * x[k];
        x[i] = s / H[i * N + i];
    }
    return info;
}

/* Dense LU with partial pivoting (torch.linalg.solve fallback, al_utils.py:518,528). */
static void FN(dense_lu_solve)(long N, REAL *A, const REAL *rhs, REAL *x)
{
    for (long i = 0; i < N; ++i) x[i] = rhs[i];
    for (long k = 0; k < N; ++k) {
        long p = k;
        REAL best = FABS(A[k * N + k]);
        for (long i = k + 1; i < N; ++i)
            if (FABS(A[i * N + k]) > best) { best = FABS(A[i * N + k]); p = i; }
        if (p != k) {
            for (long j = 0; j < N; ++j) { REAL tmp = A[k * N + j]; A[k * N + j] = A[p * N + j]; A[p * N + j] = tmp; }
            REAL tmp = x[k]; x[k] = x[p]; x[p] = tmp;
        }
        for (long i = k + 1; i < N; ++i) {
            REAL m = A[i * N + k] / A[k * N + k];
            A[i * N + k] = m;
            for (long j = k + 1; j < N; ++j) A[i * N + j] -= m * A[k * N + j];
            x[i] -= m * x[k];
        }
    }
    for (long i = N - 1; i >= 0; --i) {
        REAL s = x[i];
        for (long j = i + 1; j < N; ++j) s -= A[i * N + j] * x[j];
        x[i] = s / A[i * N + i];
    }
}

static int FN(has_nonfinite)(const REAL *v, long len)
{
    for (long i = 0; i < len; ++i)
        if (!(v[i] - v[i] == 0)) return 1;
    return 0;
}

/* ---- exported building blocks --------------------------------------------- */

#define IDX_U(b) ((long)(b) * sb_u)

/* g[B][T][n], Hd[B][T][n][n], Hs[B][T-1][n][n] at z, given xnext[B][T-1][nx]. */
void FN(orc_grad_hess)(int B, int T, int nx, int nu, const REAL *z, const REAL *xnext,
                       const REAL *F, const REAL *x0, const REAL *lam, const REAL *rho,
                       const REAL *Qd, const REAL *q, const REAL *ulo, const REAL *uhi,
                       long sb_u, long st_u, REAL *g, REAL *Hd, REAL *Hs)
{
    int n = nx + nu, M = T * nx + T * NINEQ_T;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        REAL *res = (REAL *)malloc(sizeof(REAL) * M);
        FN(residual_one)(T, nx, nu, z + (long)b * T * n, xnext + (long)b * (T - 1) * nx,
                         x0 + (long)b * nx, ulo + IDX_U(b), uhi + IDX_U(b), st_u, res, OBS_OF(b));
        FN(grad_hess_one)(T, nx, nu, z + (long)b * T * n, res, F + (long)b * (T - 1) * nx * n,
                          lam + (long)b * M, rho[b], Qd + (long)b * T * n, q + (long)b * T * n,
                          g + (long)b * T * n, Hd + (long)b * T * n * n,
                          Hs + (long)b * (T - 1) * n * n, OBS_OF(b));
        free(res);
    }
}

/* d = -H^{-1} g for every instance. solver: 0 banded Cholesky, 1 dense Cholesky,
 * 2 dense LU. Lout/Wout (nullable) receive the banded factor. info[B]. */
void FN(orc_newton_dir)(int B, int T, int nx, int nu, int solver, const REAL *g,
                        const REAL *Hd, const REAL *Hs, REAL *d, REAL *Lout, REAL *Wout,
                        int *info)
{
    int n = nx + nu;
    long N = (long)T * n;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        REAL *rhs = (REAL *)malloc(sizeof(REAL) * N);
        for (long i = 0; i < N; ++i) rhs[i] = -g[b * N + i];
        if (solver == 0) {
            REAL *L = (REAL *)malloc(sizeof(REAL) * T * n * n);
            REAL *W = (REAL *)malloc(sizeof(REAL) * (T > 1 ? T - 1 : 1) * nx * n);
            int fi = FN(bt_factor_one)(T, nx, nu, Hd + b * N * n, Hs + (long)b * (T - 1) * n * n, L, W);
            FN(bt_solve_one)(T, nx, nu, L, W, rhs, d + b * N);
            if (info) info[b] = fi;
            if (Lout) memcpy(Lout + b * N * n, L, sizeof(REAL) * T * n * n);
            if (Wout) memcpy(Wout + (long)b * (T - 1) * nx * n, W, sizeof(REAL) * (T - 1) * nx * n);
            free(L); free(W);
        } else {
            REAL *H = (REAL *)malloc(sizeof(REAL) * N * N);
            FN(dense_from_band)(T, n, Hd + b * N * n, Hs + (long)b * (T - 1) * n * n, H);
            if (solver == 1) {
                int fi = FN(dense_chol_solve)(N, H, rhs, d + b * N);
                if (info) info[b] = fi;
            } else {
                FN(dense_lu_solve)(N, H, rhs, d + b * N);
                if (info) info[b] = 0;
            }
            free(H);
        }
        free(rhs);
    }
}

/* phi[B] = merit(z) and rp2[B] = sum r+^2, given xnext. */
void FN(orc_merit)(int B, int T, int nx, int nu, const REAL *z, const REAL *xnext,
                   const REAL *x0, const REAL *lam, const REAL *rho, const REAL *Qd,
                   const REAL *q, const REAL *ulo, const REAL *uhi, long sb_u, long st_u,
                   REAL *phi, REAL *rp2)
{
    int n = nx + nu, M = T * nx + T * NINEQ_T;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        REAL *res = (REAL *)malloc(sizeof(REAL) * M);
        FN(residual_one)(T, nx, nu, z + (long)b * T * n, xnext + (long)b * (T - 1) * nx,
                         x0 + (long)b * nx, ulo + IDX_U(b), uhi + IDX_U(b), st_u, res, OBS_OF(b));
        REAL sq;
        phi[b] = FN(merit_one)(T, nx, nu, z + (long)b * T * n, res, lam + (long)b * M, rho[b],
                               Qd + (long)b * T * n, q + (long)b * T * n, &sq);
        if (rp2) rp2[b] = sq;
        free(res);
    }
}

/* Line-search decision, qpth/al_utils.py:634-641: first argmin over the 20
 * candidates (a NaN wins like torch.min), accept iff strictly below phi_prev.
 * phi_all[n_ls][B]. Writes k[B], accept[B], phi_min[B]. */
void FN(orc_linesearch_pick)(int B, int n_ls, const REAL *phi_all, const REAL *phi_prev,
                             int *kout, int *accept, REAL *phi_min)
{
    for (int b = 0; b < B; ++b) {
        int best = 0;
        REAL bv = phi_all[b];
        for (int k = 1; k < n_ls; ++k) {
            REAL v = phi_all[(long)k * B + b];
            if (bv != bv) break;
            if (v != v || v < bv) { bv = v; best = k; }
        }
        kout[b] = best;
        phi_min[b] = bv;
        accept[b] = (bv < phi_prev[b]) ? 1 : 0;
    }
}

/* lam <- lam + rho r (unclamped r, every row); lam[neq:] <- max(0, .); rho <- 10 rho.
 * qpth/AL_mpc.py:315-317, 325. */
void FN(orc_dual_update)(int B, int T, int nx, int nu, const REAL *z, const REAL *xnext,
                         const REAL *x0, const REAL *ulo, const REAL *uhi, long sb_u,
                         long st_u, REAL *lam, REAL *rho)
{
    int n = nx + nu, neq = T * nx, M = neq + T * NINEQ_T;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        REAL *res = (REAL *)malloc(sizeof(REAL) * M);
        FN(residual_one)(T, nx, nu, z + (long)b * T * n, xnext + (long)b * (T - 1) * nx,
                         x0 + (long)b * nx, ulo + IDX_U(b), uhi + IDX_U(b), st_u, res, OBS_OF(b));
        REAL *l = lam + (long)b * M;
        for (int r = 0; r < M; ++r) {
            REAL v = l[r] + rho[b] * res[r];
            l[r] = (r >= neq && v < 0) ? 0 : v;
        }
        rho[b] *= 10;
        free(res);
    }
}

/* ---- the whole LinDx solve -------------------------------------------------- */

/* One instance-parallel restatement of MPC.al_solve (AL_mpc.py:260-339) driving
 * NewtonAL.forward (al_utils.py:451-576) on affine dynamics.
 *   exit_mode 0: always max_newton steps (no batch-global decisions)
 *   exit_mode 1: the reference's batch-global early exit (al_utils.py:486,552,560-564)
 *   solver     : 0 banded, 1 dense Cholesky (2 = LU is entered automatically and
 *                stickily when any update is NaN/Inf, :517-531; only with solver 1)
 * Trace arrays are nullable; step-major [max_steps][B]...
 * Returns the total number of Newton steps executed. newton_per_al[al_iter]. */
typedef struct {
    REAL *g, *d, *phi, *phi_prev, *z;   /* [S][B][T][n], .., [S][20][B], [S][B], [S][B][T][n] */
    int *k, *accept;                    /* [S][B] */
    REAL *Hd, *Hs;                      /* first step of each AL iteration: [al][B][T][n][n] .. */
    int max_steps;
} FN(orc_trace);

int FN(orc_solve_lin)(int B, int T, int nx, int nu, int al_iter, int max_newton, int n_ls,
                      int exit_mode, int solver, const REAL *Qd, const REAL *q, const REAL *F,
                      const REAL *c, const REAL *x0, const REAL *ulo, const REAL *uhi,
                      long sb_u, long st_u, REAL *z, REAL *lam, REAL *rho,
                      unsigned char *status, int *newton_per_al, FN(orc_trace) *tr,
                      REAL *Lsave, REAL *zsave)
{
    int n = nx + nu, neq = T * nx, M = neq + T * NINEQ_T;
    long N = (long)T * n, XN = (long)(T - 1) * nx;
    REAL *xnext = (REAL *)malloc(sizeof(REAL) * B * (XN > 0 ? XN : 1));
    REAL *g = (REAL *)malloc(sizeof(REAL) * B * N);
    REAL *d = (REAL *)malloc(sizeof(REAL) * B * N);
    REAL *Hd = (REAL *)malloc(sizeof(REAL) * B * N * n);
    long HSN = (long)(T > 1 ? T - 1 : 1) * n * n;
    REAL *Hs = (REAL *)malloc(sizeof(REAL) * B * HSN);
    REAL *phi = (REAL *)malloc(sizeof(REAL) * B);
    REAL *rp2 = (REAL *)malloc(sizeof(REAL) * B);
    REAL *phi_all = (REAL *)malloc(sizeof(REAL) * n_ls * B);
    REAL *phi_min = (REAL *)malloc(sizeof(REAL) * B);
    REAL *zc = (REAL *)malloc(sizeof(REAL) * B * N);
    int *kk = (int *)malloc(sizeof(int) * B), *acc = (int *)malloc(sizeof(int) * B);
    int *info = (int *)malloc(sizeof(int) * B);
    REAL *Lf = (REAL *)malloc(sizeof(REAL) * B * N * n);
    int total = 0, chol_fail = 0;
    if (status) for (int b = 0; b < B; ++b) status[b] = 1;

#define XNEXT_ALL(zz)                                                                  \
    _Pragma("omp parallel for schedule(static)")                                       \
    for (int b = 0; b < B; ++b)                                                        \
        FN(affine_next_one)(T, nx, nu, (zz) + b * N, F + (long)b * XN * n, c + (long)b * XN, \
                            xnext + (long)b * XN);

    for (int it = 0; it < al_iter; ++it) {
        /* NewtonAL.forward */
        XNEXT_ALL(z)
        FN(orc_merit)(B, T, nx, nu, z, xnext, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, phi, rp2);
        double old = 0;
        for (int b = 0; b < B; ++b) old += rp2[b];
        old = sqrt(old);
        int nstep = 0;
        chol_fail = 0; /* cholesky_fail is local to each NewtonAL.forward call (:490) */
        while (nstep < max_newton) {
            nstep++;
            FN(orc_grad_hess)(B, T, nx, nu, z, xnext, F, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, g, Hd, Hs);
            if (zsave) memcpy(zsave, z, sizeof(REAL) * B * N);
            if (!chol_fail) {
                FN(orc_newton_dir)(B, T, nx, nu, solver, g, Hd, Hs, d, solver == 0 ? Lf : 0, 0, info);
                if (solver == 1 && FN(has_nonfinite)(d, B * N)) {
                    FN(orc_newton_dir)(B, T, nx, nu, 2, g, Hd, Hs, d, 0, 0, info);
                    chol_fail = 1;
                }
            } else {
                FN(orc_newton_dir)(B, T, nx, nu, 2, g, Hd, Hs, d, 0, 0, info);
            }
            if (Lsave && solver == 0) memcpy(Lsave, Lf, sizeof(REAL) * B * N * n);
            /* line search: 20 candidates z + 2^-k d, one merit call each (:623-633) */
            for (int k = 0; k < n_ls; ++k) {
                REAL alpha = (REAL)ldexp(1.0, -k);
                for (long i = 0; i < B * N; ++i) zc[i] = z[i] + alpha * d[i];
                XNEXT_ALL(zc)
                FN(orc_merit)(B, T, nx, nu, zc, xnext, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u,
                              phi_all + (long)k * B, 0);
            }
            FN(orc_linesearch_pick)(B, n_ls, phi_all, phi, kk, acc, phi_min);
            if (tr && total < tr->max_steps) {
                long s = total;
                if (tr->g) memcpy(tr->g + s * B * N, g, sizeof(REAL) * B * N);
                if (tr->d) memcpy(tr->d + s * B * N, d, sizeof(REAL) * B * N);
                if (tr->phi) memcpy(tr->phi + s * n_ls * B, phi_all, sizeof(REAL) * n_ls * B);
                if (tr->phi_prev) memcpy(tr->phi_prev + s * B, phi, sizeof(REAL) * B);
                if (tr->k) memcpy(tr->k + s * B, kk, sizeof(int) * B);
                if (tr->accept) memcpy(tr->accept + s * B, acc, sizeof(int) * B);
                if (nstep == 1 && tr->Hd) {
                    memcpy(tr->Hd + (long)it * B * N * n, Hd, sizeof(REAL) * B * N * n);
                    memcpy(tr->Hs + (long)it * B * (long)(T - 1) * n * n, Hs, sizeof(REAL) * B * (long)(T - 1) * n * n);
                }
            }
            for (int b = 0; b < B; ++b) {
                if (acc[b]) {
                    REAL alpha = (REAL)ldexp(1.0, -kk[b]);
                    for (long i = 0; i < N; ++i) z[b * N + i] += alpha * d[b * N + i];
                }
                if (status && FN(has_nonfinite)(z + b * N, N)) status[b] = 0;
            }
            if (tr && total < tr->max_steps && tr->z)
                memcpy(tr->z + (long)total * B * N, z, sizeof(REAL) * B * N);
            total++;
            XNEXT_ALL(z)
            FN(orc_merit)(B, T, nx, nu, z, xnext, x0, lam, rho, Qd, q, ulo, uhi, sb_u, st_u, phi_all, rp2);
            double nw = 0;
            for (int b = 0; b < B; ++b) nw += rp2[b];
            nw = sqrt(nw);
            if (exit_mode == 1 && (fabs(old - nw) / nw < 1e-3 || nw < 1e-3)) break;
            old = nw;
            for (int b = 0; b < B; ++b) phi[b] = phi_min[b]; /* merit <- new_merit (:569) */
        }
        if (newton_per_al) newton_per_al[it] = nstep;
        /* dual update with the true dynamics (AL_mpc.py:315-325); xnext is at z */
        FN(orc_dual_update)(B, T, nx, nu, z, xnext, x0, ulo, uhi, sb_u, st_u, lam, rho);
    }
    (void)M;
    free(xnext); free(g); free(d); free(Hd); free(Hs); free(phi); free(rp2); free(phi_all);
    free(phi_min); free(zc); free(kk); free(acc); free(info); free(Lf);
    return total;
#undef XNEXT_ALL
}

/* NewtonAL.backward (al_utils.py:578-615): w = -H^{-1} gbar with the saved banded
 * factor; q_grad = w, Qd_grad = w * z_saved. L[B][T][n][n]; W is rebuilt from F, rho. */
void FN(orc_backward)(int B, int T, int nx, int nu, const REAL *L, const REAL *F,
                      const REAL *rho, const REAL *zs, const REAL *gbar, REAL *q_grad,
                      REAL *Qd_grad)
{
    int n = nx + nu;
    long N = (long)T * n;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        REAL *W = (REAL *)malloc(sizeof(REAL) * (T > 1 ? T - 1 : 1) * nx * n);
        REAL *rhs = (REAL *)malloc(sizeof(REAL) * N);
        for (int t = 0; t < T - 1; ++t) {
            const REAL *Lt = L + (b * N + (long)t * n) * n;
            const REAL *Ft = F + ((long)b * (T - 1) + t) * nx * n;
            REAL *Wt = W + (long)t * nx * n;
            for (int r = 0; r < nx; ++r)
                for (int j = 0; j < n; ++j) {
                    REAL s = -rho[b] * Ft[r * n + j];
                    for (int k = 0; k < j; ++k) s -= Wt[r * n + k] * Lt[j * n + k];
                    Wt[r * n + j] = s / Lt[j * n + j];
                }
        }
        for (long i = 0; i < N; ++i) rhs[i] = -gbar[b * N + i];
        FN(bt_solve_one)(T, nx, nu, L + b * N * n, W, rhs, q_grad + b * N);
        for (long i = 0; i < N; ++i) Qd_grad[b * N + i] = q_grad[b * N + i] * zs[b * N + i];
        free(W); free(rhs);
    }
}

#undef IDX_U
#undef FN
#undef CAT
#undef CAT_
