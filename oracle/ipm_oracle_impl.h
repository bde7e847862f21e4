/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 *
 * CPU restatement (plain C) of the reference's INTERIOR-POINT QP solve for the MPC problem
 * (SURVEY.md 8f-1): `qpth.qp.DenseQPFunction` (qp.py:187-270) = `pdipm_b_LU.forward`
 * (qpth/solvers/pdipm/batch_LU.py:29-197), `get_step` (:200-208), `solve_kkt` (:212-244) on the dense
 * QP that `qp_wrapper.MPC.single_qp` assembles (qp_wrapper.py:295-321, 612-653):
 *
 *     min 1/2 z'Qz + p'z   s.t.  G z <= h,  A z = b,      z = (tau_0 .. tau_{T-1}),  tau_t = (x_t, u_t)
 *     Q  block-diagonal cost (DIAGONAL here: what policies.Tracking_MPC builds, policies.py:1172)
 *     G  = [+I on the controls ; -I on the controls], h = [u_upper tiled ; -u_lower tiled]   (:638-653)
 *     A  rows t*nx..: F_t tau_t - x_{t+1} = -f_t (t < T-1); last nx rows: x_0 = x0          (:612-629)
 *
 * Two KKT solvers behind the same iteration:
 *   solver 1: the literal one - dense K of order nz + 2 nineq + neq, LU with partial pivoting of the
 *             regularised Ktilde, one step of iterative refinement against K (batch_LU.py:212-244);
 *   solver 0: the structured elimination the HIP kernel implements (ds, dz eliminated, a diagonal
 *             Phi = Q + eps + G'DG, block-tridiagonal Schur complement A Phi^-1 A' + eps on the equality
 *             multipliers) - the SAME regularised system and the same refinement step, so in exact
 *             arithmetic both give the same iterates.
 * Batch-global decisions are restated as they are: the exit rule (no instance improved
 * `notImprovedLim` times in a row, or max best residual < eps, or min mu > 1e32, :147-151) and
 * get_step's `a.max()` over the WHOLE batch (:207). exit_mode 1 ("fixed") replaces both with their
 * per-instance forms (all maxIter iterations, best iterate kept per instance; no cap from other
 * instances) - what the one-launch kernel does.
 *
 * Included twice by ipm_oracle.c (REAL = double / float). Parity pin: tests/test_ip_golden.py against
 * tests/golden/ip_*.npz, produced by tools/gen_golden_ip.py from the reference itself.
 *
 * Layouts (batch-major, per instance contiguous): Qd,p [B][T][n]; F [B][T-1][nx][n]; f [B][T-1][nx];
 * x0 [B][nx]; uhi, ulo [nu]; x [B][T*n]; s,z [B][2*T*nu] (upper rows t*nu+j, then lower rows
 * T*nu + t*nu+j); y [B][T*nx] (dynamics rows t*nx+i, then the initial-state rows (T-1)*nx+i).
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

typedef void (*FN(ipm_ry_cb))(const REAL *x /*[B][T*n]*/, REAL *ry /*[B][T*nx]*/, void *ctx);

typedef struct {
    int T, nx, nu, n, nz, ni, ne;
    const REAL *Qd, *p, *F, *f, *x0, *uhi, *ulo;
} FN(ipm_prob);

/* ---- products with the structured K (no regularisation) -------------------------------------- */
/* out = K(zd, sd) * (dx, ds, dz, dy); any out_* may alias nothing. */
static void FN(ipm_Kmul)(const FN(ipm_prob) *P, const REAL *zd, const REAL *sd, const REAL *dx,
                         const REAL *ds, const REAL *dz, const REAL *dy, REAL *ox, REAL *os, REAL *oz,
                         REAL *oy)
{
    int T = P->T, nx = P->nx, nu = P->nu, n = P->n, Tn = T * nu;
    for (int i = 0; i < P->nz; ++i) ox[i] = P->Qd[i] * dx[i];
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < nu; ++j) ox[t * n + nx + j] += dz[t * nu + j] - dz[Tn + t * nu + j];
    for (int t = 0; t < T - 1; ++t) {
        const REAL *Ft = P->F + (long)t * nx * n;
        for (int r = 0; r < nx; ++r) {
            REAL yr = dy[t * nx + r];
            for (int k = 0; k < n; ++k) ox[t * n + k] += Ft[r * n + k] * yr;
            ox[(t + 1) * n + r] -= yr;
        }
    }
    for (int r = 0; r < nx; ++r) ox[r] += dy[(T - 1) * nx + r];
    for (int i = 0; i < P->ni; ++i) os[i] = zd[i] * ds[i] + sd[i] * dz[i];
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < nu; ++j) {
            oz[t * nu + j] = dx[t * n + nx + j] + ds[t * nu + j];
            oz[Tn + t * nu + j] = -dx[t * n + nx + j] + ds[Tn + t * nu + j];
        }
    for (int t = 0; t < T - 1; ++t) {
        const REAL *Ft = P->F + (long)t * nx * n;
        for (int r = 0; r < nx; ++r) {
            REAL a = 0;
            for (int k = 0; k < n; ++k) a += Ft[r * n + k] * dx[t * n + k];
            oy[t * nx + r] = a - dx[(t + 1) * n + r];
        }
    }
    for (int r = 0; r < nx; ++r) oy[(T - 1) * nx + r] = dx[r];
}

/* ---- solver 0: structured solve of the regularised system ------------------------------------- */
/* (Qd+e1) dx + G'dz + A'dy = bx ; (zd+e2) ds + sd dz = bs ; G dx + ds - e3 dz = bz ; A dx - e4 dy = by.
 * Workspace w: at least nz + 3 ni + 2 ne + T*(2 nx*nx) reals. Returns 0, or >0 on a non-positive pivot. */
static int FN(ipm_solve_struct)(const FN(ipm_prob) *P, const REAL *zd, const REAL *sd, REAL e1, REAL e2,
                                REAL e3, REAL e4, const REAL *bx, const REAL *bs, const REAL *bz,
                                const REAL *by, REAL *dx, REAL *ds, REAL *dz, REAL *dy, REAL *w)
{
    int T = P->T, nx = P->nx, nu = P->nu, n = P->n, Tn = T * nu, info = 0;
    REAL *Pinv = w;                 /* [nz]   1/Phi */
    REAL *Dt = Pinv + P->nz;        /* [ni]   D~ = 1/(sd/(zd+e2) + e3) */
    REAL *wv = Dt + P->ni;          /* [ni]   bs/(zd+e2) - bz */
    REAL *r1 = wv + P->ni;          /* [nz]   rhs of the x rows */
    REAL *ry = r1 + P->nz;          /* [ne]   rhs of S, internal order: block 0 = init, block t+1 = dyn t */
    REAL *Ld = ry + P->ne;          /* [T][nx][nx] diagonal Cholesky blocks of S */
    REAL *Lo = Ld + (long)T * nx * nx; /* [T][nx][nx] sub-diagonal blocks W_m (m >= 1) */
    for (int i = 0; i < P->ni; ++i) {
        REAL zt = zd[i] + e2;
        Dt[i] = (REAL)1 / (sd[i] / zt + e3);
        wv[i] = bs[i] / zt - bz[i];
    }
    for (int i = 0; i < P->nz; ++i) { Pinv[i] = P->Qd[i] + e1; r1[i] = bx[i]; }
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < nu; ++j) {
            int iu = t * nu + j, il = Tn + iu, k = t * n + nx + j;
            Pinv[k] += Dt[iu] + Dt[il];
            r1[k] -= Dt[iu] * wv[iu] - Dt[il] * wv[il];
        }
    for (int i = 0; i < P->nz; ++i) Pinv[i] = (REAL)1 / Pinv[i];
    /* rhs of S: A Phi^-1 r1 - by */
    for (int r = 0; r < nx; ++r) ry[r] = Pinv[r] * r1[r] - by[(T - 1) * nx + r];
    for (int t = 0; t < T - 1; ++t) {
        const REAL *Ft = P->F + (long)t * nx * n;
        for (int r = 0; r < nx; ++r) {
            REAL a = 0;
            for (int k = 0; k < n; ++k) a += Ft[r * n + k] * Pinv[t * n + k] * r1[t * n + k];
            ry[(t + 1) * nx + r] = a - Pinv[(t + 1) * n + r] * r1[(t + 1) * n + r] - by[t * nx + r];
        }
    }
    /* S blocks + block-tridiagonal Cholesky */
    for (int m = 0; m < T; ++m) {
        REAL *Sm = Ld + (long)m * nx * nx, *Wm = Lo + (long)m * nx * nx;
        for (int i = 0; i < nx * nx; ++i) Sm[i] = 0;
        if (m == 0) {
            for (int r = 0; r < nx; ++r) Sm[r * nx + r] = Pinv[r] + e4;
        } else {
            int t = m - 1;
            const REAL *Ft = P->F + (long)t * nx * n;
            for (int a = 0; a < nx; ++a)
                for (int b2 = 0; b2 <= a; ++b2) {
                    REAL acc = 0;
                    for (int k = 0; k < n; ++k) acc += Ft[a * n + k] * Pinv[t * n + k] * Ft[b2 * n + k];
                    Sm[a * nx + b2] = acc;
                }
            for (int r = 0; r < nx; ++r) Sm[r * nx + r] += Pinv[(t + 1) * n + r] + e4;
            /* S_{m,m-1}: m = 1: F_0 P_0 E' ; m >= 2: -F_t[:, :nx] diag(P_t x)   (rows dyn t, cols dyn t-1) */
            REAL sign = (m == 1) ? (REAL)1 : (REAL)-1;
            for (int a = 0; a < nx; ++a)
                for (int c = 0; c < nx; ++c) Wm[a * nx + c] = sign * Ft[a * n + c] * Pinv[t * n + c];
            /* W_m = S_{m,m-1} L_{m-1}^{-T} */
            const REAL *Lp = Ld + (long)(m - 1) * nx * nx;
            for (int a = 0; a < nx; ++a)
                for (int c = 0; c < nx; ++c) {
                    REAL v = Wm[a * nx + c];
                    for (int k = 0; k < c; ++k) v -= Wm[a * nx + k] * Lp[c * nx + k];
                    Wm[a * nx + c] = v / Lp[c * nx + c];
                }
            for (int a = 0; a < nx; ++a)
                for (int b2 = 0; b2 <= a; ++b2) {
                    REAL acc = 0;
                    for (int k = 0; k < nx; ++k) acc += Wm[a * nx + k] * Wm[b2 * nx + k];
                    Sm[a * nx + b2] -= acc;
                }
        }
        for (int j = 0; j < nx; ++j) {   /* in-place lower Cholesky */
            REAL d = Sm[j * nx + j];
            for (int k = 0; k < j; ++k) d -= Sm[j * nx + k] * Sm[j * nx + k];
            if (!(d > 0) && !info) info = m * nx + j + 1;
            REAL l = SQRT(d < 0 ? -d : d);
            Sm[j * nx + j] = l;
            for (int i = j + 1; i < nx; ++i) {
                REAL v = Sm[i * nx + j];
                for (int k = 0; k < j; ++k) v -= Sm[i * nx + k] * Sm[j * nx + k];
                Sm[i * nx + j] = v / l;
            }
        }
    }
    /* forward / backward substitution (ry in place -> dy in internal order) */
    for (int m = 0; m < T; ++m) {
        const REAL *Lm = Ld + (long)m * nx * nx, *Wm = Lo + (long)m * nx * nx;
        REAL *v = ry + m * nx;
        if (m > 0)
            for (int a = 0; a < nx; ++a) {
                REAL acc = 0;
                for (int k = 0; k < nx; ++k) acc += Wm[a * nx + k] * ry[(m - 1) * nx + k];
                v[a] -= acc;
            }
        for (int a = 0; a < nx; ++a) {
            REAL acc = v[a];
            for (int k = 0; k < a; ++k) acc -= Lm[a * nx + k] * v[k];
            v[a] = acc / Lm[a * nx + a];
        }
    }
    for (int m = T - 1; m >= 0; --m) {
        const REAL *Lm = Ld + (long)m * nx * nx;
        REAL *v = ry + m * nx;
        if (m < T - 1) {
            const REAL *Wn = Lo + (long)(m + 1) * nx * nx;
            for (int k = 0; k < nx; ++k) {
                REAL acc = 0;
                for (int a = 0; a < nx; ++a) acc += Wn[a * nx + k] * ry[(m + 1) * nx + a];
                v[k] -= acc;
            }
        }
        for (int a = nx - 1; a >= 0; --a) {
            REAL acc = v[a];
            for (int k = a + 1; k < nx; ++k) acc -= Lm[k * nx + a] * v[k];
            v[a] = acc / Lm[a * nx + a];
        }
    }
    for (int r = 0; r < nx; ++r) dy[(T - 1) * nx + r] = ry[r];
    for (int t = 0; t < T - 1; ++t)
        for (int r = 0; r < nx; ++r) dy[t * nx + r] = ry[(t + 1) * nx + r];
    /* dx = Phi^-1 (r1 - A'dy) */
    for (int i = 0; i < P->nz; ++i) dx[i] = r1[i];
    for (int t = 0; t < T - 1; ++t) {
        const REAL *Ft = P->F + (long)t * nx * n;
        for (int r = 0; r < nx; ++r) {
            REAL yr = dy[t * nx + r];
            for (int k = 0; k < n; ++k) dx[t * n + k] -= Ft[r * n + k] * yr;
            dx[(t + 1) * n + r] += yr;
        }
    }
    for (int r = 0; r < nx; ++r) dx[r] -= dy[(T - 1) * nx + r];
    for (int i = 0; i < P->nz; ++i) dx[i] *= Pinv[i];
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < nu; ++j) {
            int iu = t * nu + j, il = Tn + iu;
            REAL du = dx[t * n + nx + j];
            dz[iu] = Dt[iu] * (du + wv[iu]);
            dz[il] = Dt[il] * (-du + wv[il]);
        }
    for (int i = 0; i < P->ni; ++i) ds[i] = (bs[i] - sd[i] * dz[i]) / (zd[i] + e2);
    return info;
}

/* ---- solver 1: dense K / Ktilde, LU with partial pivoting (the literal restatement) ----------- */
static void FN(ipm_dense_K)(const FN(ipm_prob) *P, const REAL *zd, const REAL *sd, REAL e1, REAL e2, REAL e3,
                            REAL e4, REAL *K /*[NK][NK]*/)
{
    int T = P->T, nx = P->nx, nu = P->nu, n = P->n, Tn = T * nu, nz = P->nz, ni = P->ni, ne = P->ne;
    long NK = nz + 2 * ni + ne;
    for (long i = 0; i < NK * NK; ++i) K[i] = 0;
    int os = nz, oz = nz + ni, oy = nz + 2 * ni;
#define KK(r, c) K[(long)(r) * NK + (c)]
    for (int i = 0; i < nz; ++i) KK(i, i) = P->Qd[i] + e1;
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < nu; ++j) {
            int iu = t * nu + j, il = Tn + iu, k = t * n + nx + j;
            KK(k, oz + iu) = 1; KK(k, oz + il) = -1;
            KK(oz + iu, k) = 1; KK(oz + il, k) = -1;
        }
    for (int i = 0; i < ni; ++i) {
        KK(os + i, os + i) = zd[i] + e2;
        KK(os + i, oz + i) = sd[i];
        KK(oz + i, os + i) = 1;
        KK(oz + i, oz + i) = -e3;
    }
    for (int t = 0; t < T - 1; ++t) {
        const REAL *Ft = P->F + (long)t * nx * n;
        for (int r = 0; r < nx; ++r) {
            for (int k = 0; k < n; ++k) { KK(oy + t * nx + r, t * n + k) = Ft[r * n + k]; KK(t * n + k, oy + t * nx + r) = Ft[r * n + k]; }
            KK(oy + t * nx + r, (t + 1) * n + r) = -1; KK((t + 1) * n + r, oy + t * nx + r) = -1;
        }
    }
    for (int r = 0; r < nx; ++r) { KK(oy + (T - 1) * nx + r, r) = 1; KK(r, oy + (T - 1) * nx + r) = 1; }
    for (int i = 0; i < ne; ++i) KK(oy + i, oy + i) = -e4;
#undef KK
}

static void FN(ipm_lu_factor)(long N, REAL *A, int *piv)
{
    for (long k = 0; k < N; ++k) {
        long p = k;
        REAL best = FABS(A[k * N + k]);
        for (long i = k + 1; i < N; ++i)
            if (FABS(A[i * N + k]) > best) { best = FABS(A[i * N + k]); p = i; }
        piv[k] = (int)p;
        if (p != k)
            for (long j = 0; j < N; ++j) { REAL tmp = A[k * N + j]; A[k * N + j] = A[p * N + j]; A[p * N + j] = tmp; }
        REAL d = A[k * N + k];
        for (long i = k + 1; i < N; ++i) {
            REAL l = A[i * N + k] / d;
            A[i * N + k] = l;
            if (l != 0)
                for (long j = k + 1; j < N; ++j) A[i * N + j] -= l * A[k * N + j];
        }
    }
}

static void FN(ipm_lu_solve)(long N, const REAL *A, const int *piv, REAL *b)
{
    for (long k = 0; k < N; ++k) {   /* rows were swapped whole (multipliers included): permute first */
        long p = piv[k];
        if (p != k) { REAL tmp = b[k]; b[k] = b[p]; b[p] = tmp; }
    }
    for (long k = 0; k < N; ++k)
        for (long i = k + 1; i < N; ++i) b[i] -= A[i * N + k] * b[k];
    for (long k = N - 1; k >= 0; --k) {
        REAL v = b[k];
        for (long j = k + 1; j < N; ++j) v -= A[k * N + j] * b[j];
        b[k] = v / A[k * N + k];
    }
}

/* ---- solve_kkt (batch_LU.py:212-244): l = Ktilde^-1 r, one refinement step against K ---------- */
/* ctx of one instance: a factor is built once per (zd, sd) and used for several right-hand sides. */
typedef struct {
    const FN(ipm_prob) *P;
    int solver;
    REAL e1, e2, e3, e4;     /* regularisation of Ktilde (KKTeps, or 0 for the backward pass) */
    const REAL *zd, *sd;     /* diagonal entries of K at (s,s) and (s,z) */
    REAL *lu; int *piv;      /* solver 1 */
    REAL *w;                 /* solver 0 workspace */
    REAL *tmp;               /* [2 NK] */
    int info;
} FN(ipm_kkt);

static void FN(ipm_kkt_factor)(FN(ipm_kkt) *k)
{
    if (k->solver == 1) {
        long NK = k->P->nz + 2 * k->P->ni + k->P->ne;
        FN(ipm_dense_K)(k->P, k->zd, k->sd, k->e1, k->e2, k->e3, k->e4, k->lu);
        FN(ipm_lu_factor)(NK, k->lu, k->piv);
    }
}

static void FN(ipm_kkt_apply)(FN(ipm_kkt) *k, const REAL *bx, const REAL *bs, const REAL *bz, const REAL *by,
                              REAL *dx, REAL *ds, REAL *dz, REAL *dy)
{
    const FN(ipm_prob) *P = k->P;
    int nz = P->nz, ni = P->ni, ne = P->ne;
    if (k->solver == 1) {
        long NK = nz + 2 * ni + ne;
        REAL *v = k->tmp;
        memcpy(v, bx, sizeof(REAL) * nz); memcpy(v + nz, bs, sizeof(REAL) * ni);
        memcpy(v + nz + ni, bz, sizeof(REAL) * ni); memcpy(v + nz + 2 * ni, by, sizeof(REAL) * ne);
        FN(ipm_lu_solve)(NK, k->lu, k->piv, v);
        memcpy(dx, v, sizeof(REAL) * nz); memcpy(ds, v + nz, sizeof(REAL) * ni);
        memcpy(dz, v + nz + ni, sizeof(REAL) * ni); memcpy(dy, v + nz + 2 * ni, sizeof(REAL) * ne);
    } else {
        int fi = FN(ipm_solve_struct)(P, k->zd, k->sd, k->e1, k->e2, k->e3, k->e4, bx, bs, bz, by, dx, ds, dz, dy, k->w);
        if (fi && !k->info) k->info = fi;
    }
}

/* rx, rs, rz, ry are the residuals; the solution solves K l = -(rx, rs, rz, ry). */
static void FN(ipm_solve_kkt)(FN(ipm_kkt) *k, const REAL *rx, const REAL *rs, const REAL *rz, const REAL *ry,
                              REAL *dx, REAL *ds, REAL *dz, REAL *dy, REAL *scratch /*[3 NK]*/)
{
    const FN(ipm_prob) *P = k->P;
    int nz = P->nz, ni = P->ni, ne = P->ne;
    long NK = nz + 2 * ni + ne;
    REAL *r = scratch, *res = scratch + NK, *d = scratch + 2 * NK;
    for (int i = 0; i < nz; ++i) r[i] = -rx[i];
    for (int i = 0; i < ni; ++i) { r[nz + i] = -rs[i]; r[nz + ni + i] = -rz[i]; }
    for (int i = 0; i < ne; ++i) r[nz + 2 * ni + i] = -ry[i];
    FN(ipm_kkt_apply)(k, r, r + nz, r + nz + ni, r + nz + 2 * ni, dx, ds, dz, dy);
    /* res = r - K l ; l += Ktilde^-1 res   (niter = 1) */
    FN(ipm_Kmul)(P, k->zd, k->sd, dx, ds, dz, dy, res, res + nz, res + nz + ni, res + nz + 2 * ni);
    for (long i = 0; i < NK; ++i) res[i] = r[i] - res[i];
    FN(ipm_kkt_apply)(k, res, res + nz, res + nz + ni, res + nz + 2 * ni, d, d + nz, d + nz + ni, d + nz + 2 * ni);
    for (int i = 0; i < nz; ++i) dx[i] += d[i];
    for (int i = 0; i < ni; ++i) { ds[i] += d[nz + i]; dz[i] += d[nz + ni + i]; }
    for (int i = 0; i < ne; ++i) dy[i] += d[nz + 2 * ni + i];
}

/* get_step (batch_LU.py:200-208) for the whole batch: a = -v/dv; a[dv == 0] = 1; a[dv > 0] =
 * max(1, a.max() over the WHOLE batch); per-instance min. per_instance != 0: no cap from others. */
static void FN(ipm_get_step)(int B, int ni, const REAL *v, const REAL *dv, REAL *out, int per_instance)
{
    REAL amax = -INFINITY;
    for (long i = 0; i < (long)B * ni; ++i) {
        REAL a = (dv[i] == 0) ? (REAL)1 : -v[i] / dv[i];
        if (a > amax) amax = a;
    }
    REAL big = (amax > (REAL)1) ? amax : (REAL)1;   /* python max(1.0, a.max()): NaN -> 1.0 */
    if (per_instance) big = INFINITY;
    for (int b = 0; b < B; ++b) {
        REAL m = INFINITY;
        int nan_seen = 0;
        for (int i = 0; i < ni; ++i) {
            REAL d = dv[(long)b * ni + i], a;
            if (d == 0) a = 1;
            else if (d > 0) a = big;
            else a = -v[(long)b * ni + i] / d;
            if (a != a) nan_seen = 1;
            if (a < m) m = a;
        }
        out[b] = nan_seen ? (REAL)NAN : m;   /* torch.min propagates NaN */
    }
}

/*
 * pdipm_b_LU.forward for a batch. Returns the number of iterations executed (the index i at which the
 * exit rule fired, or maxIter). ry_cb == NULL: equality residual A x - b (LinDx data); otherwise the
 * caller's residual of the TRUE dynamics (qp_wrapper.py:306, 323-342; batch_LU.py:95).
 * Outputs (best iterate per instance): xb [B][nz], yb [B][ne], zb, sb [B][ni], resid [B], and the
 * initial point (x, s, z, y after solve_kkt and BEFORE the positivity shift) in init_* (nullable).
 */
int FN(orc_ipm_forward)(int B, int T, int nx, int nu, const REAL *Qd, const REAL *p, const REAL *F, const REAL *f,
                        const REAL *x0, const REAL *uhi, const REAL *ulo, int solver, int exit_mode,
                        double eps, int notImprovedLim, int maxIter, FN(ipm_ry_cb) ry_cb, void *cb_ctx,
                        REAL *xb, REAL *yb, REAL *zb, REAL *sb, REAL *resid_b, int *iter_best,
                        REAL *init_x, REAL *init_s, REAL *init_z, REAL *init_y, int *info)
{
    const REAL KKTeps = (REAL)1e-7;
    int n = nx + nu, nz = T * n, ni = 2 * T * nu, ne = T * nx;
    long NK = nz + 2 * ni + ne;
    REAL *x = calloc((size_t)B * nz, sizeof(REAL)), *s = calloc((size_t)B * ni, sizeof(REAL));
    REAL *z = calloc((size_t)B * ni, sizeof(REAL)), *y = calloc((size_t)B * ne, sizeof(REAL));
    REAL *rx = calloc((size_t)B * nz, sizeof(REAL)), *rs = calloc((size_t)B * ni, sizeof(REAL));
    REAL *rz = calloc((size_t)B * ni, sizeof(REAL)), *ry = calloc((size_t)B * ne, sizeof(REAL));
    REAL *dxa = calloc((size_t)B * nz, sizeof(REAL)), *dsa = calloc((size_t)B * ni, sizeof(REAL));
    REAL *dza = calloc((size_t)B * ni, sizeof(REAL)), *dya = calloc((size_t)B * ne, sizeof(REAL));
    REAL *dxc = calloc((size_t)B * nz, sizeof(REAL)), *dsc = calloc((size_t)B * ni, sizeof(REAL));
    REAL *dzc = calloc((size_t)B * ni, sizeof(REAL)), *dyc = calloc((size_t)B * ne, sizeof(REAL));
    REAL *h = calloc((size_t)ni, sizeof(REAL)), *mu = calloc(B, sizeof(REAL)), *resids = calloc(B, sizeof(REAL));
    REAL *st1 = calloc(B, sizeof(REAL)), *st2 = calloc(B, sizeof(REAL)), *alpha = calloc(B, sizeof(REAL));
    REAL *ones = calloc((size_t)ni, sizeof(REAL)), *zpe = calloc((size_t)B * ni, sizeof(REAL));
    REAL *zero = calloc((size_t)NK, sizeof(REAL));
    long wsz = nz + 3L * ni + 2L * ne + 2L * T * nx * nx + 16;
    REAL *w = calloc((size_t)B * wsz, sizeof(REAL)), *tmp = calloc((size_t)B * 2 * NK, sizeof(REAL));
    REAL *scr = calloc((size_t)B * 3 * NK, sizeof(REAL));
    REAL *lu = solver == 1 ? calloc((size_t)B * NK * NK, sizeof(REAL)) : NULL;
    int *piv = solver == 1 ? calloc((size_t)B * NK, sizeof(int)) : NULL;
    FN(ipm_prob) *P = calloc(B, sizeof(*P));
    FN(ipm_kkt) *kk = calloc(B, sizeof(*kk));
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < nu; ++j) { h[t * nu + j] = uhi[j]; h[T * nu + t * nu + j] = -ulo[j]; }
    for (int i = 0; i < ni; ++i) ones[i] = 1;
    for (int b = 0; b < B; ++b) {
        P[b] = (FN(ipm_prob)){T, nx, nu, n, nz, ni, ne, Qd + (long)b * nz, p + (long)b * nz,
                              F + (long)b * (T - 1) * nx * n, f + (long)b * (T - 1) * nx, x0 + (long)b * nx, uhi, ulo};
        kk[b] = (FN(ipm_kkt)){&P[b], solver, KKTeps, KKTeps, KKTeps, KKTeps, ones, ones,
                              lu ? lu + (long)b * NK * NK : NULL, piv ? piv + (long)b * NK : NULL,
                              w + (long)b * wsz, tmp + (long)b * 2 * NK, 0};
        if (info) info[b] = 0;
    }
    /* ---- initial point: solve_kkt(K, Ktilde, p, 0, -h, -b) with Sv = Zv = 1 (:44-69) */
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        REAL *nb = scr + (long)b * 3 * NK;
        REAL *mh = calloc(ni + ne, sizeof(REAL));    /* -h, -b passed as the "residuals" rz, ry */
        for (int i = 0; i < ni; ++i) mh[i] = -h[i];
        for (int t = 0; t < T - 1; ++t)
            for (int r = 0; r < nx; ++r) mh[ni + t * nx + r] = P[b].f[t * nx + r];   /* -b = +f */
        for (int r = 0; r < nx; ++r) mh[ni + (T - 1) * nx + r] = -P[b].x0[r];
        FN(ipm_kkt_factor)(&kk[b]);
        FN(ipm_solve_kkt)(&kk[b], P[b].p, zero, mh, mh + ni, x + (long)b * nz, s + (long)b * ni, z + (long)b * ni,
                          y + (long)b * ne, nb);
        free(mh);
        if (init_x) memcpy(init_x + (long)b * nz, x + (long)b * nz, sizeof(REAL) * nz);
        if (init_s) memcpy(init_s + (long)b * ni, s + (long)b * ni, sizeof(REAL) * ni);
        if (init_z) memcpy(init_z + (long)b * ni, z + (long)b * ni, sizeof(REAL) * ni);
        if (init_y) memcpy(init_y + (long)b * ne, y + (long)b * ne, sizeof(REAL) * ne);
        /* positivity shift (:71-81): if min < 0: v -= min - 1 */
        REAL *vv[2] = {s + (long)b * ni, z + (long)b * ni};
        for (int q = 0; q < 2; ++q) {
            REAL m = INFINITY;
            for (int i = 0; i < ni; ++i) if (vv[q][i] < m) m = vv[q][i];
            if (m < 0) for (int i = 0; i < ni; ++i) vv[q][i] -= m - 1;
        }
    }
    int have_best = 0, nNotImproved = 0, it_done = maxIter;
    for (int it = 0; it < maxIter; ++it) {
        /* ---- residuals (:86-103) */
        if (ry_cb) ry_cb(x, ry, cb_ctx);
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; ++b) {
            REAL *xb_ = x + (long)b * nz, *sb_ = s + (long)b * ni, *zb_ = z + (long)b * ni, *yb_ = y + (long)b * ne;
            REAL *rxb = rx + (long)b * nz, *rsb = rs + (long)b * ni, *rzb = rz + (long)b * ni, *ryb = ry + (long)b * ne;
            /* rx = A'y + G'z + Qx + p, via K*(x, 0, z, y) rows x (s part unused) */
            REAL *t_os = scr + (long)b * 3 * NK, *t_oz = t_os + ni, *t_oy = t_oz + ni;
            FN(ipm_Kmul)(&P[b], ones, ones, xb_, zero, zb_, yb_, rxb, t_os, t_oz, t_oy);
            for (int i = 0; i < nz; ++i) rxb[i] += P[b].p[i];
            for (int i = 0; i < ni; ++i) { rsb[i] = sb_[i] * zb_[i]; rzb[i] = t_oz[i] + sb_[i] - h[i]; }
            if (!ry_cb) {   /* A x - b */
                for (int t = 0; t < T - 1; ++t)
                    for (int r = 0; r < nx; ++r) ryb[t * nx + r] = t_oy[t * nx + r] + P[b].f[t * nx + r];
                for (int r = 0; r < nx; ++r) ryb[(T - 1) * nx + r] = t_oy[(T - 1) * nx + r] - P[b].x0[r];
            }
            REAL sz = 0, nzr = 0, nyr = 0, nxr = 0;
            for (int i = 0; i < ni; ++i) { sz += sb_[i] * zb_[i]; nzr += rzb[i] * rzb[i]; }
            for (int i = 0; i < ne; ++i) nyr += ryb[i] * ryb[i];
            for (int i = 0; i < nz; ++i) nxr += rxb[i] * rxb[i];
            mu[b] = FABS(sz / ni);
            resids[b] = SQRT(nyr) + SQRT(nzr) + SQRT(nxr) + ni * mu[b];
        }
        /* ---- best iterate + exit rule (:120-151) */
        int any = 0;
        for (int b = 0; b < B; ++b) {
            int better = !have_best || resids[b] < resid_b[b];
            if (better) {
                any = 1;
                resid_b[b] = resids[b];
                memcpy(xb + (long)b * nz, x + (long)b * nz, sizeof(REAL) * nz);
                memcpy(yb + (long)b * ne, y + (long)b * ne, sizeof(REAL) * ne);
                memcpy(zb + (long)b * ni, z + (long)b * ni, sizeof(REAL) * ni);
                memcpy(sb + (long)b * ni, s + (long)b * ni, sizeof(REAL) * ni);
                if (iter_best) iter_best[b] = it;
            }
        }
        if (!have_best) { have_best = 1; nNotImproved = 0; }
        else if (any) nNotImproved = 0;
        else nNotImproved++;
        if (exit_mode == 0) {
            REAL bmax = -INFINITY, mumin = INFINITY;
            for (int b = 0; b < B; ++b) { if (resid_b[b] > bmax) bmax = resid_b[b]; if (mu[b] < mumin) mumin = mu[b]; }
            if (nNotImproved == notImprovedLim || bmax < eps || mumin > 1e32) { it_done = it; break; }
        }
        /* ---- affine scaling direction (:153-159); Ktilde diag: z + KKTeps, s (:107-110) */
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; ++b) {
            kk[b].zd = z + (long)b * ni; kk[b].sd = s + (long)b * ni;
            FN(ipm_kkt_factor)(&kk[b]);
            FN(ipm_solve_kkt)(&kk[b], rx + (long)b * nz, rs + (long)b * ni, rz + (long)b * ni, ry + (long)b * ne,
                              dxa + (long)b * nz, dsa + (long)b * ni, dza + (long)b * ni, dya + (long)b * ne,
                              scr + (long)b * 3 * NK);
        }
        FN(ipm_get_step)(B, ni, z, dza, st1, exit_mode);
        FN(ipm_get_step)(B, ni, s, dsa, st2, exit_mode);
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; ++b) {
            REAL a = st1[b] < st2[b] ? st1[b] : st2[b];
            if (st1[b] != st1[b] || st2[b] != st2[b]) a = NAN;
            if (!(a < 1) && a == a) a = 1;           /* torch.min(a, 1): NaN stays NaN */
            REAL t3 = 0, t4 = 0;
            for (int i = 0; i < ni; ++i) {
                long k = (long)b * ni + i;
                t3 += (s[k] + a * dsa[k]) * (z[k] + a * dza[k]);
                t4 += s[k] * z[k];
            }
            REAL sig = t3 / t4; sig = sig * sig * sig;
            for (int i = 0; i < ni; ++i) { long k = (long)b * ni + i; rs[k] = -mu[b] * sig + dsa[k] * dza[k]; }
            /* centering-corrector (:173-180): rx = rz = ry = 0 */
            FN(ipm_solve_kkt)(&kk[b], zero, rs + (long)b * ni, zero, zero, dxc + (long)b * nz, dsc + (long)b * ni,
                              dzc + (long)b * ni, dyc + (long)b * ne, scr + (long)b * 3 * NK);
            for (int i = 0; i < nz; ++i) dxa[(long)b * nz + i] += dxc[(long)b * nz + i];
            for (int i = 0; i < ni; ++i) { dsa[(long)b * ni + i] += dsc[(long)b * ni + i]; dza[(long)b * ni + i] += dzc[(long)b * ni + i]; }
            for (int i = 0; i < ne; ++i) dya[(long)b * ne + i] += dyc[(long)b * ne + i];
            if (info && kk[b].info && !info[b]) info[b] = kk[b].info;
        }
        FN(ipm_get_step)(B, ni, z, dza, st1, exit_mode);
        FN(ipm_get_step)(B, ni, s, dsa, st2, exit_mode);
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; ++b) {
            REAL a = st1[b] < st2[b] ? st1[b] : st2[b];
            if (st1[b] != st1[b] || st2[b] != st2[b]) a = NAN;
            a = (REAL)0.999 * a;
            if (!(a < 1) && a == a) a = 1;
            alpha[b] = a;
            for (int i = 0; i < nz; ++i) x[(long)b * nz + i] += a * dxa[(long)b * nz + i];
            for (int i = 0; i < ni; ++i) { s[(long)b * ni + i] += a * dsa[(long)b * ni + i]; z[(long)b * ni + i] += a * dza[(long)b * ni + i]; }
            for (int i = 0; i < ne; ++i) y[(long)b * ne + i] += a * dya[(long)b * ne + i];
        }
    }
    (void)zpe;
    free(x); free(s); free(z); free(y); free(rx); free(rs); free(rz); free(ry);
    free(dxa); free(dsa); free(dza); free(dya); free(dxc); free(dsc); free(dzc); free(dyc);
    free(h); free(mu); free(resids); free(st1); free(st2); free(alpha); free(ones); free(zpe); free(zero);
    free(w); free(tmp); free(scr); free(lu); free(piv); free(P); free(kk);
    return it_done;
}

/*
 * Backward of DenseQPFunction (qp.py:238-270): solve_kkt(K, K, g, 0, 0, 0) with K at the returned
 * (best) iterate's lams / slacks - no regularisation - then dx, dlam (= dz), dnu (= dy).
 */
void FN(orc_ipm_backward)(int B, int T, int nx, int nu, const REAL *Qd, const REAL *F, const REAL *lams,
                          const REAL *slacks, const REAL *g, int solver, REAL *dx, REAL *dlam, REAL *dnu)
{
    int n = nx + nu, nz = T * n, ni = 2 * T * nu, ne = T * nx;
    long NK = nz + 2 * ni + ne;
    long wsz = nz + 3L * ni + 2L * ne + 2L * T * nx * nx + 16;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        FN(ipm_prob) P = {T, nx, nu, n, nz, ni, ne, Qd + (long)b * nz, NULL, F + (long)b * (T - 1) * nx * n, NULL, NULL, NULL, NULL};
        REAL *w = calloc(wsz, sizeof(REAL)), *tmp = calloc(2 * NK, sizeof(REAL)), *scr = calloc(3 * NK, sizeof(REAL));
        REAL *zero = calloc(NK, sizeof(REAL)), *ds = calloc(ni, sizeof(REAL));
        REAL *lu = solver == 1 ? calloc((size_t)NK * NK, sizeof(REAL)) : NULL;
        int *piv = solver == 1 ? calloc(NK, sizeof(int)) : NULL;
        FN(ipm_kkt) k = {&P, solver, 0, 0, 0, 0, lams + (long)b * ni, slacks + (long)b * ni, lu, piv, w, tmp, 0};
        FN(ipm_kkt_factor)(&k);
        FN(ipm_solve_kkt)(&k, g + (long)b * nz, zero, zero, zero, dx + (long)b * nz, ds, dlam + (long)b * ni,
                          dnu + (long)b * ne, scr);
        free(w); free(tmp); free(scr); free(zero); free(ds); free(lu); free(piv);
    }
}
