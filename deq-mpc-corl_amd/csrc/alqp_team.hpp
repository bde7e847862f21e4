// alqp_team.hpp - device-side "team" that solves ONE QP instance inside a wavefront.
//
// MI355X / gfx950 only. A team is G consecutive lanes of a 64-lane wavefront
// (G = 8/16/32/64, the smallest power of two holding NROWS = 2n+nx+1 lanes);
// 64/G instances share a wavefront and a workgroup is exactly one wavefront, so
// every synchronisation below is wave-local.
//
// The Newton system H d = -g of the AL merit (qpth/al_utils.py:80-123) is block
// tridiagonal: n x n diagonal blocks H_tt and sub-diagonal blocks
// H_{t+1,t} = -rho E'F_t whose only non-zero rows are the first nx. The team sweeps the
// horizon once forward (factor + forward substitution) and once backward.
//
// Forward stage t is ONE left-looking panel factorisation with a lane per panel row:
//   lanes [0,n)            rows of H_tt            -> L_tt            (Cholesky)
//   lanes [n,n+nx)         rows of -rho F_t        -> W_t = H_{t+1,t} L_tt^{-T}
//   lane  n+nx             the right-hand side -g_t -> y_t            (forward subst.)
//   lanes (n+nx,2n+nx]     rows of the identity    -> X_t = L_tt^{-T} (explicit inverse)
// Every lane does the same arithmetic on its own row held in registers:
//   row[j] = (row0[j] + rho*<Fcol_row,Fcol_j> - <Wprev_row,Wprev_j> - sum_{k<j} row[k] L[j][k]) / L[j][j]
// and only "row j" operands are broadcast (LDS reads at a wave-uniform address, no
// bank conflicts). H_tt is never materialised: its F'F term and the Schur complement
// W_{t-1}W_{t-1}' of the previous stage are folded into the same dot products.
// Only X_t (packed, n(n+1)/2 words per stage) is kept in LDS for the backward sweep,
// which is then pure mat-vec work (no serial triangular solves):
//   d_t = X_t ( y_t + rho * X_t' F_t' dx_{t+1} ).
#pragma once
#include <hip/hip_runtime.h>

namespace alqp {

constexpr int pad4(int v) { return (v + 3) & ~3; }

template <typename real, int NX_, int NU_>
struct Cfg {
    static constexpr int NX = NX_, NU = NU_, N = NX_ + NU_;
    static constexpr int NP = pad4(N), NXP = pad4(NX_);
    static constexpr int NROWS = 2 * N + NX + 1;
    static_assert(NROWS <= 64, "one QP must fit a wavefront");
    static constexpr int G = NROWS <= 8 ? 8 : NROWS <= 16 ? 16 : NROWS <= 32 ? 32 : 64;
    static constexpr int QPW = 64 / G;          // instances per wavefront
    static constexpr int XT = N * (N + 1) / 2;  // packed upper triangle of X_t
    // per-team LDS scratch, offsets in reals (all multiples of 4 -> 16-byte aligned)
    static constexpr int oFs = 0;                      // raw F_t            [NX*N]
    static constexpr int oFt = oFs + pad4(NX * N);     // F_t transposed     [N][NXP]
    static constexpr int oWb = oFt + N * NXP;          // W_{t-1}, W_t       [2][NX][NP]
    static constexpr int oLw = oWb + 2 * NX * NP;      // rows of L_tt       [N][NP]
    static constexpr int oVs = oLw + N * NP;           // nx-vector          [NXP]
    static constexpr int oGs = oVs + NXP;              // n-vector           [NP]
    static constexpr int oRs = oGs + NP;               // n-vector           [NP]
    static constexpr int SCRATCH = oRs + NP;
    __host__ __device__ static constexpr int M(int T) { return T * NX + 2 * T * NU; }
    // persistent arrays: z, d (y), r_eq, s_eq = (J d)_eq, lam, X
    __host__ __device__ static constexpr int team_words(int T) {
        return SCRATCH + 2 * pad4(T * N) + 2 * pad4(T * NX) + pad4(M(T)) + pad4(T * XT);
    }
};

// ---- cross-lane helpers ----------------------------------------------------------

template <int G>
__device__ inline float team_bcast(float v, int src_in_team, int team_base) {
    if constexpr (G == 64) {
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_in_team));
    } else {
        return __shfl(v, team_base + src_in_team, 64);
    }
}
template <int G>
__device__ inline double team_bcast(double v, int src_in_team, int team_base) {
    if constexpr (G == 64) {
        long long b = __builtin_bit_cast(long long, v);
        int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_in_team);
        int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_in_team);
        long long r = ((long long)hi << 32) | (unsigned int)lo;
        return __builtin_bit_cast(double, r);
    } else {
        return __shfl(v, team_base + src_in_team, 64);
    }
}
template <int G, typename real>
__device__ inline real team_sum(real v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int G>
__device__ inline int team_or(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}

__device__ inline void wave_sync() { __syncthreads(); }  // workgroup == one wavefront

// ---- the team -------------------------------------------------------------------

template <typename real, int NX, int NU>
struct Team {
    using C = Cfg<real, NX, NU>;
    static constexpr int N = C::N, NP = C::NP, NXP = C::NXP, G = C::G, XT = C::XT;

    // LDS
    real *Fs, *Ft, *Wb, *Lw, *vs, *gs, *rs;
    real *zs, *ds, *req, *seq, *lams, *Xp;
    // identity
    int li, team_base, T, b;
    bool isH, isW, isY, isU;
    int hi, wr, ui, jmin, jmax;
    // problem (global memory, this instance)
    const real *gQd, *gq, *gF, *gc, *gx0, *gulo, *guhi, *gxnext;
    long st_u;
    real rho;
    int info;

    __device__ void init(real *lds_team, int lane_in_team, int team_base_, int T_, int b_) {
        li = lane_in_team; team_base = team_base_; T = T_; b = b_;
        Fs = lds_team + C::oFs; Ft = lds_team + C::oFt; Wb = lds_team + C::oWb;
        Lw = lds_team + C::oLw; vs = lds_team + C::oVs; gs = lds_team + C::oGs; rs = lds_team + C::oRs;
        real *p = lds_team + C::SCRATCH;
        zs = p; p += pad4(T * N);
        ds = p; p += pad4(T * N);
        req = p; p += pad4(T * NX);
        seq = p; p += pad4(T * NX);
        lams = p; p += pad4(C::M(T));
        Xp = p;
        isH = li < N; isW = li >= N && li < N + NX; isY = li == N + NX;
        isU = li > N + NX && li < C::NROWS;
        hi = isH ? li : 0; wr = isW ? li - N : 0; ui = isU ? li - (N + NX + 1) : 0;
        // row entries kept after the column step j: jmin <= j <= jmax
        jmin = isU ? ui : 0;
        jmax = isH ? hi : ((isW || isY || isU) ? N - 1 : -1);
        info = 0;
        gxnext = nullptr;
    }

    __device__ real uhi(int t, int j) const { return guhi[t * st_u + j]; }
    __device__ real ulo(int t, int j) const { return gulo[t * st_u + j]; }

    // coalesced copy of F_t into LDS (raw row-major [NX][N])
    __device__ void load_F(int t) {
        const real *Fg = gF + (size_t)t * NX * N;
        for (int e = li; e < NX * N; e += G) Fs[e] = Fg[e];
    }

    // equality residual of every stage at the current zs -> req (kernel start)
    __device__ void residual_sweep() {
        for (int t = 0; t < T - 1; ++t) {
            real xn = 0;
            if (gxnext) {
                if (isW) xn = gxnext[t * NX + wr];
            } else {
                load_F(t);
                wave_sync();
                if (isW) {
                    real s = 0;
#pragma unroll
                    for (int k = 0; k < N; ++k) s += Fs[wr * N + k] * zs[t * N + k];
                    xn = s + gc[t * NX + wr];
                }
            }
            if (isW) req[t * NX + wr] = zs[(t + 1) * N + wr] - xn;
            wave_sync();
        }
        if (li < NX) req[(T - 1) * NX + li] = zs[li] - gx0[li];
        wave_sync();
    }

    // Forward sweep: gradient, factorisation and forward substitution, stage by stage.
    // Leaves y_t in ds[t], X_t in Xp[t], r_eq in req. g_out (nullable, global [T][N]).
    __device__ void forward_sweep(real *g_out) {
        real l[N], fa[N], bb[N];
#pragma unroll
        for (int k = 0; k < N; ++k) bb[k] = 0;
        // initial-state residual (eq row block T-1), al_utils.py:274
        if (li < NX) req[(T - 1) * NX + li] = zs[li] - gx0[li];
        for (int t = 0; t < T; ++t) {
            const bool dyn = t < T - 1;
            const int cur = t & 1, prev = cur ^ 1;
            real *Wc = Wb + cur * NX * NP;
            const real *Wp = Wb + prev * NX * NP;
            if (dyn) load_F(t);
            wave_sync();
            // ---- own F row (W lanes) / F column (H lanes) into registers
#pragma unroll
            for (int k = 0; k < N; ++k) fa[k] = 0;
            if (dyn) {
                if (isW) {
#pragma unroll
                    for (int k = 0; k < N; ++k) fa[k] = Fs[wr * N + k];
                } else if (isH) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) fa[r] = Fs[r * N + hi];
#pragma unroll
                    for (int r = 0; r < NXP; ++r) Ft[hi * NXP + r] = r < NX ? fa[r] : real(0);
                }
            }
            // ---- dynamics residual r_t and multiplier estimate v = lam + rho r (W lanes)
            if (dyn && isW) {
                real xn;
                if (gxnext) {
                    xn = gxnext[t * NX + wr];
                } else {
                    real s = 0;
#pragma unroll
                    for (int k = 0; k < N; ++k) s += fa[k] * zs[t * N + k];
                    xn = s + gc[t * NX + wr];
                }
                real r = zs[(t + 1) * N + wr] - xn;
                req[t * NX + wr] = r;
                vs[wr] = lams[t * NX + wr] + rho * r;
            }
            wave_sync();
            // ---- gradient entry and diagonal of H_tt (H lanes), al_utils.py:113-120
            real D = 0;
            if (isH) {
                real zv = zs[t * N + hi];
                real Qv = gQd[t * N + hi];
                real g = Qv * zv + gq[t * N + hi];
                D = Qv;
                if (hi < NX) {
                    int row = (t == 0) ? (T - 1) * NX + hi : (t - 1) * NX + hi;
                    g += lams[row] + rho * req[row];
                    D += rho;
                } else {
                    int j = hi - NX;
                    int ru = T * NX + t * 2 * NU + j, rl = ru + NU;
                    real vu = zv - uhi(t, j), vl = -zv + ulo(t, j);
                    real au = vu >= 0 ? real(1) : real(0), al = vl >= 0 ? real(1) : real(0);
                    D += rho * (au + al);
                    g += (lams[ru] + rho * (vu > 0 ? vu : real(0))) - (lams[rl] + rho * (vl > 0 ? vl : real(0)));
                }
                if (dyn) {
                    real s = 0;
#pragma unroll
                    for (int r = 0; r < NX; ++r) s += fa[r] * vs[r];
                    g -= s;
                }
                gs[hi] = g;
                if (g_out) g_out[t * N + hi] = g;
            }
            // ---- Schur-complement operand rows: W_{t-1} row (H lanes < NX), y_{t-1} (Y lane)
            if (t > 0 && isH && hi < NX) {
#pragma unroll
                for (int k = 0; k < N; ++k) bb[k] = Wp[hi * NP + k];
            }
            wave_sync();
            // ---- initial row values
#pragma unroll
            for (int k = 0; k < N; ++k) {
                real v = 0;
                if (isH) v = (k == hi) ? D : real(0);
                else if (isW) v = -rho * fa[k];
                else if (isY) v = -gs[k];
                else if (isU) v = (k == ui) ? real(1) : real(0);
                l[k] = v;
            }
            // ---- column steps (left-looking)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                real acc = l[j];
                if (dyn) {
                    real s = 0;
#pragma unroll
                    for (int r = 0; r < NX; ++r) s += fa[r] * Ft[j * NXP + r];
                    acc += isH ? rho * s : real(0);
                }
                if (t > 0 && j < NX) {
                    real s = 0;
#pragma unroll
                    for (int k = 0; k < N; ++k) s += bb[k] * Wp[j * NP + k];
                    acc -= s;
                }
                {
                    real s = 0;
#pragma unroll
                    for (int k = 0; k < j; ++k) s += l[k] * Lw[j * NP + k];
                    acc -= s;
                }
                real p = team_bcast<G>(acc, j, team_base);
                if (!(p > 0) && info == 0) info = t * N + j + 1;
                real rinv = real(1) / sqrt(p);
                real v = (j >= jmin && j <= jmax) ? acc * rinv : real(0);
                l[j] = v;
                if (isH) Lw[hi * NP + j] = v;
                else if (isW) Wc[wr * NP + j] = v;
                wave_sync();
            }
            // ---- stage results
            if (isY) {
#pragma unroll
                for (int k = 0; k < N; ++k) ds[t * N + k] = l[k];
            }
            if (isU) {
                real *Xr = Xp + (size_t)t * XT + (ui * N - (ui * (ui - 1)) / 2);
#pragma unroll
                for (int k = 0; k < N; ++k)
                    if (k >= ui) Xr[k - ui] = l[k];
            }
            // Y lane carries y_t as the Schur operand of the next right-hand side
#pragma unroll
            for (int k = 0; k < N; ++k) bb[k] = isY ? l[k] : real(0);
            wave_sync();
        }
    }

    // Backward sweep: d_t = X_t ( y_t + rho X_t' F_t' dx_{t+1} ), and s = (J d)_eq.
    __device__ void backward_sweep() {
        for (int t = T - 1; t >= 0; --t) {
            const bool dyn = t < T - 1;
            const real *Xt = Xp + (size_t)t * XT;
            real frow[N];
            if (dyn) {
                load_F(t);
                wave_sync();
                // v = F_t' dx_{t+1}
                if (isH) {
                    real s = 0;
#pragma unroll
                    for (int r = 0; r < NX; ++r) s += Fs[r * N + hi] * ds[(t + 1) * N + r];
                    gs[hi] = s;
                }
                if (isW) {
#pragma unroll
                    for (int k = 0; k < N; ++k) frow[k] = Fs[wr * N + k];
                }
                wave_sync();
            }
            // rhs_j = y_j + rho * sum_{i<=j} X[i][j] v_i
            if (isH) {
                real rhs = ds[t * N + hi];
                if (dyn) {
                    real s = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        if (i <= hi) s += Xt[(i * N - (i * (i - 1)) / 2) + (hi - i)] * gs[i];
                    rhs += rho * s;
                }
                rs[hi] = rhs;
            }
            wave_sync();
            // d_i = sum_{j>=i} X[i][j] rhs_j
            if (isH) {
                const real *Xr = Xt + (hi * N - (hi * (hi - 1)) / 2);
                real s = 0;
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j >= hi) s += Xr[j - hi] * rs[j];
                ds[t * N + hi] = s;
            }
            wave_sync();
            if (dyn && isW) {
                real s = 0;
#pragma unroll
                for (int k = 0; k < N; ++k) s += frow[k] * ds[t * N + k];
                seq[t * NX + wr] = ds[(t + 1) * N + wr] - s;
            }
        }
        if (li < NX) seq[(T - 1) * NX + li] = ds[li];
        wave_sync();
    }

    // Forward substitution only, with the factor already in Xp: ds[t] (rhs) -> y_t.
    // W_{t-1} y_{t-1} = -rho F_{t-1} X_{t-1} y_{t-1}.
    __device__ void forward_solve_only() {
        for (int t = 0; t < T; ++t) {
            const real *Xt = Xp + (size_t)t * XT;
            if (t > 0) {
                const real *Xq = Xp + (size_t)(t - 1) * XT;
                load_F(t - 1);
                // e = X_{t-1} y_{t-1}
                if (isH) {
                    const real *Xr = Xq + (hi * N - (hi * (hi - 1)) / 2);
                    real s = 0;
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        if (j >= hi) s += Xr[j - hi] * ds[(t - 1) * N + j];
                    gs[hi] = s;
                }
                wave_sync();
                if (isW) {
                    real s = 0;
#pragma unroll
                    for (int k = 0; k < N; ++k) s += Fs[wr * N + k] * gs[k];
                    ds[t * N + wr] += rho * s;
                }
                wave_sync();
            }
            if (isH) rs[hi] = ds[t * N + hi];
            wave_sync();
            if (isH) {
                real s = 0;
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (i <= hi) s += Xt[(i * N - (i * (i - 1)) / 2) + (hi - i)] * rs[i];
                ds[t * N + hi] = s;
            }
            wave_sync();
        }
    }

    // Merit of K candidates z + alpha_k d (alpha_k = 2^-k), al_utils.py:73-77.
    // Equality residuals move linearly along d: r + alpha s (affine dynamics).
    // Every lane returns all K sums. K = 1 evaluates the merit at z itself.
    template <int K>
    __device__ void merit_candidates(real (&phi)[K], bool at_z) {
        real acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0;
        const int neq = T * NX;
        for (int e = li; e < T * N; e += G) {
            int t = e / N, j = e - t * N;
            real z = zs[e], d = at_z ? real(0) : ds[e];
            real Qv = gQd[e], qv = gq[e];
            bool isu = j >= NX;
            real lu = 0, ll = 0, bu = 0, bl = 0;
            if (isu) {
                int ru = neq + t * 2 * NU + (j - NX);
                lu = lams[ru]; ll = lams[ru + NU];
                bu = uhi(t, j - NX); bl = ulo(t, j - NX);
            }
            real alpha = 1;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                real zk = z + alpha * d;
                real v = (real(0.5) * Qv * zk + qv) * zk;
                if (isu) {
                    real vu = zk - bu, vl = -zk + bl;
                    real cu = vu > 0 ? vu : real(0), cl = vl > 0 ? vl : real(0);
                    v += lu * vu + ll * vl + real(0.5) * rho * (cu * cu + cl * cl);
                }
                acc[k] += v;
                alpha *= real(0.5);
            }
        }
        for (int e = li; e < neq; e += G) {
            real r = req[e], s = at_z ? real(0) : seq[e], lm = lams[e];
            real alpha = 1;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                real rk = r + alpha * s;
                acc[k] += lm * rk + real(0.5) * rho * rk * rk;
                alpha *= real(0.5);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) phi[k] = team_sum<G>(acc[k]);
    }

    // sum r+(z)^2 at the current zs/req (al_utils.py:552 sums this over the batch)
    __device__ real rplus2() {
        real acc = 0;
        for (int e = li; e < T * NX; e += G) acc += req[e] * req[e];
        for (int e = li; e < T * NU; e += G) {
            int t = e / NU, j = e - t * NU;
            real u = zs[t * N + NX + j];
            real vu = u - uhi(t, j), vl = -u + ulo(t, j);
            real cu = vu > 0 ? vu : real(0), cl = vl > 0 ? vl : real(0);
            acc += cu * cu + cl * cl;
        }
        return team_sum<G>(acc);
    }

    // lam <- lam + rho r ; lam_ineq <- max(0, .)   (AL_mpc.py:316-317)
    __device__ void dual_update() {
        const int neq = T * NX;
        for (int e = li; e < neq; e += G) lams[e] += rho * req[e];
        for (int e = li; e < T * NU; e += G) {
            int t = e / NU, j = e - t * NU;
            real u = zs[t * N + NX + j];
            int ru = neq + t * 2 * NU + j, rl = ru + NU;
            real a = lams[ru] + rho * (u - uhi(t, j));
            real c = lams[rl] + rho * (-u + ulo(t, j));
            lams[ru] = a < 0 ? real(0) : a;
            lams[rl] = c < 0 ? real(0) : c;
        }
        wave_sync();
    }
};

}  // namespace alqp
