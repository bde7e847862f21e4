// alqp_team.hpp - device-side "team" that solves ONE QP instance inside a wavefront.
//
// MI355X / gfx950 only. A team is G consecutive lanes of a 64-lane wavefront
// (G = 8/16/32/64, the smallest power of two holding NROWS = 2n+nx+1 lanes);
// 64/G instances share a wavefront and a workgroup is exactly one wavefront, so
// every synchronisation below is wave-local: LDS instructions of one wave execute in
// issue order, so a ds_write followed by a ds_read of another lane's word needs no
// barrier, only a compiler fence (wave_sync) that keeps the two in program order.
//
// The Newton system H d = -g of the AL merit (qpth/al_utils.py:80-123) is block
// tridiagonal: n x n diagonal blocks H_tt and sub-diagonal blocks
// H_{t+1,t} = -rho E'F_t whose only non-zero rows are the first nx. The team sweeps the
// horizon once forward (factor + forward substitution) and once backward.
//
// Forward stage t:
//  (1) SYRK phase, 2x2 output blocks spread over the lanes (lower triangle only):
//        Hs = rho * Ft Ft' - Sb Sb'
//      with Ft = [F_t' ; 0] (rows = columns of F_t) and Sb = [W_{t-1} ; 0 ; y_{t-1}'],
//      i.e. the F'F term of H_tt, the Schur complement W_{t-1}W_{t-1}' of the previous
//      stage AND the coupling W_{t-1} y_{t-1} of the right-hand side in one pass.
//      Operands are read from LDS as 16-byte vectors.
//  (2) panel phase, a lane per panel row, rows in registers:
//        lanes [0,n)          rows of H_tt          -> L_tt            (Cholesky)
//        lanes [n,n+nx)       rows of -rho F_t      -> W_t = H_{t+1,t} L_tt^{-T}
//        lane  n+nx           right-hand side -g_t  -> y_t             (forward subst.)
//        lanes (n+nx,2n+nx]   rows of the identity  -> X_t = L_tt^{-T} (explicit inverse)
//      left-looking: row[j] = (row[j] - sum_{k<j} row[k] L[j][k]) / L[j][j]; the L[j][k]
//      operands come straight out of lane j's registers (v_readlane, no LDS round trip).
// Only X_t (packed, n(n+1)/2 words per stage) is kept in LDS for the backward sweep,
// which is then pure mat-vec work (no serial triangular solves):
//        d_t = X_t ( y_t + rho * X_t' F_t' dx_{t+1} ).
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
    static constexpr int NB = (N + 2) / 2;      // 2x2 block rows covering rows 0..N (N = rhs row)
    static constexpr int RB = 2 * NB;
    static constexpr int NBLK = NB * (NB + 1) / 2;
    static constexpr int HP = pad4(RB);         // row stride of Hs: the 2x2 blocks span RB columns
    static_assert(NBLK <= G, "SYRK blocks must fit the team");
    // per-team LDS scratch, offsets in reals (all multiples of 4 -> 16-byte aligned)
    static constexpr int oFt = 0;                      // [F_t' ; 0]                 [RB][NXP]
    static constexpr int oSb = oFt + RB * NXP;         // [W_{t-1} ; 0 ; y_{t-1}']   [RB][NP]  (also s_eq)
    static constexpr int oHs = oSb + RB * NP;          // SYRK result                [RB][HP]  (also raw F_t)
    static constexpr int oVs = oHs + RB * HP;          // nx-vector                  [NXP]
    static constexpr int oGs = oVs + NXP;              // n-vector                   [NP]
    static constexpr int oRs = oGs + NP;               // n-vector                   [NP]
    static constexpr int SCRATCH = oRs + NP;
    static_assert(pad4(NX * N) <= RB * HP, "raw F_t is staged in the Hs region");
    __host__ __device__ static constexpr int M(int T) { return T * NX + 2 * T * NU; }
    // s_eq = (J d)_eq lives in the Sb region between the backward sweep and the line
    // search when it fits (short horizons), else in its own array
    __host__ __device__ static constexpr bool seq_in_scratch(int T) { return T * NX <= RB * NP; }
    // persistent arrays: z, d (y), r_eq, [s_eq], X      (lam stays in global memory / L2)
    __host__ __device__ static constexpr int team_words(int T) {
        return SCRATCH + 2 * pad4(T * N) + pad4(T * NX) + (seq_in_scratch(T) ? 0 : pad4(T * NX)) +
               pad4(T * XT);
    }
};

// ---- small helpers ---------------------------------------------------------------

__device__ inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1/sqrt(p): fp32 uses v_rsq_f32 (<= 1 ulp), fp64 the correctly rounded sequence
__device__ inline float fabs_(float a) { return __builtin_fabsf(a); }
__device__ inline double fabs_(double a) { return __builtin_fabs(a); }
__device__ inline float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ inline double fmax_(double a, double b) { return __builtin_fmax(a, b); }
__device__ inline float rsqrt_(float p) { return __builtin_amdgcn_rsqf(p); }
// fp64: hardware estimate + two coupled Newton steps (g -> sqrt p, h -> 1 / (2 sqrt p)), ~1 ulp, 9 instructions. The
// correctly rounded 1.0 / sqrt(p) is a 30-instruction dependent sequence (scaled square root, then an IEEE division)
// on the critical path of every pivot of the panel: half of a stage's cycles at the reference's batch size.
__device__ inline double rsqrt_(double p) {
    const double y = __builtin_amdgcn_rsq(p);
    double g = p * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, r, h);
    return h + h;
}
__device__ inline float rcp_(float p) { return __builtin_amdgcn_rcpf(p); }
// fp64: hardware estimate + two Newton steps (~1 ulp) instead of the IEEE division sequence
__device__ inline double rcp_(double p) {
    double r = __builtin_amdgcn_rcp(p);
    r = __builtin_fma(__builtin_fma(-p, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-p, r, 1.0), r, r);
    return r;
}

// v if d >= 0 else +0, without a lane-mask compare (loop-invariant compares get hoisted
// into SGPR pairs by the compiler and then spilled: 2 SGPRs per unrolled index)
__device__ inline float keep_if_nonneg(float v, int d) {
    return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & ~(d >> 31));
}
__device__ inline double keep_if_nonneg(double v, int d) {
    return __builtin_bit_cast(double, __builtin_bit_cast(long long, v) & ~(long long)(d >> 31));
}

// 16-byte LDS vector access (p must be 16-byte aligned)
__device__ inline void ld4(const float *p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ inline void ld4(const double *p, double (&v)[4]) {
    double2 a = *reinterpret_cast<const double2 *>(p);
    double2 b = *reinterpret_cast<const double2 *>(p + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ inline void st4(float *p, float a, float b, float c, float d) {
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
}
__device__ inline void st4(double *p, double a, double b, double c, double d) {
    *reinterpret_cast<double2 *>(p) = make_double2(a, b);
    *reinterpret_cast<double2 *>(p + 2) = make_double2(c, d);
}

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
// A 32-bit move on the DPP network (row_shr:n = 0x110 + n, row_bcast15 = 0x142, row_bcast31 = 0x143); lanes whose source
// is outside the row / not in `rows` read 0.
template <int CTRL, int ROWS>
__device__ inline float dpp_zfill(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xf, true));
}
template <int CTRL, int ROWS>
__device__ inline double dpp_zfill(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROWS, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROWS, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int G, typename real>
__device__ inline real team_sum(real v) {
    if constexpr (G == 64) {
        // whole-wavefront team: scan inside the 16-lane rows, rows 0 / 2 into 1 / 3, the lower half into the upper - on the
        // DPP network, no LDS round trip per step (the shuffle version: six exposed ds_bpermute round trips per sum, and the
        // 20-candidate merit takes 23 sums per Newton step); lane 63 holds the total
        v += dpp_zfill<0x111, 0xf>(v);
        v += dpp_zfill<0x112, 0xf>(v);
        v += dpp_zfill<0x114, 0xf>(v);
        v += dpp_zfill<0x118, 0xf>(v);
        v += dpp_zfill<0x142, 0xa>(v);
        v += dpp_zfill<0x143, 0xc>(v);
        return team_bcast<64>(v, 63, 0);
    } else {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
}
template <int G>
__device__ inline int team_or(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}

// Compiler-only fence: LDS traffic of one wave is executed in order by the hardware.
__device__ inline void wave_sync() { asm volatile("" ::: "memory"); }

// ---- the team -------------------------------------------------------------------

template <typename real, int NX, int NU>
struct Team {
    using C = Cfg<real, NX, NU>;
    static constexpr int N = C::N, NP = C::NP, NXP = C::NXP, G = C::G, XT = C::XT;
    static constexpr int RB = C::RB, HP = C::HP;

    // LDS
    real *Fs, *Ft, *Sb, *Hs, *vs, *gs, *rs;
    real *zs, *ds, *req, *seq, *Xp;
    real *lams;  // GLOBAL memory (this instance's multipliers, updated in place)
    real ufl;    // float(ui) for identity rows, -100 elsewhere
    // identity
    int li, team_base, T, b;
    bool isH, isW, isY, isU, isBlk;
    int hi, wr, ui, bi, bj;
    // problem (global memory, this instance)
    const real *gQd, *gq, *gF, *gc, *gx0, *gulo, *guhi, *gxnext;
    long st_u;
    real rho;
    int info;
    // obstacle rows (Obstacle_MPC, qpth/AL_mpc_custom.py; al_utils.py:313-323, 351-388): nobs spheres per
    // stage, c_k = r^2 - |x_t[0:3] - o_k|^2 <= 0, behind the stage's 2 NU bound rows. 0 = plain problem.
    const real *gobs;   // this instance's centres [T][nobs][3] (global memory)
    int nobs;
    real obs_r2;
    bool no_init;       // state-estimator variant: no initial-state rows, zero cost gradient on u (al_utils_se.py:186-200, 300-310)

#ifdef ALQP_PHASE_TIMING
    // debug build only (tools/team_timing.py): cycles per phase of the team kernel. Buckets: 0 stage inputs + gradient,
    // 1 SYRK, 2 panel, 3 stage results, 4 backward sweep, 5 line-search merits, 6 pick + apply, 7 everything else
    unsigned long long tacc[8], tlast;
    __device__ __forceinline__ void stamp(int bucket) {
        unsigned long long now;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (bucket >= 0) tacc[bucket] += now - tlast;
        tlast = now;
    }
#define TSTAMP(b) stamp(b)
#else
#define TSTAMP(b)
#endif

    // `lds_team` must not be provably wave-uniform (see the kernels): uniform LDS reads
    // get scalarised by the compiler into ds_read + v_readfirstlane + SGPR-spill chains.
    __device__ void init(real *lds_team, int lane_in_team, int team_base_, int T_, int b_) {
        li = lane_in_team; team_base = team_base_; T = T_; b = b_;
        Ft = lds_team + C::oFt; Sb = lds_team + C::oSb; Hs = lds_team + C::oHs;
        Fs = Hs;
        vs = lds_team + C::oVs; gs = lds_team + C::oGs; rs = lds_team + C::oRs;
        real *p = lds_team + C::SCRATCH;
        zs = p; p += pad4(T * N);
        ds = p; p += pad4(T * N);
        req = p; p += pad4(T * NX);
        if (C::seq_in_scratch(T)) seq = Sb; else { seq = p; p += pad4(T * NX); }
        Xp = p;
        lams = nullptr;
        isH = li < N; isW = li >= N && li < N + NX; isY = li == N + NX;
        isU = li > N + NX && li < C::NROWS;
        hi = isH ? li : 0; wr = isW ? li - N : 0; ui = isU ? li - (N + NX + 1) : 0;
        ufl = isU ? real(ui) : real(-100);
        // SYRK block owned by this lane: lower-triangular enumeration (bi >= bj)
        isBlk = li < C::NBLK;
        bi = 0; bj = 0;
        {
            int rem = isBlk ? li : 0;
            while (rem > bi) { rem -= bi + 1; ++bi; }
            bj = rem;
        }
        info = 0;
        gxnext = nullptr;
        gobs = nullptr; nobs = 0; obs_r2 = 0; no_init = false;
        // constant zero rows / pads of the SYRK operands
        for (int e = li; e < RB * NXP; e += G) Ft[e] = 0;
        for (int e = li; e < RB * NP; e += G) Sb[e] = 0;
        for (int e = li; e < RB * HP; e += G) Hs[e] = 0;
        for (int e = li; e < NXP; e += G) vs[e] = 0;
        for (int e = li; e < NP; e += G) gs[e] = 0;
        wave_sync();
    }

    __device__ real uhi(int t, int j) const { return guhi[t * st_u + j]; }
    __device__ real ulo(int t, int j) const { return gulo[t * st_u + j]; }

    static constexpr int FCH = (NX * N + G - 1) / G;  // F words per lane

    __device__ void fetch_F(int t, real (&buf)[FCH]) const {
        const real *Fg = gF + (size_t)t * NX * N;
#pragma unroll
        for (int i = 0; i < FCH; ++i) {
            int e = li + i * G;
            buf[i] = e < NX * N ? Fg[e] : real(0);
        }
    }
    __device__ void stash_F(const real (&buf)[FCH]) {
#pragma unroll
        for (int i = 0; i < FCH; ++i) {
            int e = li + i * G;
            if (e < NX * N) Fs[e] = buf[i];
        }
    }
    __device__ void load_F(int t) {
        real buf[FCH];
        fetch_F(t, buf);
        stash_F(buf);
    }

    // equality residual of every stage at the current zs -> req (kernel start)
    __device__ void residual_sweep() {
        real fbuf[FCH];
        if (!gxnext && T > 1) fetch_F(0, fbuf);
        for (int t = 0; t < T - 1; ++t) {
            real xn = 0;
            if (gxnext) {
                if (isW) xn = gxnext[t * NX + wr];
            } else {
                stash_F(fbuf);                               // F_t, fetched a stage ahead (its global round trip
                wave_sync();                                 // runs under the previous stage's products)
                if (t + 2 < T) fetch_F(t + 1, fbuf);
                if (isW) {
                    real s = gc[t * NX + wr];
                    real fv[N], zv[N];   // operands first, products after (see backward_sweep)
#pragma unroll
                    for (int k = 0; k < N; ++k) { fv[k] = Fs[wr * N + k]; zv[k] = zs[t * N + k]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < N; ++k) s = fma_(fv[k], zv[k], s);
                    xn = s;
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
        real l[N], fa[N];
        real fbuf[FCH];
        // initial-state residual (eq row block T-1), al_utils.py:274
        if (li < NX) req[(T - 1) * NX + li] = no_init ? real(0) : zs[li] - gx0[li];
        fetch_F(0, fbuf);
        real Qn = isH ? gQd[hi] : real(0), qn = isH ? gq[hi] : real(0);
        real cn = isW ? (gxnext ? gxnext[wr] : gc[wr]) : real(0);
        // multipliers this lane needs at stage t (global memory, prefetched a stage ahead):
        //   x rows: lam of the eq row that pins x_t (init row for t = 0, dynamics row t-1)
        //   u rows: upper / lower bound multipliers;  W lanes: lam of dynamics row t
        const bool isHx = isH && hi < NX, isHu = isH && hi >= NX;
        real lan = 0, lbn = 0;
        if (isHx) lan = lams[(T - 1) * NX + hi];
        if (isHu) { lan = lams[T * NX + (hi - NX)]; lbn = lams[T * NX + NU + (hi - NX)]; }
        if (isW) lan = lams[wr];
        // Sb may hold s_eq of the previous Newton step: restore its constant zero rows
        for (int e = li; e < RB * NP; e += G) Sb[e] = 0;
        TSTAMP(7);
        for (int t = 0; t < T; ++t) {
            const bool dyn = t < T - 1;
            const real Qv = Qn, qv = qn, cv = cn, la = lan, lb = lbn;
            if (dyn) stash_F(fbuf);
            wave_sync();
            // ---- own F row (W lanes) / F column (H lanes) into registers
#pragma unroll
            for (int k = 0; k < N; ++k) fa[k] = 0;
            if (dyn && (isW || isH)) {
                const int base = isW ? wr * N : hi, stride = isW ? 1 : N;
#pragma unroll
                for (int k = 0; k < N; ++k)
                    if (isW || k < NX) fa[k] = Fs[base + k * stride];
                if (isH) {
#pragma unroll
                    for (int r4 = 0; r4 < NXP; r4 += 4)
                        st4(Ft + hi * NXP + r4, fa[r4], r4 + 1 < NX ? fa[r4 + 1] : real(0),
                            r4 + 2 < NX ? fa[r4 + 2] : real(0), r4 + 3 < NX ? fa[r4 + 3] : real(0));
                }
            }
            wave_sync();
            // ---- prefetch the next stage's inputs (they stay in flight during this stage)
            if (t + 2 < T) fetch_F(t + 1, fbuf);
            if (t + 1 < T && isH) { Qn = gQd[(t + 1) * N + hi]; qn = gq[(t + 1) * N + hi]; }
            if (t + 2 < T && isW) {
                cn = gxnext ? gxnext[(t + 1) * NX + wr] : gc[(t + 1) * NX + wr];
                lan = lams[(t + 1) * NX + wr];
            }
            if (t + 1 < T) {
                if (isHx) lan = lams[t * NX + hi];
                if (isHu) {
                    lan = lams[T * NX + (t + 1) * (2 * NU + nobs) + (hi - NX)];
                    lbn = lams[T * NX + (t + 1) * (2 * NU + nobs) + NU + (hi - NX)];
                }
            }
            // ---- dynamics residual r_t and multiplier estimate v = lam + rho r (W lanes)
            if (dyn && isW) {
                real xn = cv;
                if (!gxnext) {
#pragma unroll
                    for (int k = 0; k < N; ++k) xn = fma_(fa[k], zs[t * N + k], xn);
                }
                real r = zs[(t + 1) * N + wr] - xn;
                req[t * NX + wr] = r;
                vs[wr] = fma_(rho, r, la);
            }
            wave_sync();
            // ---- gradient entry and diagonal of H_tt (H lanes), al_utils.py:113-120
            real D = 0;
            real hob[3] = {0, 0, 0};   // obstacle rows: rho J_k'J_k entries (hi, 0..2) of the active rows
            if (isH) {
                real zv = zs[t * N + hi];
                real g = fma_(Qv, zv, qv);
                D = Qv;
                if (hi < NX) {
                    int row = (t == 0) ? (T - 1) * NX + hi : (t - 1) * NX + hi;
                    if (!(no_init && t == 0)) {
                        g += fma_(rho, req[row], la);
                        D += rho;
                    }
                } else {
                    int j = hi - NX;
                    real vu = zv - uhi(t, j), vl = -zv + ulo(t, j);
                    real au = vu >= 0 ? real(1) : real(0), al = vl >= 0 ? real(1) : real(0);
                    D = fma_(rho, au + al, D);
                    g += fma_(rho, vu > 0 ? vu : real(0), la) - fma_(rho, vl > 0 ? vl : real(0), lb);
                    if (no_init) g = 0;   // the given controls carry no gradient (al_utils_se.py:300-310)
                }
                if (dyn) {
                    real s = 0;
#pragma unroll
                    for (int r4 = 0; r4 < NXP; r4 += 4) {
                        real v4[4];
                        ld4(vs + r4, v4);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (r4 + i < NX) s = fma_(fa[r4 + i], v4[i], s);
                    }
                    g -= s;
                }
                if constexpr (NX >= 3) {
                    if (nobs > 0 && hi < 3) {
                        // J_k = -2 (p - o_k)' on x_t[0:3]: g += (lam_k + rho max(c_k, 0)) J_k', and for the
                        // rows with c_k >= 0 the Gauss-Newton term rho J_k'J_k (al_utils.py:373-386, 113-120)
                        const real *lk = lams + T * NX + t * (2 * NU + nobs) + 2 * NU;
                        for (int k = 0; k < nobs; ++k) {
                            const real *o = gobs + (size_t)(t * nobs + k) * 3;
                            const real d0 = zs[t * N] - o[0], d1 = zs[t * N + 1] - o[1], d2 = zs[t * N + 2] - o[2];
                            const real ck = obs_r2 - fma_(d0, d0, fma_(d1, d1, d2 * d2));
                            const real dh = hi == 0 ? d0 : (hi == 1 ? d1 : d2);
                            g = fma_(fma_(rho, ck > 0 ? ck : real(0), lk[k]), real(-2) * dh, g);
                            if (ck >= 0) {
                                const real w4 = real(4) * rho * dh;
                                hob[0] = fma_(w4, d0, hob[0]); hob[1] = fma_(w4, d1, hob[1]); hob[2] = fma_(w4, d2, hob[2]);
                            }
                        }
                    }
                }
                gs[hi] = g;
                if (g_out) g_out[t * N + hi] = g;
            }
            TSTAMP(0);
            // ---- SYRK phase: Hs = rho Ft Ft' - Sb Sb' on 2x2 blocks (lower triangle)
            if (isBlk) {
                real a00 = 0, a01 = 0, a10 = 0, a11 = 0;
                const int i0 = 2 * bi, j0 = 2 * bj;
                if (dyn) {
#pragma unroll
                    for (int c4 = 0; c4 < NXP; c4 += 4) {
                        real x0[4], x1[4], y0[4], y1[4];
                        ld4(Ft + i0 * NXP + c4, x0); ld4(Ft + (i0 + 1) * NXP + c4, x1);
                        ld4(Ft + j0 * NXP + c4, y0); ld4(Ft + (j0 + 1) * NXP + c4, y1);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            a00 = fma_(x0[i], y0[i], a00); a01 = fma_(x0[i], y1[i], a01);
                            a10 = fma_(x1[i], y0[i], a10); a11 = fma_(x1[i], y1[i], a11);
                        }
                    }
                    a00 *= rho; a01 *= rho; a10 *= rho; a11 *= rho;
                }
                if (t > 0) {
#pragma unroll
                    for (int c4 = 0; c4 < NP; c4 += 4) {
                        real x0[4], x1[4], y0[4], y1[4];
                        ld4(Sb + i0 * NP + c4, x0); ld4(Sb + (i0 + 1) * NP + c4, x1);
                        ld4(Sb + j0 * NP + c4, y0); ld4(Sb + (j0 + 1) * NP + c4, y1);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            a00 = fma_(-x0[i], y0[i], a00); a01 = fma_(-x0[i], y1[i], a01);
                            a10 = fma_(-x1[i], y0[i], a10); a11 = fma_(-x1[i], y1[i], a11);
                        }
                    }
                }
                Hs[i0 * HP + j0] = a00; Hs[i0 * HP + j0 + 1] = a01;
                Hs[(i0 + 1) * HP + j0] = a10; Hs[(i0 + 1) * HP + j0 + 1] = a11;
            }
            wave_sync();
            TSTAMP(1);
            // ---- diagonal of H_tt on top of the SYRK result (same lane order: in-order LDS)
            if (isH) Hs[hi * HP + hi] += D;
            if constexpr (NX >= 3) {
                if (nobs > 0 && isH && hi < 3) {
                    Hs[hi * HP + 0] += hob[0]; Hs[hi * HP + 1] += hob[1]; Hs[hi * HP + 2] += hob[2];
                }
            }
            wave_sync();
            // ---- initial row values of the panel: H rows / rhs row from LDS, W rows (-rho F_t)
            //      from registers, identity rows start at 0 and get their 1 at column ui
            {
                const int row = isH ? hi : N;
                const real gmul = isY ? real(1) : real(0), hmul = isU ? real(0) : real(1);
#pragma unroll
                for (int k4 = 0; k4 < NP; k4 += 4) {
                    real h4[4], g4[4];
                    ld4(Hs + row * HP + k4, h4);
                    ld4(gs + k4, g4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = k4 + i;
                        if (k < N) l[k] = isW ? -rho * fa[k] : fma_(-gmul, g4[i], hmul * h4[i]);
                    }
                }
            }
            // ---- panel phase (left-looking, operands broadcast from lane j's registers)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                // identity entry: 1 on lane (identity row j), 0 elsewhere, as arithmetic on a
                // per-lane float (a compare per j would be hoisted into 2 SGPRs each)
                real acc = l[j] + fmax_(real(0), real(1) - fabs_(ufl - real(j)));
                // row j of L first (j lane reads into j different scalar registers), the products after: hipcc otherwise
                // reuses ONE scalar register and emits v_readlane, s_nop 1 (the scalar-write -> vector-read hazard), v_fma
                // per term - twice the cycles of the 136 terms of a stage
                real lj[N > 1 ? N - 1 : 1];
#pragma unroll
                for (int k = 0; k < j; ++k) lj[k] = team_bcast<G>(l[k], j, team_base);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < j; ++k) acc = fma_(-l[k], lj[k], acc);
                real p = team_bcast<G>(acc, j, team_base);
                if (!(p > 0) && info == 0) info = t * N + j + 1;
                // no masking needed: H rows above the diagonal hold unused garbage, identity
                // rows stay exactly 0 left of their diagonal (X is upper triangular)
                l[j] = acc * rsqrt_(fabs_(p));  // |p|: modified Cholesky on a non-positive pivot (flagged in info)
            }
            TSTAMP(2);
            // ---- stage results
            if (isW || isY) {
                real *dst = isW ? Sb + wr * NP : Sb + N * NP;
#pragma unroll
                for (int k4 = 0; k4 < NP; k4 += 4)
                    st4(dst + k4, l[k4], k4 + 1 < N ? l[k4 + 1] : real(0), k4 + 2 < N ? l[k4 + 2] : real(0),
                        k4 + 3 < N ? l[k4 + 3] : real(0));
            }
            if (isY) {
#pragma unroll
                for (int k = 0; k < N; ++k) ds[t * N + k] = l[k];
            }
            if (isU) {
                real *Xr = Xp + (size_t)t * XT + (ui * N - (ui * (ui - 1)) / 2) - ui;
                // entries left of the diagonal are exact zeros: park them on the diagonal slot,
                // which the k == ui store then overwrites (stores of one lane stay in order)
#pragma unroll
                for (int k = 0; k < N; ++k) Xr[k > ui ? k : ui] = l[k];
            }
            wave_sync();
            TSTAMP(3);
        }
    }

    // Backward sweep: d_t = X_t ( y_t + rho X_t' F_t' dx_{t+1} ), and s = (J d)_eq.
    __device__ void backward_sweep() {
        real fbuf[FCH];
        if (T > 1) fetch_F(T - 2, fbuf);
        for (int t = T - 1; t >= 0; --t) {
            const bool dyn = t < T - 1;
            const real *Xt = Xp + (size_t)t * XT;
            real frow[N];
#pragma unroll
            for (int k = 0; k < N; ++k) frow[k] = 0;
            if (dyn) {
                stash_F(fbuf);
                wave_sync();
                if (t > 0) fetch_F(t - 1, fbuf);
                // own F column (H lanes) / own F row (W lanes)
                if (isH || isW) {
                    const int base = isW ? wr * N : hi, stride = isW ? 1 : N;
#pragma unroll
                    for (int k = 0; k < N; ++k)
                        if (isW || k < NX) frow[k] = Fs[base + k * stride];
                }
                // v = F_t' dx_{t+1}
                // (Every mat-vec of this sweep reads ALL its LDS operands first and multiplies afterwards, the two
                //  kept apart by a scheduling barrier: left to itself hipcc reuses two registers for the operands and
                //  emits read, wait, two FMAs, seven times per product - 68 exposed LDS round trips per stage.)
                if (isH) {
                    real dv[NX];
#pragma unroll
                    for (int r = 0; r < NX; ++r) dv[r] = ds[(t + 1) * N + r];
                    __builtin_amdgcn_sched_barrier(0);
                    real s = 0;
#pragma unroll
                    for (int r = 0; r < NX; ++r) s = fma_(frow[r], dv[r], s);
                    gs[hi] = s;
                }
                wave_sync();
            }
            // rhs_j = y_j + rho * sum_{i<=j} X[i][j] v_i
            if (isH) {
                real rhs = ds[t * N + hi];
                if (dyn) {
                    real xv[N], gv[N];
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        xv[i] = Xt[(i * N - (i * (i - 1)) / 2) + (hi - i)];
                        gv[i] = gs[i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    real s = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i) s = fma_(keep_if_nonneg(xv[i], hi - i), gv[i], s);
                    rhs = fma_(rho, s, rhs);
                }
                rs[hi] = rhs;
            }
            wave_sync();
            // d_i = sum_{j>=i} X[i][j] rhs_j
            if (isH) {
                const real *Xr = Xt + (hi * N - (hi * (hi - 1)) / 2) - hi;
                real xv[N], rv[N];
#pragma unroll
                for (int j = 0; j < N; ++j) { xv[j] = Xr[j]; rv[j] = rs[j]; }
                __builtin_amdgcn_sched_barrier(0);
                real s = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) s = fma_(keep_if_nonneg(xv[j], j - hi), rv[j], s);
                ds[t * N + hi] = s;
            }
            wave_sync();
            if (dyn && isW) {
                real dv[N];
#pragma unroll
                for (int k = 0; k < N; ++k) dv[k] = ds[t * N + k];
                const real dn = ds[(t + 1) * N + wr];
                __builtin_amdgcn_sched_barrier(0);
                real s = 0;
#pragma unroll
                for (int k = 0; k < N; ++k) s = fma_(frow[k], dv[k], s);
                seq[t * NX + wr] = dn - s;
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
                    const real *Xr = Xq + (hi * N - (hi * (hi - 1)) / 2) - hi;
                    real xv[N], dv[N];   // operands first, products after (see backward_sweep)
#pragma unroll
                    for (int j = 0; j < N; ++j) { xv[j] = Xr[j]; dv[j] = ds[(t - 1) * N + j]; }
                    __builtin_amdgcn_sched_barrier(0);
                    real s = 0;
#pragma unroll
                    for (int j = 0; j < N; ++j) s = fma_(keep_if_nonneg(xv[j], j - hi), dv[j], s);
                    gs[hi] = s;
                }
                wave_sync();
                if (isW) {
                    real fv[N], gv[N];
#pragma unroll
                    for (int k = 0; k < N; ++k) { fv[k] = Fs[wr * N + k]; gv[k] = gs[k]; }
                    const real d0 = ds[t * N + wr];
                    __builtin_amdgcn_sched_barrier(0);
                    real s = 0;
#pragma unroll
                    for (int k = 0; k < N; ++k) s = fma_(fv[k], gv[k], s);
                    ds[t * N + wr] = fma_(rho, s, d0);
                }
                wave_sync();
            }
            if (isH) rs[hi] = ds[t * N + hi];
            wave_sync();
            if (isH) {
                real xv[N], rv[N];
#pragma unroll
                for (int i = 0; i < N; ++i) { xv[i] = Xt[(i * N - (i * (i - 1)) / 2) + (hi - i)]; rv[i] = rs[i]; }
                __builtin_amdgcn_sched_barrier(0);
                real s = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) s = fma_(keep_if_nonneg(xv[i], hi - i), rv[i], s);
                ds[t * N + hi] = s;
            }
            wave_sync();
        }
    }

    // Merit of K candidates z + alpha_k d (alpha_k = 2^-k), al_utils.py:73-77.
    // Along the Newton direction the cost and the equality terms are exactly quadratic in
    // alpha (affine dynamics: r + alpha s), so they are accumulated once as three sums
    //   phi(alpha) = c0 + alpha c1 + alpha^2 c2 + bound terms(alpha)
    // and only the clamped bound terms are evaluated per candidate. Every lane returns
    // all K values. K = 1 with at_z evaluates the merit at z itself.
    template <int K>
    __device__ void merit_candidates(real (&phi)[K], bool at_z) {
        const int neq = T * NX;
        real c0 = 0, c1 = 0, c2 = 0;
        for (int e = li; e < T * N; e += G) {
            real z = zs[e], d = at_z ? real(0) : ds[e];
            real Qv = gQd[e], qv = gq[e];
            real gz = fma_(Qv, z, qv);
            c0 = fma_(fma_(real(0.5) * Qv, z, qv), z, c0);
            c1 = fma_(gz, d, c1);
            c2 = fma_(real(0.5) * Qv * d, d, c2);
        }
        for (int e = li; e < neq; e += G) {
            real r = req[e], sv = at_z ? real(0) : seq[e], lm = lams[e];
            c0 = fma_(fma_(real(0.5) * rho, r, lm), r, c0);
            c1 = fma_(fma_(rho, r, lm), sv, c1);
            c2 = fma_(real(0.5) * rho * sv, sv, c2);
        }
        real acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0;
        for (int e = li; e < T * NU; e += G) {
            int t = e / NU, j = e - t * NU;
            real z = zs[t * N + NX + j], d = at_z ? real(0) : ds[t * N + NX + j];
            int ru = neq + t * 2 * NU + j;
            real lu = lams[ru], ll = lams[ru + NU];
            real bu = uhi(t, j), bl = ulo(t, j);
            real alpha = 1;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                real zk = fma_(alpha, d, z);
                real vu = zk - bu, vl = bl - zk;
                real cu = fmax_(vu, real(0)), cl = fmax_(vl, real(0));
                acc[k] += fma_(lu, vu, ll * vl) + real(0.5) * rho * fma_(cu, cu, cl * cl);
                alpha *= real(0.5);
            }
        }
        c0 = team_sum<G>(c0);
        c1 = team_sum<G>(c1);
        c2 = team_sum<G>(c2);
        real alpha = 1;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            phi[k] = team_sum<G>(acc[k]) + fma_(alpha, fma_(alpha, c2, c1), c0);
            alpha *= real(0.5);
        }
    }

    // sum r+(z)^2 at the current zs/req (al_utils.py:552 sums this over the batch)
    __device__ real rplus2() {
        real acc = 0;
        for (int e = li; e < T * NX; e += G) acc = fma_(req[e], req[e], acc);
        for (int e = li; e < T * NU; e += G) {
            int t = e / NU, j = e - t * NU;
            real u = zs[t * N + NX + j];
            real vu = u - uhi(t, j), vl = -u + ulo(t, j);
            real cu = vu > 0 ? vu : real(0), cl = vl > 0 ? vl : real(0);
            acc += fma_(cu, cu, cl * cl);
        }
        return team_sum<G>(acc);
    }

    // lam <- lam + rho r ; lam_ineq <- max(0, .)   (AL_mpc.py:316-317)
    __device__ void dual_update() {
        const int neq = T * NX;
        for (int e = li; e < neq; e += G) lams[e] = fma_(rho, req[e], lams[e]);
        for (int e = li; e < T * NU; e += G) {
            int t = e / NU, j = e - t * NU;
            real u = zs[t * N + NX + j];
            int ru = neq + t * 2 * NU + j, rl = ru + NU;
            real a = fma_(rho, u - uhi(t, j), lams[ru]);
            real c = fma_(rho, -u + ulo(t, j), lams[rl]);
            lams[ru] = a < 0 ? real(0) : a;
            lams[rl] = c < 0 ? real(0) : c;
        }
        wave_sync();
    }
};

}  // namespace alqp
