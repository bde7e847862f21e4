// alqp_quad.hpp - "quad" variant of the fused LinDx solve: FOUR lanes per QP instance,
// 16 instances per 64-lane wavefront (gfx950 / MI355X only).
//
// Why a second variant. In the team variant (alqp_team.hpp) the whole factor lives in
// LDS: traffic stays at the algorithmic minimum, but 20 KB of LDS per instance caps a
// CU at 8 instances and a 64-lane wave works on a 17-wide problem (lane utilisation
// ~10 %). Here the n x n blocks are small enough that FOUR lanes hold a stage's panel
// entirely in registers (rows dealt cyclically, row i on lane i % 4) and every
// cross-lane operand is a DPP quad broadcast (a VALU operand modifier, no LDS, no
// v_readlane): ~6x fewer wave instructions per solve. The price: the per-stage factor
// and the step vectors are streamed through an HBM workspace, written in the forward
// sweep and read back in the backward sweep. MI355X's 8 TB/s HBM is otherwise idle on this
// problem, so spending bandwidth to buy lane utilisation is the right trade once the batch
// fills the chip (B >= 4096; below that the team variant has the lower latency).
//
// Workspace record of one (instance, stage) (QCfg): the unscaled root-free factor
// (Hh[k][j] = Lh[k][j] p_j, 1/p_j on the diagonal), y/d, r, s, and the stage's slice of
// z, lam, diag Q, q, c and the bounds: everything a sweep needs apart from F_t sits in a
// few whole 128-byte lines. stage_in() copies the slices in, iter_end() copies z, lam out.
// The n- and nx-vectors of a record are stored LANE-MAJOR (element k = 4m + q at word q*SY + m):
// the elements a lane owns are contiguous, so one 16-byte access (+ one word) moves them instead
// of one word per instruction - the CU's vector-memory pipeline, shared by its four wavefronts,
// is what this kernel queues on (profiles/r02/experiments/README.md).
//
// One Newton step = forward sweep (applies the pending line-search step, residual,
// gradient, H_tt, right-looking LDL' panel, Schur complement for the next stage) +
// backward sweep (substitutions, s_t = (J d)_t, and in fp32 the merits of the 20
// line-search candidates accumulated on the fly). One AL iteration ends with iter_end():
// apply, dual update, next starting merit, ||r+||^2, copy-out in ONE pass.
// With a dynamics model compiled in (alqp_dyn.hpp) the same sweeps run on linearisations
// made on the device (linearize) and the line search / dual update use the true dynamics
// (merit_nonlin, iter_end<Dyn>): the nonlinear MPC solve in one launch.
#pragma once
#include <hip/hip_runtime.h>

#include "alqp_team.hpp"  // fma_, rsqrt_, fmax_, fabs_, pad4
#include "alqp_dyn.hpp"   // inlinable dynamics models (nonlinear fused solve)

namespace alqp {

// ---- quad-level cross-lane primitives (DPP quad_perm) ------------------------------
template <int Q>
__device__ inline float qb(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), Q * 0x55, 0xF, 0xF, true));
}
template <int Q>
__device__ inline double qb(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), Q * 0x55, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), Q * 0x55, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// broadcast from quad lane `src`; src is a compile-time constant after unrolling, the switch
// folds away (the builtin needs a literal control word at each call site)
template <typename real>
__device__ inline real qbv(real v, int src) {
    switch (src & 3) {
        case 0: return qb<0>(v);
        case 1: return qb<1>(v);
        case 2: return qb<2>(v);
        default: return qb<3>(v);
    }
}
template <int CTRL>
__device__ inline float qperm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ inline double qperm(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// all-reduce over the 4 lanes of a quad
template <typename real>
__device__ inline real qsum(real v) {
    v += qperm<0xB1>(v);  // quad_perm [1,0,3,2]
    v += qperm<0x4E>(v);  // quad_perm [2,3,0,1]
    return v;
}
__device__ inline int qor(int v) {
    v |= __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
    v |= __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
    return v;
}
// element q of (a0,a1,a2,a3), q = lane within the quad
template <typename real>
__device__ inline real sel4(real a0, real a1, real a2, real a3, int q) {
    real lo = (q & 1) ? a1 : a0, hi = (q & 1) ? a3 : a2;
    return (q & 2) ? hi : lo;
}

// ---- 4-byte-aligned vector access to global memory ----------------------------------
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(8))) d4u { double x, y, z, w; };
__device__ inline void gld4(const float *p, float &a, float &b, float &c, float &d) {
    f4u t = *reinterpret_cast<const f4u *>(p);
    a = t.x; b = t.y; c = t.z; d = t.w;
}
__device__ inline void gld4(const double *p, double &a, double &b, double &c, double &d) {
    d4u t = *reinterpret_cast<const d4u *>(p);
    a = t.x; b = t.y; c = t.z; d = t.w;
}
__device__ inline void gst4(float *p, float a, float b, float c, float d) {
    f4u t; t.x = a; t.y = b; t.z = c; t.w = d;
    *reinterpret_cast<f4u *>(p) = t;
}
__device__ inline void gst4(double *p, double a, double b, double c, double d) {
    d4u t; t.x = a; t.y = b; t.z = c; t.w = d;
    *reinterpret_cast<d4u *>(p) = t;
}
// LEN contiguous reals into a register array (static indices only)
template <int LEN, typename real>
__device__ inline void gload(const real *p, real (&dst)[LEN]) {
#pragma unroll
    for (int c = 0; c + 4 <= LEN; c += 4) gld4(p + c, dst[c], dst[c + 1], dst[c + 2], dst[c + 3]);
#pragma unroll
    for (int c = (LEN / 4) * 4; c < LEN; ++c) dst[c] = p[c];
}

// ---- the nx rows of (-rho) F_t a lane works on during a stage: W[s][k], s < SW, k < N -------------
// fp32: registers. fp64 (ALQP_W_LDS): LDS, one 64-lane column per word ((s*N + k)*64 + lane: every
// access is a conflict-free ds_read/write_b64) - the quad kernels use no LDS otherwise, and in fp64
// these 2*SW*N dwords per lane are what pushed 872 B per lane into scratch (14 GB of HBM traffic per
// headline launch). 4 x 17 x 64 x 8 B = 34 KB per wavefront, four wavefronts per CU.
#ifndef ALQP_W_LDS
#define ALQP_W_LDS 1
#endif
// Byte-diet upper bounds (round 3, profiles/r03/experiments/README.md): builds with one of these set drop the named
// memory accesses WITHOUT replacing what they deliver - results are invalid, the launch time says what the real
// change could gain at most. Never set in the product build.
#ifndef ALQP_EXP_NOQ
#define ALQP_EXP_NOQ 0     // diag Q neither copied into the record nor read back (uniform-Q-in-SGPRs bound)
#endif
#ifndef ALQP_EXP_NORS
#define ALQP_EXP_NORS 0    // r and s neither stored nor re-read (recompute-instead-of-store bound)
#endif
#ifndef ALQP_EXP_NOL
#define ALQP_EXP_NOL 0     // factor of the last ALQP_EXP_NOL stages neither stored nor re-read (LDS turn-around bound)
#endif
template <typename real, int SW, int N, bool IN_LDS>
struct WPanel {
    real a[SW][N];
    __device__ __forceinline__ real (&operator[](int s))[N] { return a[s]; }
    __device__ __forceinline__ const real (&operator[](int s) const)[N] { return a[s]; }
};
template <typename real, int N>
struct WRowLds {
    real *p;
    __device__ __forceinline__ real &operator[](int k) const { return p[k * 64]; }
};
template <typename real, int SW, int N>
struct WPanel<real, SW, N, true> {
    real *base;  // LDS, already offset by the lane
    __device__ __forceinline__ WRowLds<real, N> operator[](int s) const { return WRowLds<real, N>{base + s * N * 64}; }
};
template <int LEN, typename real>
__device__ __forceinline__ void gload(const real *p, WRowLds<real, LEN> row) {
    real tmp[LEN];
    gload<LEN>(p, tmp);
#pragma unroll
    for (int k = 0; k < LEN; ++k) row[k] = tmp[k];
}

// "no dynamics model": the affine (LinDx) solve reads F and c from the caller's arrays
struct NoDyn {
    static constexpr int ID = 0;
};

template <typename real, int NX_, int NU_>
struct QCfg {
    static constexpr int NX = NX_, NU = NU_, N = NX_ + NU_;
    static constexpr int SH = (N + 3) / 4;   // slots of H rows per lane (row 4s+q)
    static constexpr int SW = (NX + 3) / 4;  // slots of W / F rows per lane
    static constexpr int SY = SH;            // own elements of an n-vector (k = 4m+q)
    static constexpr int HT = 2 * SH * (SH + 1);  // registers of the trimmed H panel: slot s has 4(s+1) columns
    __host__ __device__ static constexpr int hidx(int s, int j) { return 2 * s * (s + 1) + j; }
    static constexpr int ST = 2 * SW * (SW + 1);  // same trimming for the Schur accumulator
    // workspace record of one (instance, stage), in reals. L: slot s is s+1 chunks of 4 words
    // per lane, the lanes of a quad interleaved (64 contiguous bytes per chunk); in the last
    // slot only the NLAST lanes that own a real row store anything.
    static constexpr int NLAST = N - 4 * (SH - 1);                       // valid rows of the last slot (1..4)
    __host__ __device__ static constexpr int lanes_of(int s) { return s == SH - 1 ? NLAST : 4; }
    __host__ __device__ static constexpr int lbase(int s) {             // word offset of slot s
        int o = 0;
        for (int i = 0; i < s; ++i) o += (i + 1) * 4 * lanes_of(i);
        return o;
    }
    static constexpr int LW = lbase(SH);
    // fp32: the starting merit comes out of the first forward sweep; fp64 is short of registers
    // (it would spill 0.7 KB more) and keeps the residual pre-pass + a merit pass
    static constexpr bool PHI0_FWD = sizeof(real) == 4;   // (fp64, tried again with the W panel in LDS: 520 B of scratch come back)
#ifndef ALQP_S_AFTER_F32
#define ALQP_S_AFTER_F32 0   // experiment: 1 gives fp32 256 + 256 registers and 144 B of scratch (profiles/r02/experiments)
#endif
    static constexpr bool S_AFTER = sizeof(real) == 8 || ALQP_S_AFTER_F32;  // fp64: Schur accumulation after the panel (register pressure)
#ifndef ALQP_W_LDS_F32
#define ALQP_W_LDS_F32 0
#endif
    static constexpr bool W_LDS = ALQP_W_LDS && (sizeof(real) == 8 || ALQP_W_LDS_F32);   // the W rows live in LDS (WPanel)
    static constexpr int WLDS_WORDS = W_LDS ? SW * N * 64 : 1;          // reals of LDS per wavefront
    __host__ __device__ static constexpr int p4(int x) { return (x + 3) & ~3; }
    // The record also carries the stage's slice of every small per-stage array (working copies
    // of z and lam, copies of diag Q, q, c and the bounds): in their own arrays these are 52-68
    // byte pieces, each costing one or two 128-byte lines per touch; side by side in the record
    // the sweeps and the line search touch a few whole lines. stage_in()/stage_out() copy them.
    static constexpr int oL = 0;
    static constexpr int oY = oL + p4(LW);         // y_t, later d_t : element k at oY + k
    static constexpr int oR = oY + 4 * SY;         // r_t (eq residual of row block t): row r at oR + r
    static constexpr int oS = oR + 4 * SW;         // s_t = (J d)_eq
    static constexpr int oZ = oS + 4 * SW;         // z_t
    static constexpr int oLE = oZ + 4 * SY;        // lam, equality row block t
    // the bound rows of stage t, lane-major like the vectors: words 4q..4q+3 = (lam_upper, lam_lower, u_upper,
    // u_lower) of the ONE control lane q owns (element NX + ju with (NX + ju) % 4 == q; zeros when it owns none)
    static_assert(NU <= 4, "the u-slot layout holds one control per lane");
    static constexpr int oUS = oLE + 4 * SW;
    static constexpr int oQ = oUS + 16;            // diag Q_t
    static constexpr int oq = oQ + 4 * SY;         // q_t
    static constexpr int oC = oq + 4 * SY;         // c_t
    static constexpr int RECW = (oC + 4 * SW + 31) & ~31;  // whole 128-byte lines (fp32)
    // lane-major placement inside a vector field: element k of an n-vector (fields oY, oZ, oQ, oq; 4*SY words),
    // row r of an nx-vector (fields oR, oS, oLE, oC; 4*SW words). Padding words hold zeros.
    __host__ __device__ static constexpr int pn(int k) { return (k & 3) * SY + (k >> 2); }
    __host__ __device__ static constexpr int px(int r) { return (r & 3) * SW + (r >> 2); }
    __host__ __device__ static constexpr int M(int T) { return T * NX + 2 * T * NU; }
    // ONE mapping from (instance, stage) to the workspace: every kernel addresses its records through rec_base()
    // (+ t * RECW + word), and the size the caller allocates is checked on the host against the highest word that
    // mapping can touch (ws_covers) - so a future layout (e.g. blocks of 16 interleaved instances, where the padding
    // quads of a batch that is not a multiple of 16 have records of their own) cannot silently outgrow ws_words().
    __host__ __device__ static constexpr size_t rec_base(int b, int T) { return (size_t)b * T * RECW; }
    __host__ __device__ static constexpr size_t ws_words(int B, int T) { return (size_t)B * T * RECW; }
    // padding quads alias instance B - 1 (k_*_quad: `b = active ? b_raw : B - 1`), so b never exceeds B - 1
    __host__ static bool ws_covers(int B, int T) { return rec_base(B - 1, T) + (size_t)(T - 1) * RECW + RECW <= ws_words(B, T); }
};

template <typename real, int NX, int NU>
struct Quad {
    using C = QCfg<real, NX, NU>;
    static constexpr int N = C::N, SH = C::SH, SW = C::SW, SY = C::SY, HT = C::HT, ST = C::ST, RECW = C::RECW;

    int q, T;
    bool active;
    const real *gQd, *gq, *gF, *gc, *gx0, *gulo, *guhi;
    long st_u;
    real *gz, *glam, *rec;
    real rho;
    int info;
    real *wl;    // fp64: this lane's column of the W panel in LDS (WPanel), else unused
    using WT = WPanel<real, SW, N, C::W_LDS>;
    __device__ __forceinline__ WT wpanel() const {
        if constexpr (C::W_LDS) return WT{wl};
        else return WT{};
    }
    real *gFw;   // nonlinear fused solve: this instance's F_t linearisations (inside the workspace; gF points here too)
    real dyn_h;  //   step length of the dynamics model
    // caller-mode step kernel only (forward<EXT = true>): obstacle rows (Obstacle_MPC, al_utils.py:313-323, 351-388; as in
    // Team::forward_sweep) and the state-estimator variant's row set (al_utils_se.py:186-200, 300-310)
    const real *gobs;   // this instance's sphere centres [T][nobs][3] (global memory), nullptr without obstacles
    int nobs;
    real obs_r2;
    bool no_init;

    __device__ __forceinline__ real uhi(int t, int j) const { return guhi[t * st_u + j]; }
    __device__ __forceinline__ real ulo(int t, int j) const { return gulo[t * st_u + j]; }
    __device__ __forceinline__ real *recp(int t) const { return rec + (size_t)t * RECW; }
#ifdef ALQP_PHASE_TIMING
    // debug build only (tools/phase_timing.py): cycles per phase; a stamp drains the memory queue,
    // so "wait" buckets hold the exposed latency of the loads issued before them
    unsigned long long tacc[10], tlast;
    __device__ __forceinline__ void stamp(int bucket) {
        unsigned long long now;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (bucket >= 0) tacc[bucket] += now - tlast;
        tlast = now;
    }
#define ALQP_STAMP(b) stamp(b)
#else
#define ALQP_STAMP(b)
#endif

    // ---- lane-major vector fields of a record (QCfg::pn / px) ----
    // own elements: k = 4m + q (m < SY) of an n-vector, rows r = 4s + q (s < SW) of an nx-vector;
    // slots without an element read / write the field's zero padding
    __device__ __forceinline__ void ld_own_n(const real *field, real (&v)[SY]) const { gload<SY>(field + q * SY, v); }
    __device__ __forceinline__ void ld_own_x(const real *field, real (&v)[SW]) const { gload<SW>(field + q * SW, v); }
    // x-part rows 4s + q of an n-vector (rows >= NX: whatever element 4s+q holds, finite)
    __device__ __forceinline__ void ld_ownx_of_n(const real *field, real (&v)[SW]) const { gload<SW>(field + q * SY, v); }
    template <int LEN>
    __device__ __forceinline__ void gstore(real *p, const real (&v)[LEN]) const {
#pragma unroll
        for (int c = 0; c + 4 <= LEN; c += 4) gst4(p + c, v[c], v[c + 1], v[c + 2], v[c + 3]);
#pragma unroll
        for (int c = (LEN / 4) * 4; c < LEN; ++c) p[c] = v[c];
    }
    __device__ __forceinline__ void st_own_n(real *field, const real (&v)[SY]) const { gstore<SY>(field + q * SY, v); }
    __device__ __forceinline__ void st_own_x(real *field, const real (&v)[SW]) const { gstore<SW>(field + q * SW, v); }
    // every lane gets the whole vector (the 4*SY / 4*SW words are read by all four lanes alike)
    __device__ __forceinline__ void ld_rep_n(const real *field, real (&v)[N]) const {
        real w[4 * SY];
        gload<4 * SY>(field, w);
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = w[C::pn(k)];
    }
    __device__ __forceinline__ void ld_repx_of_n(const real *field, real (&v)[NX]) const {   // x part of an n-vector
        real w[4 * SY];
        gload<4 * SY>(field, w);
#pragma unroll
        for (int k = 0; k < NX; ++k) v[k] = w[C::pn(k)];
    }
    __device__ __forceinline__ void ld_rep_x(const real *field, real (&v)[NX]) const {
        real w[4 * SW];
        gload<4 * SW>(field, w);
#pragma unroll
        for (int r = 0; r < NX; ++r) v[r] = w[C::px(r)];
    }
    // own elements out of a vector every lane holds (k = 4m + q; 0 where there is none)
    template <int LEN, int SL>
    __device__ __forceinline__ void own_of(const real (&full)[LEN], real (&v)[SL]) const {
#pragma unroll
        for (int m = 0; m < SL; ++m)
            v[m] = (4 * m < LEN) ? sel4(full[(4 * m < LEN) ? 4 * m : 0], (4 * m + 1 < LEN) ? full[(4 * m + 1 < LEN) ? 4 * m + 1 : 0] : real(0),
                                        (4 * m + 2 < LEN) ? full[(4 * m + 2 < LEN) ? 4 * m + 2 : 0] : real(0),
                                        (4 * m + 3 < LEN) ? full[(4 * m + 3 < LEN) ? 4 * m + 3 : 0] : real(0), q)
                                 : real(0);
    }
    // caller's contiguous array -> own elements (0 where there is none) and back
    template <int LEN, int SL>
    __device__ __forceinline__ void ld_own_ext(const real *src, real (&v)[SL]) const {
#pragma unroll
        for (int m = 0; m < SL; ++m) {
            const int k = 4 * m + q;
            const real x = src[(4 * m + 3 < LEN || k < LEN) ? k : LEN - 1];
            v[m] = (4 * m + 3 < LEN || k < LEN) ? x : real(0);
        }
    }
    template <int LEN, int SL>
    __device__ __forceinline__ void st_own_ext(real *dst, const real (&v)[SL]) const {
#pragma unroll
        for (int m = 0; m < SL; ++m)
            if (4 * m + 3 < LEN || 4 * m + q < LEN) dst[4 * m + q] = v[m];
    }

    // ---- bound rows (QCfg::oUS): the control this lane owns is ju = (q - NX) mod 4, if that is < NU
    __device__ __forceinline__ int own_ju() const { return (q - NX) & 3; }
    __device__ __forceinline__ void ld_own_us(const real *rp, real &lu, real &ll, real &bu, real &bl) const {
        gld4(rp + C::oUS + 4 * q, lu, ll, bu, bl);
    }
    __device__ __forceinline__ void ld_rep_us(const real *rp, real (&lu)[NU], real (&ll)[NU], real (&bu)[NU], real (&bl)[NU]) const {
        real w[16];
        gload<16>(rp + C::oUS, w);
#pragma unroll
        for (int ju = 0; ju < NU; ++ju) {
            const int ql = (NX + ju) & 3;
            lu[ju] = w[4 * ql];
            ll[ju] = w[4 * ql + 1];
            bu[ju] = w[4 * ql + 2];
            bl[ju] = w[4 * ql + 3];
        }
    }

    // F_t rows of this lane: row 4s+q (zeros for rows >= NX). Loads are unconditional (row
    // index clamped, result masked) so that all of a stage's loads go out in one batch:
    // a load inside a divergent branch cannot be hoisted and costs its own round trip.
    __device__ __forceinline__ void load_F_rows(int t, WT &W) const {
        const real *Fg = gF + (size_t)t * NX * N;
#pragma unroll
        for (int s = 0; s < SW; ++s) {
            const int r = 4 * s + q;
            const int rc = (4 * s + 3 < NX) ? r : (r < NX ? r : NX - 1);
            gload<N>(Fg + rc * N, W[s]);
            if (4 * s + 3 >= NX) {
                const real m = r < NX ? real(1) : real(0);
#pragma unroll
                for (int k = 0; k < N; ++k) W[s][k] *= m;
            }
        }
    }
    // LEN contiguous words src -> dst, the quad's lanes taking 16-byte chunks in turn
    template <int LEN>
    __device__ __forceinline__ void copy_slice(const real *src, real *dst) const {
#pragma unroll
        for (int c = 0; 16 * c < LEN; ++c) {
            const int w0 = 16 * c + 4 * q;
            if (16 * c + 15 < LEN || w0 + 3 < LEN) {
                real a, b, cc, d;
                gld4(src + w0, a, b, cc, d);
                gst4(dst + w0, a, b, cc, d);
            } else {
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (w0 + w < LEN) dst[w0 + w] = src[w0 + w];
            }
        }
    }
    // Kernel start: the stage's slice of z, lam, diag Q, q, c and the bounds -> its record, and
    // (when the launch starts with a merit evaluation or is a dual update only) the equality
    // residuals at the current z; a forward sweep recomputes them anyway.
    __device__ __forceinline__ void stage_in(bool with_residual, bool have_c = true) {
#pragma unroll 2
        for (int t = 0; t < T; ++t) {
            const bool dyn = t < T - 1;
            real *rp = recp(t);
            if (active) {
                real vn[SY], vx[SW];
                ld_own_ext<N>(gz + t * N, vn);
                st_own_n(rp + C::oZ, vn);
                ld_own_ext<N>(gQd + t * N, vn);
                st_own_n(rp + C::oQ, vn);
                ld_own_ext<N>(gq + t * N, vn);
                st_own_n(rp + C::oq, vn);
                ld_own_ext<NX>(glam + t * NX, vx);
                st_own_x(rp + C::oLE, vx);
                {
                    const int ju = own_ju(), jc = ju < NU ? ju : 0;
                    const real *lb = glam + T * NX + t * 2 * NU;
                    const real a0 = lb[jc], a1 = lb[NU + jc], a2 = guhi[t * st_u + jc], a3 = gulo[t * st_u + jc];
                    const bool has = ju < NU;
                    gst4(rp + C::oUS + 4 * q, has ? a0 : real(0), has ? a1 : real(0), has ? a2 : real(0), has ? a3 : real(0));
                }
                if (dyn && have_c) ld_own_ext<NX>(gc + t * NX, vx);
                else {
#pragma unroll
                    for (int s = 0; s < SW; ++s) vx[s] = 0;
                }
                if (have_c) st_own_x(rp + C::oC, vx);
            }
            if (with_residual) {
                WT W = wpanel();
                real zt[N];
                load_F_rows(dyn ? t : (T > 1 ? T - 2 : 0), W);
                gload<N>(gz + t * N, zt);
                real rr[SW];
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    const int r = 4 * s + q;
                    rr[s] = 0;
                    if (4 * s + 3 < NX || r < NX) {
                        real xn = dyn ? gc[t * NX + r] : real(0);
#pragma unroll
                        for (int k = 0; k < N; ++k) xn = fma_(W[s][k], zt[k], xn);
                        // row block T-1 holds the initial-state rows x_0 - x_init (al_utils.py:274)
                        rr[s] = dyn ? gz[(t + 1) * N + r] - xn : gz[r] - gx0[r];
                    }
                }
                if (active) st_own_x(rp + C::oR, rr);
            }
        }
    }
    // Kernel end: the working copies of z and lam back to the caller's arrays.
    __device__ __forceinline__ void stage_out() {
        if (!active) return;
#pragma unroll 2
        for (int t = 0; t < T; ++t) {
            const real *rp = recp(t);
            real vn[SY], vx[SW];
            ld_own_n(rp + C::oZ, vn);
            st_own_ext<N>(gz + t * N, vn);
            ld_own_x(rp + C::oLE, vx);
            st_own_ext<NX>(glam + t * NX, vx);
            {
                real lu, ll, bu, bl;
                ld_own_us(rp, lu, ll, bu, bl);
                const int ju = own_ju();
                if (ju < NU) {
                    glam[T * NX + t * 2 * NU + ju] = lu;
                    glam[T * NX + t * 2 * NU + NU + ju] = ll;
                }
            }
        }
    }

    // ---- nonlinear fused solve (a dynamics model Dyn is inlined, alqp_dyn.hpp) -----------------
    // Linearisation pass of a Newton step (al_utils.py:233-248, dx_jac at the current iterate):
    // applies the pending line-search step, then per stage x+ = f(z_t) and J = df/dz by dual
    // numbers (the value in every lane, the n tangents split over the quad's four lanes),
    // F_t = J -> the workspace, c_t = f(z_t) - J z_t and r_t = x_{t+1} - f(z_t) -> the record. The
    // sweeps then run unchanged on (F_t, c_t).
    template <class Dyn>
    __device__ __forceinline__ void linearize(real alpha, bool pend) {
        static_assert(Dyn::NX == NX && Dyn::NU == NU, "model / kernel size mismatch");
        for (int t = 0; t < T; ++t) {
            const bool dyn = t < T - 1;
            real *rp = recp(t);
            const real *rn = recp(dyn ? t + 1 : t);
            real zt[N], dt[N], zn[SW], dn[SW];
            ld_rep_n(rp + C::oZ, zt);
            ld_rep_n(rp + C::oY, dt);
            ld_ownx_of_n(rn + C::oZ, zn);
            ld_ownx_of_n(rn + C::oY, dn);
            if (pend) {
#pragma unroll
                for (int k = 0; k < N; ++k) zt[k] = fma_(alpha, dt[k], zt[k]);
#pragma unroll
                for (int s = 0; s < SW; ++s) zn[s] = fma_(alpha, dn[s], zn[s]);
                if (active) {
                    real zo[SY];
                    own_of<N, SY>(zt, zo);
                    st_own_n(rp + C::oZ, zo);
                }
            }
            if (t == 0) {
                // initial-state rows x_0 - x_init live in row block T-1 (al_utils.py:274); a merit
                // evaluation may come before the first forward sweep rewrites them
                real r0[SW];
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    const int r = 4 * s + q;
                    const real z0r = sel4(zt[4 * s], (4 * s + 1 < NX) ? zt[(4 * s + 1 < NX) ? 4 * s + 1 : 0] : real(0),
                                          (4 * s + 2 < NX) ? zt[(4 * s + 2 < NX) ? 4 * s + 2 : 0] : real(0),
                                          (4 * s + 3 < NX) ? zt[(4 * s + 3 < NX) ? 4 * s + 3 : 0] : real(0), q);
                    r0[s] = (4 * s + 3 < NX || r < NX) ? z0r - gx0[(r < NX) ? r : NX - 1] : real(0);
                }
                if (active) st_own_x(recp(T - 1) + C::oR, r0);
            }
            if (dyn) {
                // the quad's lanes split the n tangents (lane q: columns 4i+q), then row 4s+q of J is
                // gathered: column k comes from lane k%4, the row index differs per destination lane
                constexpr int NTL = (N + 3) / 4;
                real xn[NX], Jl[NX][NTL];
                dyn_value_jac_split<Dyn, real>(zt, dyn_h, q, xn, Jl);
                real co[SW], ro[SW];
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    const int r = 4 * s + q;
                    real Jr[N];
#pragma unroll
                    for (int k = 0; k < N; ++k)
                        Jr[k] = sel4(qbv(Jl[4 * s][k >> 2], k),
                                     (4 * s + 1 < NX) ? qbv(Jl[(4 * s + 1 < NX) ? 4 * s + 1 : 0][k >> 2], k) : real(0),
                                     (4 * s + 2 < NX) ? qbv(Jl[(4 * s + 2 < NX) ? 4 * s + 2 : 0][k >> 2], k) : real(0),
                                     (4 * s + 3 < NX) ? qbv(Jl[(4 * s + 3 < NX) ? 4 * s + 3 : 0][k >> 2], k) : real(0), q);
                    const real xr = sel4(xn[4 * s], (4 * s + 1 < NX) ? xn[(4 * s + 1 < NX) ? 4 * s + 1 : 0] : real(0),
                                         (4 * s + 2 < NX) ? xn[(4 * s + 2 < NX) ? 4 * s + 2 : 0] : real(0),
                                         (4 * s + 3 < NX) ? xn[(4 * s + 3 < NX) ? 4 * s + 3 : 0] : real(0), q);
                    real c = xr;
#pragma unroll
                    for (int k = 0; k < N; ++k) c = fma_(-Jr[k], zt[k], c);
                    const bool okr = 4 * s + 3 < NX || r < NX;
                    if (okr && active) {
                        real *Fr = gFw + ((size_t)t * NX + r) * N;
#pragma unroll
                        for (int k = 0; k < N; ++k) Fr[k] = Jr[k];
                    }
                    co[s] = okr ? c : real(0);
                    ro[s] = okr ? zn[s] - xr : real(0);
                }
                if (active) {
                    st_own_x(rp + C::oC, co);
                    st_own_x(rp + C::oR, ro);
                }
            }
        }
    }
    // Merit of the 20 line-search candidates with the TRUE dynamics (al_utils.py:618-633): lane q of
    // the quad evaluates candidates k = 4i + q (i = 0..4) completely, then the values are exchanged.
    template <class Dyn>
    __device__ __forceinline__ void merit_nonlin(real (&phi)[20]) {
        real m[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) m[i] = 0;
        real x0v[NX], li[NX];
        gload<NX>(gx0, x0v);
        ld_rep_x(recp(T - 1) + C::oLE, li);
        const real a0 = sel4(real(1), real(0.5), real(0.25), real(0.125), q);
        for (int t = 0; t < T; ++t) {
            const bool dyn = t < T - 1;
            const real *rp = recp(t), *rn = recp(dyn ? t + 1 : t);
            real zt[N], dt[N], Qt[N], qt[N], zn[NX], dn[NX], le[NX], lu[NU], ll[NU], bu[NU], bl[NU];
            ld_rep_n(rp + C::oZ, zt);
            ld_rep_n(rp + C::oY, dt);
            ld_rep_n(rp + C::oQ, Qt);
            ld_rep_n(rp + C::oq, qt);
            ld_repx_of_n(rn + C::oZ, zn);
            ld_repx_of_n(rn + C::oY, dn);
            ld_rep_x(rp + C::oLE, le);
            ld_rep_us(rp, lu, ll, bu, bl);
            real alpha = a0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                real zc[N];
                real acc = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    zc[j] = fma_(alpha, dt[j], zt[j]);
                    acc = fma_(fma_(real(0.5) * Qt[j], zc[j], qt[j]), zc[j], acc);
                }
#pragma unroll
                for (int ju = 0; ju < NU; ++ju) {
                    const real vu = zc[NX + ju] - bu[ju], vl = bl[ju] - zc[NX + ju];
                    const real cu = fmax_(vu, real(0)), cl = fmax_(vl, real(0));
                    acc += fma_(lu[ju], vu, ll[ju] * vl) + real(0.5) * rho * fma_(cu, cu, cl * cl);
                }
                if (dyn) {
                    real xn[NX];
                    dyn_value<Dyn, real>(zc, dyn_h, xn);
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        const real rr = fma_(alpha, dn[r], zn[r]) - xn[r];
                        acc = fma_(fma_(real(0.5) * rho, rr, le[r]), rr, acc);
                    }
                }
                if (t == 0) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        const real rr = zc[r] - x0v[r];
                        acc = fma_(fma_(real(0.5) * rho, rr, li[r]), rr, acc);
                    }
                }
                m[i] += acc;
                alpha *= real(0.0625);
            }
        }
#pragma unroll
        for (int k = 0; k < 20; ++k) phi[k] = qbv(m[k >> 2], k);
    }

    // ---- forward sweep: gradient, factorisation, forward substitution ------------------
    // With `pending`, the step alpha*d the previous line search chose (d is still in the
    // records) is applied on the fly: z_t is read, advanced, used and written back here, which
    // saves the separate pass (this sweep recomputes the equality residuals anyway).
    // With `phi0` the sweep also returns the merit at the z it linearises at (everything the
    // merit needs is in registers here): the starting merit of an AL iteration costs no pass of
    // its own and the launch needs no residual pre-pass.
    // EXT (one Newton direction for a caller-side line search, k_newton_step_quad): no record was filled - the
    // stage's vectors come straight from the caller's arrays (own elements, one word per instruction) and the
    // residual is z_{t+1}[x] - xnext_t with the caller's xnext = f(z_t) (gxn); nothing but the factor and y is
    // written. Saves the copy-in pass, the pass that formed c_t = xnext - F z (a third read of F) and the re-read.
    template <bool EXT = false>
    __device__ __forceinline__ void forward(real *g_out, real alpha, bool pending, real *phi0, const real *gxn = nullptr) {
        real mdist = 0;  // merit terms, per-lane parts (own elements and rows; summed over the quad at the end)
        real S[ST], Sy[SW];
        // carried from stage to stage, own rows / elements only (row 4s+q of block t-1 pins element 4s+q of x_t):
        // vpo = lam + rho r of the previous row block, Syo = (W y) of the previous stage
        real vpo[SW], Syo[SW];
#pragma unroll
        for (int i = 0; i < ST; ++i) S[i] = 0;
#pragma unroll
        for (int s = 0; s < SW; ++s) Sy[s] = 0;
        // stage 0: x_0 is pinned by the initial-state rows (eq row block T-1), al_utils.py:274
        {
            real z0[SW], xi[SW], li[SW], r0[SW];
            if constexpr (EXT) {
                ld_own_ext<NX>(gz, z0);
                ld_own_ext<NX>(glam + (T - 1) * NX, li);
            } else {
                ld_ownx_of_n(recp(0) + C::oZ, z0);
                ld_own_x(recp(T - 1) + C::oLE, li);
            }
            ld_own_ext<NX>(gx0, xi);
            if (pending) {
                real d0[SW];
                ld_ownx_of_n(recp(0) + C::oY, d0);
#pragma unroll
                for (int s = 0; s < SW; ++s) z0[s] = fma_(alpha, d0[s], z0[s]);
            }
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                bool valid = 4 * s + 3 < NX || 4 * s + q < NX;
                if constexpr (EXT) valid = valid && !no_init;   // state estimator: no initial-state rows
                const real r = valid ? z0[s] - xi[s] : real(0);
                vpo[s] = valid ? fma_(rho, r, li[s]) : real(0);
                if constexpr (C::PHI0_FWD) mdist += valid ? fma_(fma_(real(0.5) * rho, r, li[s]), r, real(0)) : real(0);  // initial-state rows
                Syo[s] = 0;
                r0[s] = r;
            }
            if (!EXT && active) st_own_x(recp(T - 1) + C::oR, r0);
        }
        for (int t = 0; t < T; ++t) {
            const bool dyn = t < T - 1;
            WT W = wpanel();
            real Y[N];
            real dio[SY];   // diagonal of H_tt, own elements
            real v[SW];
            real zs[SY];
            real hob[3] = {0, 0, 0};   // obstacle rows (EXT): rho J_k'J_k entries (row q, columns 0..2) of the active rows
            real *rp = recp(t);
            // ---- loads (one batch) + residual + multiplier estimate. Everything per-stage is read as OWN
            // elements (one or two instructions per vector) and z_t is then broadcast inside the quad: loads
            // that return the same 80 bytes to all four lanes cost four times the L1 return bandwidth, which
            // is what the CU's four wavefronts queue on.
            {
                real zt[N];
                real Qo[SY], qo[SY];
                real cs[SW], zn[SW], lm[SW];
                real lu, ll, bu, bl;
                const int td = dyn ? t : (T > 1 ? T - 2 : 0);  // valid addresses for the last stage
                const real *rn = recp(td + 1);
                if constexpr (EXT) {
                    ld_own_ext<N>(gz + t * N, zs);
                    ld_own_ext<N>(gQd + t * N, Qo);
                    ld_own_ext<N>(gq + t * N, qo);
                    load_F_rows(td, W);
                    ld_own_ext<NX>(gxn + td * NX, cs);          // xnext_t = f(z_t): takes the place of F z + c
                    ld_own_ext<NX>(gz + (td + 1) * N, zn);
                    ld_own_ext<NX>(glam + td * NX, lm);
                    const int ju = own_ju(), jc = ju < NU ? ju : 0;
                    const real *lb = glam + T * NX + t * (2 * NU + nobs);   // obstacle rows sit behind the stage's bound rows
                    const bool has = ju < NU;
                    lu = has ? lb[jc] : real(0);
                    ll = has ? lb[NU + jc] : real(0);
                    bu = has ? guhi[t * st_u + jc] : real(0);
                    bl = has ? gulo[t * st_u + jc] : real(0);
                } else {
                    ld_own_n(rp + C::oZ, zs);
                    if constexpr (ALQP_EXP_NOQ) { for (int m = 0; m < SY; ++m) Qo[m] = real(1); }
                    else ld_own_n(rp + C::oQ, Qo);
                    ld_own_n(rp + C::oq, qo);
                    load_F_rows(td, W);
                    ld_own_x(rp + C::oC, cs);
                    ld_ownx_of_n(rn + C::oZ, zn);
                    ld_own_x(rp + C::oLE, lm);
                    ld_own_us(rp, lu, ll, bu, bl);
                }
                if (pending) {  // wave-uniform: z_t += alpha d_t, z_{t+1}[x] += alpha d_{t+1}[x]
                    real dt[SY], dn[SW];
                    ld_own_n(rp + C::oY, dt);
                    ld_ownx_of_n(rn + C::oY, dn);
#pragma unroll
                    for (int m = 0; m < SY; ++m) zs[m] = fma_(alpha, dt[m], zs[m]);   // written back with the stage results
#pragma unroll
                    for (int s = 0; s < SW; ++s) zn[s] = fma_(alpha, dn[s], zn[s]);
                }
                ALQP_STAMP(0);  // forward: exposed load latency
#pragma unroll
                for (int k = 0; k < N; ++k) zt[k] = qbv(zs[k >> 2], k);
                real rro[SW];
                if (!dyn) {
#pragma unroll
                    for (int s = 0; s < SW; ++s)
#pragma unroll
                        for (int k = 0; k < N; ++k) W[s][k] = 0;
                }
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    const int r = 4 * s + q;
                    real xn = cs[s];
                    if constexpr (!EXT) {
#pragma unroll
                        for (int k = 0; k < N; ++k) xn = fma_(W[s][k], zt[k], xn);
                    }
                    const real rr = zn[s] - xn;
                    const bool ok = dyn && r < NX;
                    v[s] = ok ? fma_(rho, rr, lm[s]) : real(0);
                    rro[s] = ok ? rr : real(0);
                    if constexpr (C::PHI0_FWD) mdist += ok ? fma_(fma_(real(0.5) * rho, rr, lm[s]), rr, real(0)) : real(0);
                }
                if (!ALQP_EXP_NORS && !EXT && dyn && active) st_own_x(rp + C::oR, rro);
                // ---- gradient and diagonal of H_tt, own elements (k = 4m + q)
                constexpr int MU0 = NX / 4;  // first element slot that can hold a control
                real zu = 0;                 // the control this lane owns (at most one: NU <= 4)
#pragma unroll
                for (int m = MU0; m < SY; ++m) zu = (4 * m + q >= NX && 4 * m + q < N) ? zs[m] : zu;
                const real vu = zu - bu, vl = bl - zu;
                const real au = vu >= 0 ? real(1) : real(0), al = vl >= 0 ? real(1) : real(0);
                const real cu = fmax_(vu, real(0)), cl = fmax_(vl, real(0));
                const real gu = fma_(rho, cu, lu) - fma_(rho, cl, ll);   // bound rows' part of the gradient on that control
                const real du = rho * (au + al);
                if constexpr (C::PHI0_FWD)
                    mdist += own_ju() < NU ? fma_(lu, vu, ll * vl) + real(0.5) * rho * fma_(cu, cu, cl * cl) : real(0);
                real go[SY];   // -(everything but the F'v term and the Schur carry), per own element
#pragma unroll
                for (int m = 0; m < SY; ++m) {
                    const int j = 4 * m + q;
                    const bool isx = (4 * m + 3 < NX) ? true : (4 * m >= NX ? false : j < NX);
                    const bool isu = (4 * m + 3 < NX) ? false : (j >= NX && j < N);
                    real g = fma_(Qo[m], zs[m], qo[m]);
                    real d = Qo[m];
                    if constexpr (C::PHI0_FWD) mdist = fma_(fma_(real(0.5) * Qo[m], zs[m], qo[m]), zs[m], mdist);   // padding: 0
                    g += isx ? vpo[m < SW ? m : 0] : (isu ? gu : real(0));
                    bool pinned = isx;
                    if constexpr (EXT) {
                        pinned = isx && !(no_init && t == 0);   // no E'E term on x_0 without initial-state rows
                        if (no_init && isu) g = 0;              // the given controls carry no gradient (al_utils_se.py:300-310)
                    }
                    d += pinned ? rho : (isu ? du : real(0));
                    dio[m] = d;
                    go[m] = g;
                }
                if constexpr (EXT && NX >= 3) {
                    if (nobs > 0) {
                        // J_k = -2 (p - o_k)' on x_t[0:3]: g += (lam_k + rho max(c_k, 0)) J_k', and for the rows with
                        // c_k >= 0 the Gauss-Newton term rho J_k'J_k on the position corner (al_utils.py:373-386, 113-120)
                        const real *lk = glam + T * NX + t * (2 * NU + nobs) + 2 * NU;
                        real gob = 0;
                        for (int k = 0; k < nobs; ++k) {
                            const real *o = gobs + (size_t)(t * nobs + k) * 3;
                            const real d0 = zt[0] - o[0], d1 = zt[1] - o[1], d2 = zt[2] - o[2];
                            const real ck = obs_r2 - fma_(d0, d0, fma_(d1, d1, d2 * d2));
                            const real dh = q == 0 ? d0 : (q == 1 ? d1 : d2);
                            gob = fma_(fma_(rho, ck > 0 ? ck : real(0), lk[k]), real(-2) * dh, gob);
                            if (ck >= 0) {
                                const real w4 = real(4) * rho * dh;
                                hob[0] = fma_(w4, d0, hob[0]); hob[1] = fma_(w4, d1, hob[1]); hob[2] = fma_(w4, d2, hob[2]);
                            }
                        }
                        go[0] += q < 3 ? gob : real(0);   // rows 0..2 of stage t are element slot 0 of lanes 0..2
                    }
                }
                // Y_j = -(g_j) - (W y)_{t-1,j}: broadcast of the own part plus the quad-reduced F'v term
#pragma unroll
                for (int m = 0; m < SY; ++m) {
                    const bool isx = (4 * m + 3 < NX) ? true : (4 * m >= NX ? false : 4 * m + q < NX);
                    go[m] = -go[m] - (isx ? Syo[m < SW ? m : 0] : real(0));
                }
                real fv[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    real p = 0;
                    if (dyn) {
#pragma unroll
                        for (int s = 0; s < SW; ++s) p = fma_(W[s][j], v[s], p);
                        p = qsum(p);
                    }
                    fv[j] = p;
                    Y[j] = qbv(go[j >> 2], j) + p;
                }
                if (g_out) {   // trace only: g = -(Y + carry)
                    real fo[SY];
                    own_of<N, SY>(fv, fo);
#pragma unroll
                    for (int m = 0; m < SY; ++m) {
                        const bool isx = (4 * m + 3 < NX) ? true : (4 * m >= NX ? false : 4 * m + q < NX);
                        if (4 * m + 3 < N || 4 * m + q < N)
                            g_out[t * N + 4 * m + q] = -(go[m] + fo[m]) - (isx ? Syo[m < SW ? m : 0] : real(0));
                    }
                }
            }
            // ---- H_tt rows (lower part, trimmed): diag + (1/rho) w_i w_j - Schur
            real H[HT];
#pragma unroll
            for (int i = 0; i < HT; ++i) H[i] = 0;
#pragma unroll
            for (int s = 0; s < SH; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * s + c < N) H[C::hidx(s, 4 * s + c)] = (q == c) ? dio[s] : real(0);
            if constexpr (EXT && NX >= 3) {
                if (nobs > 0 && q < 3) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) H[C::hidx(0, j)] += hob[j];
                }
            }
            // minus the Schur complement of stage t-1 (its registers are dead afterwards: keeps
            // the F'F phase below within the 256 architectural VGPRs)
            if (t > 0) {
#pragma unroll
                for (int s = 0; s < SW; ++s)
#pragma unroll
                    for (int b = 0; b < 4 * s + 4; ++b)
                        if (b < NX) H[C::hidx(s, b)] -= S[C::hidx(s, b)];
            }
#pragma unroll
            for (int i = 0; i < ST; ++i) S[i] = 0;
#pragma unroll
            for (int s = 0; s < SW; ++s) Sy[s] = 0;
            // W <- -rho F_t  (the F'F term is then (1/rho) W'W)
#pragma unroll
            for (int s = 0; s < SW; ++s)
#pragma unroll
                for (int k = 0; k < N; ++k) W[s][k] *= -rho;
            if (dyn) {
                const real irho = real(1) / rho;
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    real fr[N];
#pragma unroll
                    for (int j = 0; j < N; ++j) fr[j] = qbv(W[r >> 2][j], r);
#pragma unroll
                    for (int s = 0; s < SH; ++s) {
                        real fi = sel4(fr[4 * s], (4 * s + 1 < N) ? fr[(4 * s + 1 < N) ? 4 * s + 1 : 0] : real(0),
                                       (4 * s + 2 < N) ? fr[(4 * s + 2 < N) ? 4 * s + 2 : 0] : real(0),
                                       (4 * s + 3 < N) ? fr[(4 * s + 3 < N) ? 4 * s + 3 : 0] : real(0), q) * irho;
#pragma unroll
                        for (int j = 0; j < 4 * s + 4; ++j)
                            if (j < N) H[C::hidx(s, j)] = fma_(fi, fr[j], H[C::hidx(s, j)]);
                    }
                }
            }
            ALQP_STAMP(1);  // forward: residual, gradient, H assembly incl. F'F
            // ---- right-looking root-free panel factorisation: H = Lh D Lh' with D = diag(p_j) the
            // pivots and Lh unit lower. Column j is kept UNSCALED (Hh[k][j] = Lh[k][j] p_j) and
            // 1/p_j takes the pivot's place: no square root, no column scaling, and the solves
            // below use Hh[k][j] * (x_j / p_j).
            real ipv[C::S_AFTER ? N : 1];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const real p = qbv(H[C::hidx(j >> 2, j)], j);
                if (!(p > 0) && info == 0) info = t * N + j + 1;
                // |p| (a free source modifier): on a non-positive pivot - rounding at rho*cond beyond the
                // dtype, flagged in info - the panel becomes a modified Cholesky (factor of H + E, E >= 0)
                // and the step stays a descent direction for the line search to judge; the reference's
                // half-finished cholesky_ex factor (al_utils.py:510-515) has nothing to restate
                const real ip = rcp_(fabs_(p));
                if constexpr (C::W_LDS) {
                    // W lives in LDS (fp64): its columns are read KB at a time, ALL reads of a group first, then the
                    // updates, then the writes. Element by element (read, 2-3 instructions, use, write - what hipcc makes
                    // of the plain loop below) every one of the 544 read-modify-writes of a stage exposes most of an LDS
                    // round trip to the one wavefront of the SIMD.
                    constexpr int KB = 8;
                    real wj[SW];
#pragma unroll
                    for (int s = 0; s < SW; ++s) wj[s] = W[s][j];
#pragma unroll
                    for (int k0 = j + 1; k0 < N; k0 += KB) {
                        real wv[KB][SW];
#pragma unroll
                        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                            for (int s = 0; s < SW; ++s)
                                if (k0 + kk < N) wv[kk][s] = W[s][k0 + kk];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int kk = 0; kk < KB; ++kk) {
                            const int k = k0 + kk;
                            if (k < N) {
                                const real lkj = qbv(H[C::hidx(k >> 2, j)], k) * ip;
#pragma unroll
                                for (int s = (k >> 2); s < SH; ++s) H[C::hidx(s, k)] = fma_(-H[C::hidx(s, j)], lkj, H[C::hidx(s, k)]);
#pragma unroll
                                for (int s = 0; s < SW; ++s) W[s][k] = fma_(-wj[s], lkj, wv[kk][s]);
                                Y[k] = fma_(-Y[j], lkj, Y[k]);
                            }
                        }
                    }
                } else {
#pragma unroll
                for (int k = j + 1; k < N; ++k) {
                    const real lkj = qbv(H[C::hidx(k >> 2, j)], k) * ip;
#pragma unroll
                    for (int s = (k >> 2); s < SH; ++s) H[C::hidx(s, k)] = fma_(-H[C::hidx(s, j)], lkj, H[C::hidx(s, k)]);
#pragma unroll
                    for (int s = 0; s < SW; ++s) W[s][k] = fma_(-W[s][j], lkj, W[s][k]);
                    Y[k] = fma_(-Y[j], lkj, Y[k]);
                }
                }
                if constexpr (C::S_AFTER) ipv[j] = ip;
                if (!C::S_AFTER && dyn) {
#pragma unroll
                    for (int b = 0; b < NX; ++b) {
                        const real wbj = qbv(W[b >> 2][j], b) * ip;
#pragma unroll
                        for (int s = (b >> 2); s < SW; ++s) S[C::hidx(s, b)] = fma_(W[s][j], wbj, S[C::hidx(s, b)]);
                    }
                    const real yip = Y[j] * ip;
#pragma unroll
                    for (int s = 0; s < SW; ++s) Sy[s] = fma_(W[s][j], yip, Sy[s]);
                }
                H[C::hidx(j >> 2, j)] = (q == (j & 3)) ? ip : H[C::hidx(j >> 2, j)];
            }
            ALQP_STAMP(2);  // forward: panel
            // ---- stage results -> workspace (each lane its own words)
            if (active) {
                if (!(ALQP_EXP_NOL && t >= T - ALQP_EXP_NOL))
#pragma unroll
                for (int s = 0; s < SH; ++s)
#pragma unroll
                    for (int c = 0; c <= s; ++c)
                        if (C::lanes_of(s) == 4 || q < C::lanes_of(s))
                            gst4(rp + C::oL + C::lbase(s) + c * 4 * C::lanes_of(s) + 4 * q, H[C::hidx(s, 4 * c)],
                                 H[C::hidx(s, 4 * c + 1)], H[C::hidx(s, 4 * c + 2)], H[C::hidx(s, 4 * c + 3)]);
                real yo[SY];
                own_of<N, SY>(Y, yo);
                st_own_n(rp + C::oY, yo);
                if (pending) st_own_n(rp + C::oZ, zs);
            }

            if constexpr (C::S_AFTER) {
                // fp64: the Schur complement of this stage once the factor's registers are free
                if (dyn) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
#pragma unroll
                        for (int b = 0; b < NX; ++b) {
                            const real wbj = qbv(W[b >> 2][j], b) * ipv[j];
#pragma unroll
                            for (int s = (b >> 2); s < SW; ++s) S[C::hidx(s, b)] = fma_(W[s][j], wbj, S[C::hidx(s, b)]);
                        }
                        const real yip = Y[j] * ipv[j];
#pragma unroll
                        for (int s = 0; s < SW; ++s) Sy[s] = fma_(W[s][j], yip, Sy[s]);
                    }
                }
            }
            // ---- carry to the next stage (own rows): v = lam + rho r and W_t y_t
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                vpo[s] = v[s];
                Syo[s] = Sy[s];
            }
            ALQP_STAMP(3);  // forward: stores drained
        }
        if constexpr (C::PHI0_FWD) { if (phi0) *phi0 = qsum(mdist); }
    }

    // ---- backward sweep ---------------------------------------------------------------
    // LS: also accumulate the merit of the 20 line-search candidates z + 2^-k d while d_t and
    // s_t = (J d)_t are in registers (algebra: merit_candidates below; the sum runs over the
    // stages in sweep order T-1..0). The extra inputs sit in the record next to L and y.
    // d_ext (one Newton direction for the caller, k_newton_step_quad): d goes straight to the caller's array and
    // s = (J d)_eq, which only the fused line search reads, is not stored
    template <bool LS>
    __device__ __forceinline__ void backward(real (&phi)[20], real *d_ext = nullptr) {
        constexpr int MU0 = NX / 4;  // first element slot that can hold a control
        real c0 = 0, c1 = 0, c2 = 0;
        real acc[20];
        real rvT[SW], lvT[SW];
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = 0;
#pragma unroll
        for (int s = 0; s < SW; ++s) rvT[s] = lvT[s] = 0;
        real dxn[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) dxn[j] = 0;
        for (int t = T - 1; t >= 0; --t) {
            const bool dyn = t < T - 1;
            real *rp = recp(t);
            real H[HT];
            if (ALQP_EXP_NOL && t >= T - ALQP_EXP_NOL) {
#pragma unroll
                for (int i = 0; i < HT; ++i) H[i] = real(1);
            } else
#pragma unroll
            for (int s = 0; s < SH; ++s)
#pragma unroll
                for (int c = 0; c <= s; ++c) {
                    // lanes without a real row re-read lane 0's words (unconditional load, values unused)
                    const int ql = (C::lanes_of(s) == 4 || q < C::lanes_of(s)) ? q : 0;
                    gld4(rp + C::oL + C::lbase(s) + c * 4 * C::lanes_of(s) + 4 * ql, H[C::hidx(s, 4 * c)],
                         H[C::hidx(s, 4 * c + 1)], H[C::hidx(s, 4 * c + 2)], H[C::hidx(s, 4 * c + 3)]);
                }
            real yo[SY];
            ld_own_n(rp + C::oY, yo);
            real Y[N];
#pragma unroll
            for (int j = 0; j < N; ++j) Y[j] = qbv(yo[j >> 2], j);
            WT W = wpanel();
            load_F_rows(dyn ? t : (T > 1 ? T - 2 : 0), W);  // same batch as the record loads
            real zz[SY], QQ[SY], qq[SY], rv[SW], lv[SW];
            real lu = 0, ll = 0, bu = 0, bl = 0;
            if constexpr (LS) {
                ld_own_n(rp + C::oZ, zz);
                if constexpr (ALQP_EXP_NOQ) { for (int m = 0; m < SY; ++m) QQ[m] = real(1); }
                else ld_own_n(rp + C::oQ, QQ);
                ld_own_n(rp + C::oq, qq);
                ld_own_us(rp, lu, ll, bu, bl);
                if constexpr (ALQP_EXP_NORS) { for (int s = 0; s < SW; ++s) rv[s] = real(0); }
                else ld_own_x(rp + C::oR, rv);
                ld_own_x(rp + C::oLE, lv);
            }
            ALQP_STAMP(4);  // backward: exposed load latency
            real dxs[SW];
#pragma unroll
            for (int s = 0; s < SW; ++s)
                dxs[s] = sel4(dxn[4 * s], (4 * s + 1 < NX) ? dxn[(4 * s + 1 < NX) ? 4 * s + 1 : 0] : real(0),
                              (4 * s + 2 < NX) ? dxn[(4 * s + 2 < NX) ? 4 * s + 2 : 0] : real(0),
                              (4 * s + 3 < NX) ? dxn[(4 * s + 3 < NX) ? 4 * s + 3 : 0] : real(0), q);
            if (dyn) {
                // v = F_t' dx_{t+1}
                real vv[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    real p = 0;
#pragma unroll
                    for (int s = 0; s < SW; ++s) p = fma_(W[s][j], dxs[s], p);
                    vv[j] = qsum(p);
                }
                // u = Lh^{-1} v  (Lh[k][j] = Hh[k][j] / p_j)
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const real wj = vv[j] * qbv(H[C::hidx(j >> 2, j)], j);
#pragma unroll
                    for (int k = j + 1; k < N; ++k) vv[k] = fma_(-qbv(H[C::hidx(k >> 2, j)], k), wj, vv[k]);
                }
#pragma unroll
                for (int j = 0; j < N; ++j) Y[j] = fma_(rho, vv[j], Y[j]);
            }
            // d = Lh^{-T} D^{-1} rhs: d_i = (rhs_i - sum_{k>i} Hh[k][i] d_k) / p_i
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {
                const real di = Y[i] * qbv(H[C::hidx(i >> 2, i)], i);
                Y[i] = di;
#pragma unroll
                for (int j = 0; j < i; ++j) Y[j] = fma_(-qbv(H[C::hidx(i >> 2, j)], i), di, Y[j]);
            }
            if (active) {
                real down[SY];
                own_of<N, SY>(Y, down);
                if (d_ext) st_own_ext<N>(d_ext + t * N, down);
                else st_own_n(rp + C::oY, down);
            }
            real sv[SW];
#pragma unroll
            for (int s = 0; s < SW; ++s) sv[s] = 0;
            if (dyn) {
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    const int r = 4 * s + q;
                    real p = 0;
#pragma unroll
                    for (int k = 0; k < N; ++k) p = fma_(W[s][k], Y[k], p);
                    sv[s] = (4 * s + 3 < NX || r < NX) ? dxs[s] - p : real(0);
                }
                if (!ALQP_EXP_NORS && active && !d_ext) st_own_x(rp + C::oS, sv);
            }
            if constexpr (LS) {
                real zu = 0, du = 0;   // the control this lane owns (at most one: NU <= 4)
#pragma unroll
                for (int m = 0; m < SY; ++m) {
                    const int j = 4 * m + q;
                    const real ok = (4 * m + 3 < N || j < N) ? real(1) : real(0);
                    const real dj = sel4(Y[4 * m], (4 * m + 1 < N) ? Y[(4 * m + 1 < N) ? 4 * m + 1 : 0] : real(0),
                                         (4 * m + 2 < N) ? Y[(4 * m + 2 < N) ? 4 * m + 2 : 0] : real(0),
                                         (4 * m + 3 < N) ? Y[(4 * m + 3 < N) ? 4 * m + 3 : 0] : real(0), q);
                    const real z = zz[m], d = dj * ok, Qv = QQ[m] * ok, qv = qq[m] * ok;
                    c0 = fma_(fma_(real(0.5) * Qv, z, qv), z, c0);
                    c1 = fma_(fma_(Qv, z, qv), d, c1);
                    c2 = fma_(real(0.5) * Qv * d, d, c2);
                    if (m >= MU0) {
                        const bool mine = j >= NX && j < N;
                        zu = mine ? z : zu;
                        du = mine ? d : du;
                    }
                }
                {
                    const real isu = own_ju() < NU ? real(1) : real(0);
                    real alpha = 1;
#pragma unroll
                    for (int k = 0; k < 20; ++k) {
                        real zk = fma_(alpha, du, zu);
                        real vu = zk - bu, vl = bl - zk;
                        real cu = fmax_(vu, real(0)), cl = fmax_(vl, real(0));
                        acc[k] = fma_(isu, fma_(lu, vu, ll * vl) + real(0.5) * rho * fma_(cu, cu, cl * cl), acc[k]);
                        alpha *= real(0.5);
                    }
                }
                if (dyn) {
#pragma unroll
                    for (int s = 0; s < SW; ++s) {
                        const int r = 4 * s + q;
                        const real ok = (4 * s + 3 < NX || r < NX) ? real(1) : real(0);
                        const real rr = rv[s] * ok, ss = sv[s] * ok, lm = lv[s] * ok;
                        c0 = fma_(fma_(real(0.5) * rho, rr, lm), rr, c0);
                        c1 = fma_(fma_(rho, rr, lm), ss, c1);
                        c2 = fma_(real(0.5) * rho * ss, ss, c2);
                    }
                } else {
                    // row block T-1 holds the initial-state rows; their s = d_0[x] comes last
#pragma unroll
                    for (int s = 0; s < SW; ++s) {
                        rvT[s] = rv[s];
                        lvT[s] = lv[s];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) dxn[j] = Y[j];
            ALQP_STAMP(5);  // backward: solves + stores
        }
        // initial-state rows: s = d_0[x]
        if (active && !d_ext) {
            real so[SW];
            own_of<NX, SW>(dxn, so);
            st_own_x(recp(T - 1) + C::oS, so);
        }
        if constexpr (LS) {
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q;
                const real ok = (4 * s + 3 < NX || r < NX) ? real(1) : real(0);
                const real sj = sel4(dxn[4 * s], (4 * s + 1 < NX) ? dxn[(4 * s + 1 < NX) ? 4 * s + 1 : 0] : real(0),
                                     (4 * s + 2 < NX) ? dxn[(4 * s + 2 < NX) ? 4 * s + 2 : 0] : real(0),
                                     (4 * s + 3 < NX) ? dxn[(4 * s + 3 < NX) ? 4 * s + 3 : 0] : real(0), q);
                const real rr = rvT[s] * ok, ss = sj * ok, lm = lvT[s] * ok;
                c0 = fma_(fma_(real(0.5) * rho, rr, lm), rr, c0);
                c1 = fma_(fma_(rho, rr, lm), ss, c1);
                c2 = fma_(real(0.5) * rho * ss, ss, c2);
            }
            c0 = qsum(c0);
            c1 = qsum(c1);
            c2 = qsum(c2);
            real alpha = 1;
#pragma unroll
            for (int k = 0; k < 20; ++k) {
                phi[k] = qsum(acc[k]) + fma_(alpha, fma_(alpha, c2, c1), c0);
                alpha *= real(0.5);
            }
        }
    }

    // ---- forward substitution only, with the factor of a previous solve still in the workspace:
    //      y_t = L_t^{-1} ( rhs_t + rho F_{t-1} L_{t-1}^{-T} y_{t-1} [x rows] ), rhs_t = -gbar_t.
    //      Used by the backward pass of the implicit layer (NewtonAL.backward, al_utils.py:578-615);
    //      leaves y_t in the record's y slot, where backward() picks it up.
    __device__ __forceinline__ void solve_forward(const real *gbar) {
        real e[N];
#pragma unroll
        for (int j = 0; j < N; ++j) e[j] = 0;
        for (int t = 0; t < T; ++t) {
            real *rp = recp(t);
            real H[HT];
#pragma unroll
            for (int s = 0; s < SH; ++s)
#pragma unroll
                for (int c = 0; c <= s; ++c) {
                    const int ql = (C::lanes_of(s) == 4 || q < C::lanes_of(s)) ? q : 0;
                    gld4(rp + C::oL + C::lbase(s) + c * 4 * C::lanes_of(s) + 4 * ql, H[C::hidx(s, 4 * c)],
                         H[C::hidx(s, 4 * c + 1)], H[C::hidx(s, 4 * c + 2)], H[C::hidx(s, 4 * c + 3)]);
                }
            real Y[N];
            gload<N>(gbar + t * N, Y);
            WT W = wpanel();
            load_F_rows(t > 0 ? t - 1 : 0, W);
#pragma unroll
            for (int j = 0; j < N; ++j) Y[j] = -Y[j];
            if (t > 0) {
                real cpl[SW];
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    real p = 0;
#pragma unroll
                    for (int k = 0; k < N; ++k) p = fma_(W[s][k], e[k], p);
                    cpl[s] = p;
                }
#pragma unroll
                for (int j = 0; j < NX; ++j) Y[j] = fma_(rho, qbv(cpl[j >> 2], j), Y[j]);
            }
            // u = Lh^{-1} rhs
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const real yj = Y[j] * qbv(H[C::hidx(j >> 2, j)], j);
#pragma unroll
                for (int k = j + 1; k < N; ++k) Y[k] = fma_(-qbv(H[C::hidx(k >> 2, j)], k), yj, Y[k]);
            }
            if (active) {
                real yo[SY];
                own_of<N, SY>(Y, yo);
                st_own_n(rp + C::oY, yo);
            }
            // e = Lh^{-T} D^{-1} u for the coupling of the next stage
#pragma unroll
            for (int j = 0; j < N; ++j) e[j] = Y[j];
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {
                const real ei = e[i] * qbv(H[C::hidx(i >> 2, i)], i);
                e[i] = ei;
#pragma unroll
                for (int j = 0; j < i; ++j) e[j] = fma_(-qbv(H[C::hidx(i >> 2, j)], i), ei, e[j]);
            }
        }
    }

    // ---- merit of K candidates (see Team::merit_candidates for the algebra) ---------------
    // All loads of a stage are unconditional (clamped indices, masked contributions) so they
    // leave in one batch.
    template <int K>
    __device__ __forceinline__ void merit_candidates(real (&phi)[K], bool at_z) {
        real c0 = 0, c1 = 0, c2 = 0;
        real acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0;
        constexpr int MU0 = NX / 4;  // first element slot that can hold a control
#pragma unroll 4
        for (int t = 0; t < T; ++t) {  // 4 stages of loads in flight per round trip
            const real *rp = recp(t);
            real zz[SY], dd[SY], QQ[SY], qq[SY];
            real rv[SW], sv[SW], lv[SW];
            real lu, ll, bu, bl;
            ld_own_n(rp + C::oZ, zz);
            ld_own_n(rp + C::oY, dd);
            if constexpr (ALQP_EXP_NOQ) { for (int m = 0; m < SY; ++m) QQ[m] = real(1); }
            else ld_own_n(rp + C::oQ, QQ);
            ld_own_n(rp + C::oq, qq);
            ld_own_us(rp, lu, ll, bu, bl);
            if constexpr (ALQP_EXP_NORS) { for (int s = 0; s < SW; ++s) rv[s] = sv[s] = real(0); }
            else { ld_own_x(rp + C::oR, rv); ld_own_x(rp + C::oS, sv); }
            ld_own_x(rp + C::oLE, lv);
            real zu = 0, du = 0;   // the control this lane owns (at most one: NU <= 4)
#pragma unroll
            for (int m = 0; m < SY; ++m) {
                const int j = 4 * m + q;
                const real ok = (4 * m + 3 < N || j < N) ? real(1) : real(0);
                const real z = zz[m], d = at_z ? real(0) : dd[m] * ok, Qv = QQ[m] * ok, qv = qq[m] * ok;
                c0 = fma_(fma_(real(0.5) * Qv, z, qv), z, c0);
                c1 = fma_(fma_(Qv, z, qv), d, c1);
                c2 = fma_(real(0.5) * Qv * d, d, c2);
                if (m >= MU0) {
                    const bool mine = j >= NX && j < N;
                    zu = mine ? z : zu;
                    du = mine ? d : du;
                }
            }
            {
                const real isu = own_ju() < NU ? real(1) : real(0);
                real alpha = 1;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    real zk = fma_(alpha, du, zu);
                    real vu = zk - bu, vl = bl - zk;
                    real cu = fmax_(vu, real(0)), cl = fmax_(vl, real(0));
                    acc[k] = fma_(isu, fma_(lu, vu, ll * vl) + real(0.5) * rho * fma_(cu, cu, cl * cl), acc[k]);
                    alpha *= real(0.5);
                }
            }
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q;
                const real ok = (4 * s + 3 < NX || r < NX) ? real(1) : real(0);
                const real rr = rv[s] * ok, ss = at_z ? real(0) : sv[s] * ok, lm = lv[s] * ok;
                c0 = fma_(fma_(real(0.5) * rho, rr, lm), rr, c0);
                c1 = fma_(fma_(rho, rr, lm), ss, c1);
                c2 = fma_(real(0.5) * rho * ss, ss, c2);
            }
        }
        c0 = qsum(c0);
        c1 = qsum(c1);
        c2 = qsum(c2);
        real alpha = 1;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            phi[k] = qsum(acc[k]) + fma_(alpha, fma_(alpha, c2, c1), c0);
            alpha *= real(0.5);
        }
    }

    // End of an AL iteration in ONE pass over the records (own elements only): apply the pending
    // step (z += alpha d, r += alpha s), the dual update lam <- lam + rho r, bound rows clamped
    // at 0, rho <- rho * scale (AL_mpc.py:315-317,325), the merit at the new (z, lam, rho) = the
    // next iteration's starting merit (same expression order as merit_candidates<1>(., true)),
    // ||r_+||^2 and the finiteness flag (al_utils.py:545-549); on the last iteration also the
    // copy-out of z and lam.
    // With a dynamics model (Dyn::ID != 0) the residuals are re-evaluated with the TRUE dynamics at
    // the new z (AL_mpc.py:315: the dual update never uses the linearisation).
    template <class Dyn = NoDyn>
    __device__ __forceinline__ void iter_end(real alpha, bool pend, bool dual, real rho_scale, bool write_out,
                                             real &phi_next, real &rn2, int &bad) {
        constexpr int MU0 = NX / 4;
        const real rho_n = dual ? rho * rho_scale : rho;
        if (!pend) alpha = 0;
        real c0 = 0, acc0 = 0, r2 = 0;
#pragma unroll 2
        for (int t = 0; t < T; ++t) {
            real *rp = recp(t);
            real zz[SY], dd[SY], QQ[SY], qq[SY];
            real rv[SW], sv[SW], lv[SW];
            real lu, ll, bu, bl;
            ld_own_n(rp + C::oZ, zz);
            ld_own_n(rp + C::oY, dd);
            if constexpr (ALQP_EXP_NOQ) { for (int m = 0; m < SY; ++m) QQ[m] = real(1); }
            else ld_own_n(rp + C::oQ, QQ);
            ld_own_n(rp + C::oq, qq);
            ld_own_us(rp, lu, ll, bu, bl);
            if constexpr (ALQP_EXP_NORS) { for (int s = 0; s < SW; ++s) rv[s] = sv[s] = real(0); }
            else { ld_own_x(rp + C::oR, rv); ld_own_x(rp + C::oS, sv); }
            ld_own_x(rp + C::oLE, lv);
            real rtrue[SW];
#pragma unroll
            for (int s = 0; s < SW; ++s) rtrue[s] = 0;
            if constexpr (Dyn::ID != 0) {
                if (t < T - 1) {
                    const real *rn_ = recp(t + 1);
                    real zr[N], dr[N], xn[NX];
                    ld_rep_n(rp + C::oZ, zr);
                    ld_rep_n(rp + C::oY, dr);
                    if (pend) {
#pragma unroll
                        for (int k = 0; k < N; ++k) zr[k] = fma_(alpha, dr[k], zr[k]);
                    }
                    dyn_value<Dyn, real>(zr, dyn_h, xn);
                    real zno[SW], dno[SW];
                    ld_ownx_of_n(rn_ + C::oZ, zno);
                    ld_ownx_of_n(rn_ + C::oY, dno);
#pragma unroll
                    for (int s = 0; s < SW; ++s) {
                        real znn = zno[s];
                        const real dnn = dno[s];
                        if (pend) znn = fma_(alpha, dnn, znn);
                        const real xr = sel4(xn[4 * s], (4 * s + 1 < NX) ? xn[(4 * s + 1 < NX) ? 4 * s + 1 : 0] : real(0),
                                             (4 * s + 2 < NX) ? xn[(4 * s + 2 < NX) ? 4 * s + 2 : 0] : real(0),
                                             (4 * s + 3 < NX) ? xn[(4 * s + 3 < NX) ? 4 * s + 3 : 0] : real(0), q);
                        rtrue[s] = znn - xr;
                    }
                }
            }
            real zu = 0;   // the control this lane owns (at most one: NU <= 4)
#pragma unroll
            for (int m = 0; m < SY; ++m) {
                const int j = 4 * m + q;
                const bool valid = 4 * m + 3 < N || j < N;
                const real ok = valid ? real(1) : real(0);
                const real z = valid ? (pend ? fma_(alpha, dd[m], zz[m]) : zz[m]) : real(0);
                const real Qv = QQ[m] * ok, qv = qq[m] * ok;
                if (valid) bad |= !(z - z == real(0));
                zz[m] = z;
                if (valid && active) {
                    if (write_out) gz[t * N + j] = z;
                }
                c0 = fma_(fma_(real(0.5) * Qv, z, qv), z, c0);
                if (m >= MU0) zu = (j >= NX && j < N) ? z : zu;
            }
            {
                const int ju = own_ju();
                const bool isub = ju < NU;
                const real isu = isub ? real(1) : real(0);
                const real vu = zu - bu, vl = bl - zu;
                real lun = lu, lln = ll;
                if (dual) {
                    const real a = fma_(rho, vu, lu), c = fma_(rho, vl, ll);
                    lun = a < 0 ? real(0) : a;
                    lln = c < 0 ? real(0) : c;
                }
                if (isub && active) {
                    if (dual) gst4(rp + C::oUS + 4 * q, lun, lln, bu, bl);
                    if (write_out) {
                        glam[T * NX + t * 2 * NU + ju] = lun;
                        glam[T * NX + t * 2 * NU + NU + ju] = lln;
                    }
                }
                const real cu = fmax_(vu, real(0)), cl = fmax_(vl, real(0));
                acc0 = fma_(isu, fma_(lun, vu, lln * vl) + real(0.5) * rho_n * fma_(cu, cu, cl * cl), acc0);
                r2 = fma_(isu, fma_(cu, cu, cl * cl), r2);
            }
            if (pend && active) st_own_n(rp + C::oZ, zz);
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q;
                const bool valid = 4 * s + 3 < NX || r < NX;
                const real ok = valid ? real(1) : real(0);
                real rn = pend ? fma_(alpha, sv[s], rv[s]) : rv[s];
                if constexpr (Dyn::ID != 0) {
                    if (t < T - 1) rn = rtrue[s];
                }
                if (!valid) rn = 0;
                const real ln = valid ? (dual ? fma_(rho, rn, lv[s]) : lv[s]) : real(0);
                rv[s] = rn;
                lv[s] = ln;
                if (valid && active) {
                    if (write_out) glam[t * NX + r] = ln;
                }
                const real rr = rn * ok, lm = ln * ok;
                c0 = fma_(fma_(real(0.5) * rho_n, rr, lm), rr, c0);
                r2 = fma_(rr, rr, r2);
            }
            if (active) {
                if (pend || Dyn::ID != 0) st_own_x(rp + C::oR, rv);
                if (dual) st_own_x(rp + C::oLE, lv);
            }
        }
        rho = rho_n;
        c0 = qsum(c0);
        phi_next = qsum(acc0) + c0;
        rn2 = qsum(r2);
        bad = qor(bad);
    }

    __device__ __forceinline__ real rplus2(int &bad) {
        real acc = 0;
        for (int t = 0; t < T; ++t) {
            const real *rp = recp(t);
            real ro[SW], zo[SY];
            real ulu, ull, ubu, ubl;
            ld_own_x(rp + C::oR, ro);
            ld_own_n(rp + C::oZ, zo);
            ld_own_us(rp, ulu, ull, ubu, ubl);
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q;
                if (r < NX) { real rr = ro[s]; acc = fma_(rr, rr, acc); }
            }
#pragma unroll
            for (int m = 0; m < SY; ++m) {
                const int j = 4 * m + q;
                if (j < N) {
                    real z = zo[m];
                    bad |= !(z - z == real(0));
                    if (j >= NX) {
                        real cu = fmax_(z - ubu, real(0)), cl = fmax_(ubl - z, real(0));
                        acc += fma_(cu, cu, cl * cl);
                    }
                }
            }
        }
        bad = qor(bad);
        return qsum(acc);
    }
};

}  // namespace alqp
