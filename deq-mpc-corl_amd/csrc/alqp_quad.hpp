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
// (lower triangle of L_tt, ~1 KB) and the vectors y/d, r, s are streamed through an
// HBM workspace, written in the forward sweep and read back in the backward sweep /
// line search. MI355X's 8 TB/s HBM is otherwise idle on this problem (the team variant
// moves 0.05 TB/s), so spending bandwidth to buy lane utilisation is the right trade.
//
// Workspace discipline: every word of the workspace is only ever read by the lane that
// wrote it (same-thread RAW through global memory needs no fence).
//
// Forward stage t (right-looking panel factorisation, all loops fully unrolled):
//   rows of H_tt (slot s, lane q: row 4s+q), rows of -rho F_t, the replicated rhs row;
//   pivot j: broadcast pivot, v_rsq, scale column j, rank-1 update of the trailing
//   columns (one DPP broadcast of L[k][j] per (j,k) pair), and the Schur complement
//   W_t W_t' / W_t y_t for stage t+1 accumulated in the same pass.
// Backward stage t: d_t = L^{-T} ( y_t + rho L^{-1} F_t' dx_{t+1} ) by substitution on
//   replicated vectors, s_t = dx_{t+1} - F_t d_t.
#pragma once
#include <hip/hip_runtime.h>

#include "alqp_team.hpp"  // fma_, rsqrt_, fmax_, fabs_, pad4

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

// ---- global -> LDS DMA (gfx950 global_load_lds_dword / _dwordx4) ----------------------
// One wave-instruction moves SZ bytes per enabled lane from the lane's global address to LDS byte
// M0 + IMM + SZ*lane, with no register destination (tools/probes/glds_probe.hip and
// glds_x4_probe.hip pin the semantics used here: per-lane source that only needs 4-byte
// alignment, lane-linear destination, the immediate offset applied to BOTH addresses, the
// SGPR-base + VGPR-offset form, lanes switched off by EXEC write nothing). M0 is
// compiler-reserved: it is saved, written and restored inside the one statement. hipcc does not
// count an asm load in its s_waitcnt bookkeeping: the sweeps wait with dma_wait<>().
// What a DMA instruction costs is its pass through the CU's address unit (about one clock per
// 128-byte line touched, shared by the CU's four waves), so the images below are laid out for
// few, wide instructions.
template <int M0ADD, int IMM>
__device__ __forceinline__ void glds_dword(const void *gsrc, unsigned lds_base) {
    static_assert(IMM >= 0 && IMM < 4096 && M0ADD >= 0, "13-bit signed immediate");
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, off offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base), "i"(M0ADD), "i"(IMM)
                 : "memory", "scc");
}
template <int M0ADD, int IMM>
__device__ __forceinline__ void glds_x4(const void *gsrc, unsigned lds_base) {
    static_assert(IMM >= 0 && IMM < 4096 && M0ADD >= 0, "13-bit signed immediate");
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base), "i"(M0ADD), "i"(IMM)
                 : "memory", "scc");
}
// wave-uniform base (SGPR pair) + per-lane byte offset
template <int M0ADD>
__device__ __forceinline__ void glds_x4_u(const void *sbase, unsigned voff, unsigned lds_base) {
    static_assert(M0ADD >= 0, "");
    unsigned keep;
    // the base IS uniform; readfirstlane makes the register class certain ("s" alone is not)
    const unsigned long long a = (unsigned long long)(uintptr_t)sbase;
    const unsigned long long su = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull));
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(su), "s"(lds_base), "i"(M0ADD)
                 : "memory", "scc");
}
// vmcnt counts loads, stores and DMA together, in issue order: "all but the KEEP youngest are
// done". KEEP must be a LOWER bound of the vector-memory instructions issued after the DMA being
// waited for (the stage's result stores), or the image is read before it has landed.
template <int KEEP = 0>
__device__ __forceinline__ void dma_wait() {
    static_assert(KEEP >= 0 && KEEP < 64, "6-bit vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(KEEP) : "memory");
}
__device__ __forceinline__ void lds_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

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
    static constexpr int oL = 0;
    static constexpr int oY = oL + ((LW + 3) & ~3);  // y_t, later d_t : element k at oY + k
    static constexpr int oR = oY + 4 * SY;         // r_t (eq residual of row block t): row r at oR + r
    static constexpr int oS = oR + 4 * SW;         // s_t = (J d)_eq
    static constexpr int RECW = oS + 4 * SW;
    // LDS stage image: the inputs of the NEXT stage are fetched by DMA while the current stage
    // computes (one wave per SIMD: nothing else hides the HBM latency). Two layouts:
    //  big segments (F_t, the workspace record): instance-major, one x4 instruction per instance
    //    (up to 1 KB contiguous), instance p at word p*stride, stride = 4 mod 32 (bank spread);
    //  small segments: chunk-interleaved over the 16 instances, one x4 instruction per 16 words
    //    of every instance (word e of instance p at 256*(e/16) + 16p + e%16).
    // The LEN%4 trailing words of a segment travel in one dword instruction into a 64-word block
    // (word r of instance p at 4p + r): a DMA never reads past the end of a segment.
    __host__ __device__ static constexpr int big_stride(int len) {
        int st = len;
        while ((st & 31) != 4) ++st;
        return st;
    }
    __host__ __device__ static constexpr int small_words(int len) {  // image words of a small segment
        return ((len / 4 + 3) / 4) * 256 + ((len % 4) ? 64 : 0);
    }
    static constexpr int SF = big_stride(NX * N), SR = big_stride(oY + N);
    static constexpr int iF = 0;                          // F_t rows (both sweeps)
    static constexpr int iFt = iF + 16 * SF;              // ... trailing words
    static constexpr int fZ = iFt + 64;                   // forward: z_t
    static constexpr int fZN = fZ + small_words(N);       //          z_{t+1}[x]
    static constexpr int fQ = fZN + small_words(NX);      //          diag Q_t
    static constexpr int fq = fQ + small_words(N);        //          q_t
    static constexpr int fC = fq + small_words(N);        //          c_t
    static constexpr int fLE = fC + small_words(NX);      //          lam (dynamics rows t)
    static constexpr int fLU = fLE + small_words(NX);     //          lam (upper | lower bound rows t)
    static constexpr int fBU = fLU + small_words(2 * NU); //          u_upper
    static constexpr int fBL = fBU + small_words(NU);     //          u_lower
    static constexpr int FWDW = fBL + small_words(NU);
    static constexpr int bR = iFt + 64;                   // backward: workspace record words [0, oY + N)
    static constexpr int bRt = bR + 16 * SR;
    static constexpr int BWDW = bRt + 64;
    static constexpr int IMGW = FWDW > BWDW ? FWDW : BWDW;  // whole-wave image, in words
    static constexpr int LDS_BYTES = IMGW * 4;
    // 4 waves per CU share 160 KB of LDS; fp64 keeps the register-load path
    static constexpr bool DMA = sizeof(real) == 4 && LDS_BYTES <= 38 * 1024 && NX * N <= 256 && oY + N <= 256 &&
                                15 * SF + 4 * SW * N <= IMGW;
    __host__ __device__ static constexpr int M(int T) { return T * NX + 2 * T * NU; }
    __host__ __device__ static constexpr size_t ws_words(int B, int T) { return (size_t)B * T * RECW; }
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
    // DMA path (C::DMA): the wave's stage image in LDS
    float *lds;
    unsigned ldsb;       // its LDS byte address (M0 base)
    int lane, ip;        // lane in the wave, instance within the wave (lane / 4)
    int b0, Bn;          // first instance of the wave, batch size (wave-uniform)
    const real *uF;      // wave-uniform bases of the big segments
    const real *urec;

    __device__ __forceinline__ void init_image(float *img, int lane_, int b0_, int Bn_, const real *F_, const real *ws_) {
        lds = img;
        ldsb = (unsigned)(uintptr_t)img;
        lane = lane_;
        ip = lane_ >> 2;
        b0 = b0_;
        Bn = Bn_;
        uF = F_;
        urec = ws_;
    }
    // ---- small segments
    template <int LEN, int OFF>
    __device__ __forceinline__ void dma_small(const real *seg) const {
        constexpr int NCH = LEN / 4, R = LEN % 4, NB = (NCH + 3) / 4;
        const float *src = (const float *)seg + 4 * q;
        if constexpr (NB >= 1) { if (q < NCH) glds_x4<OFF * 4, 0>(src, ldsb); }
        if constexpr (NB >= 2) { if (4 + q < NCH) glds_x4<(OFF + 256) * 4 - 64, 64>(src, ldsb); }
        if constexpr (NB >= 3) { if (8 + q < NCH) glds_x4<(OFF + 512) * 4 - 128, 128>(src, ldsb); }
        static_assert(NB <= 3, "small segment longer than 48 words");
        if constexpr (R > 0) { if (q < R) glds_dword<(OFF + 256 * NB) * 4, 0>((const float *)seg + 4 * NCH + q, ldsb); }
    }
    template <int LEN, int OFF>
    __device__ __forceinline__ float small_rep(int e) const {  // word e, the same in the 4 lanes
        constexpr int NCH = LEN / 4, NB = (NCH + 3) / 4;
        if (e < 4 * NCH) return lds[OFF + 256 * (e >> 4) + (e & 15) + 16 * ip];
        return lds[OFF + 256 * NB + (e - 4 * NCH) + 4 * ip];
    }
    template <int LEN, int OFF>
    __device__ __forceinline__ float small_own(int m) const {  // word 4m+q (garbage past LEN)
        constexpr int NCH = LEN / 4, NB = (NCH + 3) / 4;
        if (m < NCH) return lds[OFF + 256 * ((4 * m) >> 4) + ((4 * m) & 15) + q + 16 * ip];
        return lds[OFF + 256 * NB + q + 4 * ip];
    }
    // ---- big segments: instance P of the wave from ubase + min(b0 + P, Bn - 1) * stride
    template <int LEN, int ST_, int OFF, int P>
    __device__ __forceinline__ void big_unit(const real *ubase, size_t stride) const {
        constexpr int NCH = LEN / 4;
        static_assert(NCH <= 64, "one instruction per instance");
        const int bp = (b0 + P < Bn) ? b0 + P : Bn - 1;
        if (lane < NCH) glds_x4_u<(OFF + P * ST_) * 4>(ubase + (size_t)bp * stride, 16u * lane, ldsb);
    }
    template <int LEN, int OFFT>
    __device__ __forceinline__ void big_tail(const real *own_seg) const {
        constexpr int NCH = LEN / 4, R = LEN % 4;
        if constexpr (R > 0) { if (q < R) glds_dword<OFFT * 4, 0>((const float *)own_seg + 4 * NCH + q, ldsb); }
    }
    // F[4s+q][k] of the image (zeros for rows >= NX)
    __device__ __forceinline__ void read_F_image(real (&W)[SW][N]) const {
        constexpr int NCH = (NX * N) / 4, R = (NX * N) % 4;
        constexpr int SL = (NX - 1) / 4, QL = (NX - 1) % 4;  // slot / lane of the last row
        const float *rowp = lds + C::iF + ip * C::SF + q * N;
#pragma unroll
        for (int s = 0; s < SW; ++s)
#pragma unroll
            for (int k = 0; k < N; ++k) {
                real v = rowp[4 * s * N + k];
                if (R > 0 && s == SL && k >= N - R) {  // trailing words of the segment: last row only
                    const real tv = lds[C::iFt + 4 * ip + (k - (N - R))];
                    v = (q == QL) ? tv : v;
                }
                W[s][k] = (4 * s + 3 < NX || 4 * s + q < NX) ? v : real(0);
            }
        static_assert(NCH * 4 + R == NX * N && NX <= 16 && 2 * N <= 40, "slot switches");
    }
    // The DMA of a stage is a list of units (one or two instructions each). A wave that issues
    // them back to back sits in the CU's memory pipeline for thousands of cycles (measured: 20 %
    // of the kernel), so inside the sweeps they are issued a few at a time between the blocks of
    // arithmetic of the stage before.
    static constexpr int FWD_UNITS = 26, BWD_UNITS = 34;
    template <int U>
    __device__ __forceinline__ void fwd_unit(int t, int td) const {
        if constexpr (U < 16) big_unit<NX * N, C::SF, C::iF, U>(uF + (size_t)td * NX * N, (size_t)(T - 1) * NX * N);
        else if constexpr (U == 16) big_tail<NX * N, C::iFt>(gF + (size_t)td * NX * N);
        else if constexpr (U == 17) dma_small<N, C::fZ>(gz + t * N);
        else if constexpr (U == 18) dma_small<NX, C::fZN>(gz + (td + 1) * N);
        else if constexpr (U == 19) dma_small<N, C::fQ>(gQd + t * N);
        else if constexpr (U == 20) dma_small<N, C::fq>(gq + t * N);
        else if constexpr (U == 21) dma_small<NX, C::fC>(gc + td * NX);
        else if constexpr (U == 22) dma_small<NX, C::fLE>(glam + td * NX);
        else if constexpr (U == 23) dma_small<2 * NU, C::fLU>(glam + T * NX + t * 2 * NU);
        else if constexpr (U == 24) dma_small<NU, C::fBU>(guhi + t * st_u);
        else if constexpr (U == 25) dma_small<NU, C::fBL>(gulo + t * st_u);
    }
    template <int U0, int CNT>
    __device__ __forceinline__ void fwd_units(int t) const {
        if constexpr (CNT > 0 && U0 < FWD_UNITS) {
            fwd_unit<U0>(t, (t < T - 1) ? t : (T > 1 ? T - 2 : 0));  // valid addresses for the last stage
            fwd_units<U0 + 1, CNT - 1>(t);
        }
    }
    template <int U>
    __device__ __forceinline__ void bwd_unit(int t, int td) const {
        if constexpr (U < 16) big_unit<NX * N, C::SF, C::iF, U>(uF + (size_t)td * NX * N, (size_t)(T - 1) * NX * N);
        else if constexpr (U == 16) big_tail<NX * N, C::iFt>(gF + (size_t)td * NX * N);
        else if constexpr (U < 33) big_unit<C::oY + N, C::SR, C::bR, U - 17>(urec + (size_t)t * RECW, (size_t)T * RECW);
        else if constexpr (U == 33) big_tail<C::oY + N, C::bRt>(recp(t));
    }
    template <int U0, int CNT>
    __device__ __forceinline__ void bwd_units(int t) const {
        if constexpr (CNT > 0 && U0 < BWD_UNITS) {
            bwd_unit<U0>(t, (t < T - 1) ? t : (T > 1 ? T - 2 : 0));
            bwd_units<U0 + 1, CNT - 1>(t);
        }
    }
    // slot is a constant after unrolling: the switch folds away (a loop variable cannot be a
    // template argument)
    template <int UPS>
    __device__ __forceinline__ void fwd_slot(int slot, int t) const {
        switch (slot) {
            case 0: fwd_units<0 * UPS, UPS>(t); break;
            case 1: fwd_units<1 * UPS, UPS>(t); break;
            case 2: fwd_units<2 * UPS, UPS>(t); break;
            case 3: fwd_units<3 * UPS, UPS>(t); break;
            case 4: fwd_units<4 * UPS, UPS>(t); break;
            case 5: fwd_units<5 * UPS, UPS>(t); break;
            case 6: fwd_units<6 * UPS, UPS>(t); break;
            case 7: fwd_units<7 * UPS, UPS>(t); break;
            case 8: fwd_units<8 * UPS, UPS>(t); break;
            case 9: fwd_units<9 * UPS, UPS>(t); break;
            case 10: fwd_units<10 * UPS, UPS>(t); break;
            case 11: fwd_units<11 * UPS, UPS>(t); break;
            case 12: fwd_units<12 * UPS, UPS>(t); break;
            case 13: fwd_units<13 * UPS, UPS>(t); break;
            case 14: fwd_units<14 * UPS, UPS>(t); break;
            case 15: fwd_units<15 * UPS, UPS>(t); break;
            default: break;
        }
    }
    template <int UPS>
    __device__ __forceinline__ void bwd_slot(int slot, int t) const {
        switch (slot) {
            case 0: bwd_units<0 * UPS, UPS>(t); break;
            case 1: bwd_units<1 * UPS, UPS>(t); break;
            case 2: bwd_units<2 * UPS, UPS>(t); break;
            case 3: bwd_units<3 * UPS, UPS>(t); break;
            case 4: bwd_units<4 * UPS, UPS>(t); break;
            case 5: bwd_units<5 * UPS, UPS>(t); break;
            case 6: bwd_units<6 * UPS, UPS>(t); break;
            case 7: bwd_units<7 * UPS, UPS>(t); break;
            case 8: bwd_units<8 * UPS, UPS>(t); break;
            case 9: bwd_units<9 * UPS, UPS>(t); break;
            case 10: bwd_units<10 * UPS, UPS>(t); break;
            case 11: bwd_units<11 * UPS, UPS>(t); break;
            case 12: bwd_units<12 * UPS, UPS>(t); break;
            case 13: bwd_units<13 * UPS, UPS>(t); break;
            case 14: bwd_units<14 * UPS, UPS>(t); break;
            case 15: bwd_units<15 * UPS, UPS>(t); break;
            case 16: bwd_units<16 * UPS, UPS>(t); break;
            case 17: bwd_units<17 * UPS, UPS>(t); break;
            case 18: bwd_units<18 * UPS, UPS>(t); break;
            case 19: bwd_units<19 * UPS, UPS>(t); break;
            case 20: bwd_units<20 * UPS, UPS>(t); break;
            case 21: bwd_units<21 * UPS, UPS>(t); break;
            case 22: bwd_units<22 * UPS, UPS>(t); break;
            case 23: bwd_units<23 * UPS, UPS>(t); break;
            case 24: bwd_units<24 * UPS, UPS>(t); break;
            case 25: bwd_units<25 * UPS, UPS>(t); break;
            case 26: bwd_units<26 * UPS, UPS>(t); break;
            case 27: bwd_units<27 * UPS, UPS>(t); break;
            case 28: bwd_units<28 * UPS, UPS>(t); break;
            case 29: bwd_units<29 * UPS, UPS>(t); break;
            case 30: bwd_units<30 * UPS, UPS>(t); break;
            case 31: bwd_units<31 * UPS, UPS>(t); break;
            case 32: bwd_units<32 * UPS, UPS>(t); break;
            case 33: bwd_units<33 * UPS, UPS>(t); break;
            case 34: bwd_units<34 * UPS, UPS>(t); break;
            case 35: bwd_units<35 * UPS, UPS>(t); break;
            case 36: bwd_units<36 * UPS, UPS>(t); break;
            case 37: bwd_units<37 * UPS, UPS>(t); break;
            case 38: bwd_units<38 * UPS, UPS>(t); break;
            case 39: bwd_units<39 * UPS, UPS>(t); break;
            default: break;
        }
    }
    __device__ __forceinline__ void issue_forward_dma(int t) const { fwd_units<0, FWD_UNITS>(t); }
    __device__ __forceinline__ void issue_backward_dma(int t) const { bwd_units<0, BWD_UNITS>(t); }

    __device__ __forceinline__ real uhi(int t, int j) const { return guhi[t * st_u + j]; }
    __device__ __forceinline__ real ulo(int t, int j) const { return gulo[t * st_u + j]; }
    __device__ __forceinline__ real *recp(int t) const { return rec + (size_t)t * RECW; }
#ifdef ALQP_PHASE_TIMING
    // debug build only (tools/phase_timing.py): cycles per phase. A stamp does not drain the
    // vector-memory queue (the DMA of the next stage has to stay in flight): waits show up in the
    // bucket of the statement that blocks on them
    unsigned long long tacc[10], tlast;
    __device__ __forceinline__ void stamp(int bucket) {
        unsigned long long now;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (bucket >= 0) tacc[bucket] += now - tlast;
        tlast = now;
    }
#define ALQP_STAMP(b) stamp(b)
#else
#define ALQP_STAMP(b)
#endif

    // F_t rows of this lane: row 4s+q (zeros for rows >= NX). Loads are unconditional (row
    // index clamped, result masked) so that all of a stage's loads go out in one batch:
    // a load inside a divergent branch cannot be hoisted and costs its own round trip.
    __device__ __forceinline__ void load_F_rows(int t, real (&W)[SW][N]) const {
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
    // equality residuals of all stages at the current z -> workspace (kernel start)
    __device__ __forceinline__ void residual_pass() {
        for (int t = 0; t < T - 1; ++t) {
            real W[SW][N], zt[N];
            load_F_rows(t, W);
            gload<N>(gz + t * N, zt);
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q;
                if (r < NX) {
                    real xn = gc[t * NX + r];
#pragma unroll
                    for (int k = 0; k < N; ++k) xn = fma_(W[s][k], zt[k], xn);
                    if (active) recp(t)[C::oR + r] = gz[(t + 1) * N + r] - xn;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < SW; ++s) {
            const int r = 4 * s + q;
            if (r < NX && active) recp(T - 1)[C::oR + r] = gz[r] - gx0[r];
        }
    }

    // ---- forward sweep: gradient, factorisation, forward substitution ------------------
    __device__ __forceinline__ void forward(real *g_out) {
        real S[ST], Sy[SW];
        real vprev[NX], Syrep[NX];
#pragma unroll
        for (int i = 0; i < ST; ++i) S[i] = 0;
#pragma unroll
        for (int s = 0; s < SW; ++s) Sy[s] = 0;
        if constexpr (C::DMA) issue_forward_dma(0);
        // stage 0: x_0 is pinned by the initial-state rows (eq row block T-1), al_utils.py:274
        {
            real z0[NX], xi[NX], li[NX];
            gload<NX>(gz, z0);
            gload<NX>(gx0, xi);
            gload<NX>(glam + (T - 1) * NX, li);
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                real r = z0[j] - xi[j];
                vprev[j] = fma_(rho, r, li[j]);
                Syrep[j] = 0;
                if ((j & 3) == q && active) recp(T - 1)[C::oR + j] = r;
            }
        }
        for (int t = 0; t < T; ++t) {
            const bool dyn = t < T - 1;
            real W[SW][N];
            real Y[N], D[N];
            real v[SW];
            real *rp = recp(t);
            // ---- loads (one batch) + residual + multiplier estimate
            {
                real zt[N], Qt[N], qt[N];
                real cs[SW], zn[SW], lm[SW];
                real lu[NU], ll[NU], bu[NU], bl[NU];
                if constexpr (C::DMA) {
                    // the image of stage t has landed; the previous stage's record stores (at least
                    // one per tile of L) were issued after its DMA and may still be in flight
                    if (t == 0) dma_wait<0>();
                    else dma_wait<SH * (SH + 1) / 2>();
                    read_F_image(W);
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        zt[k] = small_rep<N, C::fZ>(k);
                        Qt[k] = small_rep<N, C::fQ>(k);
                        qt[k] = small_rep<N, C::fq>(k);
                    }
#pragma unroll
                    for (int s = 0; s < SW; ++s) {
                        cs[s] = small_own<NX, C::fC>(s);
                        zn[s] = small_own<NX, C::fZN>(s);
                        lm[s] = small_own<NX, C::fLE>(s);
                    }
#pragma unroll
                    for (int j = 0; j < NU; ++j) {
                        lu[j] = small_rep<2 * NU, C::fLU>(j);
                        ll[j] = small_rep<2 * NU, C::fLU>(NU + j);
                        bu[j] = small_rep<NU, C::fBU>(j);
                        bl[j] = small_rep<NU, C::fBL>(j);
                    }
                    lds_reads_done();  // the image may be overwritten from here on
                } else {
                    gload<N>(gz + t * N, zt);
                    gload<N>(gQd + t * N, Qt);
                    gload<N>(gq + t * N, qt);
                    const int td = dyn ? t : (T > 1 ? T - 2 : 0);  // valid addresses for the last stage
                    load_F_rows(td, W);
#pragma unroll
                    for (int s = 0; s < SW; ++s) {
                        const int r = 4 * s + q, rc = r < NX ? r : NX - 1;
                        cs[s] = gc[td * NX + rc];
                        zn[s] = gz[(td + 1) * N + rc];
                        lm[s] = glam[td * NX + rc];
                    }
#pragma unroll
                    for (int j = 0; j < NU; ++j) {
                        lu[j] = glam[T * NX + t * 2 * NU + j];
                        ll[j] = glam[T * NX + t * 2 * NU + NU + j];
                        bu[j] = uhi(t, j);
                        bl[j] = ulo(t, j);
                    }
                }
                ALQP_STAMP(0);  // forward: inputs in registers
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
#pragma unroll
                    for (int k = 0; k < N; ++k) xn = fma_(W[s][k], zt[k], xn);
                    const real rr = zn[s] - xn;
                    const bool ok = dyn && r < NX;
                    v[s] = ok ? fma_(rho, rr, lm[s]) : real(0);
                    if (ok && active) rp[C::oR + r] = rr;
                }
                // ---- gradient (replicated in the 4 lanes) and diagonal of H_tt
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    real g = fma_(Qt[j], zt[j], qt[j]);
                    real d = Qt[j];
                    if (j < NX) {
                        g += vprev[j];
                        d += rho;
                    } else {
                        const int ju = j - NX;
                        real vu = zt[j] - bu[ju], vl = -zt[j] + bl[ju];
                        real au = vu >= 0 ? real(1) : real(0), al = vl >= 0 ? real(1) : real(0);
                        d = fma_(rho, au + al, d);
                        g += fma_(rho, fmax_(vu, real(0)), lu[ju]) - fma_(rho, fmax_(vl, real(0)), ll[ju]);
                    }
                    if (dyn) {
                        real p = 0;
#pragma unroll
                        for (int s = 0; s < SW; ++s) p = fma_(W[s][j], v[s], p);
                        g -= qsum(p);
                    }
                    D[j] = d;
                    Y[j] = -g - ((j < NX) ? Syrep[j] : real(0));
                    if (g_out && (j & 3) == q) g_out[t * N + j] = g;
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
                    if (4 * s + c < N) H[C::hidx(s, 4 * s + c)] = (q == c) ? D[4 * s + c] : real(0);
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
                    // a slice of the next stage's DMA (dyn <=> there is a next stage)
                    if constexpr (C::DMA) fwd_slot<(FWD_UNITS + NX - 1) / NX>(r, t + 1);
                }
            }
            ALQP_STAMP(1);  // forward: residual, gradient, H assembly incl. F'F
            // ---- right-looking panel factorisation
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const real p = qbv(H[C::hidx(j >> 2, j)], j);
                if (!(p > 0) && info == 0) info = t * N + j + 1;
                const real rinv = rsqrt_(p);
#pragma unroll
                for (int s = (j >> 2); s < SH; ++s) H[C::hidx(s, j)] *= rinv;
#pragma unroll
                for (int s = 0; s < SW; ++s) W[s][j] *= rinv;
                Y[j] *= rinv;
#pragma unroll
                for (int k = j + 1; k < N; ++k) {
                    const real lkj = qbv(H[C::hidx(k >> 2, j)], k);
#pragma unroll
                    for (int s = (k >> 2); s < SH; ++s) H[C::hidx(s, k)] = fma_(-H[C::hidx(s, j)], lkj, H[C::hidx(s, k)]);
#pragma unroll
                    for (int s = 0; s < SW; ++s) W[s][k] = fma_(-W[s][j], lkj, W[s][k]);
                    Y[k] = fma_(-Y[j], lkj, Y[k]);
                }
                // the solves divide by L[j][j]: keep 1/L[j][j] on the diagonal
                H[C::hidx(j >> 2, j)] = (q == (j & 3)) ? rinv : H[C::hidx(j >> 2, j)];
                if (dyn) {
#pragma unroll
                    for (int b = 0; b < NX; ++b) {
                        const real wbj = qbv(W[b >> 2][j], b);
#pragma unroll
                        for (int s = (b >> 2); s < SW; ++s) S[C::hidx(s, b)] = fma_(W[s][j], wbj, S[C::hidx(s, b)]);
                    }
#pragma unroll
                    for (int s = 0; s < SW; ++s) Sy[s] = fma_(W[s][j], Y[j], Sy[s]);
                }
            }
            ALQP_STAMP(2);  // forward: panel
            // ---- stage results -> workspace (each lane its own words)
            if (active) {
#pragma unroll
                for (int s = 0; s < SH; ++s)
#pragma unroll
                    for (int c = 0; c <= s; ++c)
                        if (C::lanes_of(s) == 4 || q < C::lanes_of(s))
                            gst4(rp + C::oL + C::lbase(s) + c * 4 * C::lanes_of(s) + 4 * q, H[C::hidx(s, 4 * c)],
                                 H[C::hidx(s, 4 * c + 1)], H[C::hidx(s, 4 * c + 2)], H[C::hidx(s, 4 * c + 3)]);
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if ((j & 3) == q) rp[C::oY + j] = Y[j];
            }
            // ---- carry to the next stage: replicated v = lam + rho r and W_t y_t
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                vprev[j] = qbv(v[j >> 2], j);
                Syrep[j] = qbv(Sy[j >> 2], j);
            }
            ALQP_STAMP(3);  // forward: stores issued
        }
    }

    // ---- backward sweep ---------------------------------------------------------------
    __device__ __forceinline__ void backward() {
        constexpr int UPSB = (BWD_UNITS + 2 * N - 1) / (2 * N);  // DMA units per loop iteration
        real dxn[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) dxn[j] = 0;
        if constexpr (C::DMA) {
            dma_wait();  // the forward sweep's stores of record T-1 are complete
            issue_backward_dma(T - 1);
        }
        for (int t = T - 1; t >= 0; --t) {
            const bool dyn = t < T - 1;
            real *rp = recp(t);
            real H[HT];
            real yo[SY];
            real W[SW][N];
            if constexpr (C::DMA) {
                // younger than the DMA of this stage: the SY stores of d_{t+1} (exactly SY, below)
                if (t == T - 1) dma_wait<0>();
                else dma_wait<SY>();
#pragma unroll
                for (int s = 0; s < SH; ++s)
#pragma unroll
                    for (int c = 0; c <= s; ++c) {
                        const int ql = (C::lanes_of(s) == 4 || q < C::lanes_of(s)) ? q : 0;
                        const float *src = lds + C::bR + ip * C::SR + C::oL + C::lbase(s) + c * 4 * C::lanes_of(s) + 4 * ql;
                        const float4 v4 = *reinterpret_cast<const float4 *>(src);
                        H[C::hidx(s, 4 * c)] = v4.x;
                        H[C::hidx(s, 4 * c + 1)] = v4.y;
                        H[C::hidx(s, 4 * c + 2)] = v4.z;
                        H[C::hidx(s, 4 * c + 3)] = v4.w;
                    }
#pragma unroll
                for (int m = 0; m < SY; ++m) {
                    constexpr int NCHR = (C::oY + N) / 4;  // full chunks of the record segment
                    const int e = C::oY + 4 * m + q;
                    real v = lds[C::bR + ip * C::SR + e];
                    if (C::oY + 4 * m + 3 >= 4 * NCHR) {
                        const int et = e - 4 * NCHR;
                        const real tv = lds[C::bRt + 4 * ip + (et >= 0 && et < 4 ? et : 0)];
                        v = (et >= 0) ? tv : v;
                    }
                    yo[m] = (4 * m + q < N) ? v : real(0);
                }
                read_F_image(W);
                lds_reads_done();  // the image may be overwritten from here on
                if (!dyn && t > 0) issue_backward_dma(t - 1);  // stage T-1 has no loops to spread it over
            } else {
#pragma unroll
                for (int s = 0; s < SH; ++s)
#pragma unroll
                    for (int c = 0; c <= s; ++c) {
                        // lanes without a real row re-read lane 0's words (unconditional load, values unused)
                        const int ql = (C::lanes_of(s) == 4 || q < C::lanes_of(s)) ? q : 0;
                        gld4(rp + C::oL + C::lbase(s) + c * 4 * C::lanes_of(s) + 4 * ql, H[C::hidx(s, 4 * c)],
                             H[C::hidx(s, 4 * c + 1)], H[C::hidx(s, 4 * c + 2)], H[C::hidx(s, 4 * c + 3)]);
                    }
#pragma unroll
                for (int m = 0; m < SY; ++m) yo[m] = (4 * m + q < N) ? rp[C::oY + 4 * m + q] : real(0);
                load_F_rows(dyn ? t : (T > 1 ? T - 2 : 0), W);  // same batch as the record loads
            }
            real Y[N];
#pragma unroll
            for (int j = 0; j < N; ++j) Y[j] = qbv(yo[j >> 2], j);
            ALQP_STAMP(4);  // backward: inputs in registers
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
                    if constexpr (C::DMA) { if (t > 0) bwd_slot<UPSB>(j, t - 1); }
                }
                // w = L^{-1} v
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const real wj = vv[j] * qbv(H[C::hidx(j >> 2, j)], j);
                    vv[j] = wj;
#pragma unroll
                    for (int k = j + 1; k < N; ++k) vv[k] = fma_(-qbv(H[C::hidx(k >> 2, j)], k), wj, vv[k]);
                    if constexpr (C::DMA) { if (t > 0) bwd_slot<UPSB>(N + j, t - 1); }
                }
#pragma unroll
                for (int j = 0; j < N; ++j) Y[j] = fma_(rho, vv[j], Y[j]);
            }
            // d = L^{-T} rhs
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {
                const real di = Y[i] * qbv(H[C::hidx(i >> 2, i)], i);
                Y[i] = di;
#pragma unroll
                for (int j = 0; j < i; ++j) Y[j] = fma_(-qbv(H[C::hidx(i >> 2, j)], i), di, Y[j]);
            }
            if (active) {
                // own elements of d_t: one store instruction per slot (dma_wait<SY> counts them)
#pragma unroll
                for (int m = 0; m < SY; ++m) {
                    const real dv = sel4(Y[4 * m], (4 * m + 1 < N) ? Y[(4 * m + 1 < N) ? 4 * m + 1 : 0] : real(0),
                                         (4 * m + 2 < N) ? Y[(4 * m + 2 < N) ? 4 * m + 2 : 0] : real(0),
                                         (4 * m + 3 < N) ? Y[(4 * m + 3 < N) ? 4 * m + 3 : 0] : real(0), q);
                    if (4 * m + 3 < N || 4 * m + q < N) rp[C::oY + 4 * m + q] = dv;
                }
            }
            if (dyn) {
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    const int r = 4 * s + q;
                    real p = 0;
#pragma unroll
                    for (int k = 0; k < N; ++k) p = fma_(W[s][k], Y[k], p);
                    if (r < NX && active) rp[C::oS + r] = dxs[s] - p;
                }
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) dxn[j] = Y[j];
            ALQP_STAMP(5);  // backward: solves + stores
        }
        // initial-state rows: s = d_0[x]
#pragma unroll
        for (int j = 0; j < NX; ++j)
            if ((j & 3) == q && active) recp(T - 1)[C::oS + j] = dxn[j];
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
            real W[SW][N];
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
            // y = L^{-1} rhs
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const real yj = Y[j] * qbv(H[C::hidx(j >> 2, j)], j);
                Y[j] = yj;
#pragma unroll
                for (int k = j + 1; k < N; ++k) Y[k] = fma_(-qbv(H[C::hidx(k >> 2, j)], k), yj, Y[k]);
            }
            if (active) {
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if ((j & 3) == q) rp[C::oY + j] = Y[j];
            }
            // e = L^{-T} y for the coupling of the next stage
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
            real lu[SY], ll[SY], bu[SY], bl[SY];
#pragma unroll
            for (int m = 0; m < SY; ++m) {
                const int j = 4 * m + q, jc = (4 * m + 3 < N) ? j : (j < N ? j : N - 1);
                zz[m] = gz[t * N + jc];
                dd[m] = rp[C::oY + jc];
                QQ[m] = gQd[t * N + jc];
                qq[m] = gq[t * N + jc];
                if (m >= MU0) {
                    const int ju = jc >= NX ? jc - NX : 0;
                    lu[m] = glam[T * NX + t * 2 * NU + ju];
                    ll[m] = glam[T * NX + t * 2 * NU + NU + ju];
                    bu[m] = uhi(t, ju);
                    bl[m] = ulo(t, ju);
                }
            }
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q, rc = (4 * s + 3 < NX) ? r : (r < NX ? r : NX - 1);
                rv[s] = rp[C::oR + rc];
                sv[s] = rp[C::oS + rc];
                lv[s] = glam[t * NX + rc];
            }
#pragma unroll
            for (int m = 0; m < SY; ++m) {
                const int j = 4 * m + q;
                const real ok = (4 * m + 3 < N || j < N) ? real(1) : real(0);
                const real z = zz[m], d = at_z ? real(0) : dd[m] * ok, Qv = QQ[m] * ok, qv = qq[m] * ok;
                c0 = fma_(fma_(real(0.5) * Qv, z, qv), z, c0);
                c1 = fma_(fma_(Qv, z, qv), d, c1);
                c2 = fma_(real(0.5) * Qv * d, d, c2);
                if (m >= MU0) {
                    const real isu = (j >= NX && j < N) ? real(1) : real(0);
                    real alpha = 1;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        real zk = fma_(alpha, d, z);
                        real vu = zk - bu[m], vl = bl[m] - zk;
                        real cu = fmax_(vu, real(0)), cl = fmax_(vl, real(0));
                        acc[k] = fma_(isu, fma_(lu[m], vu, ll[m] * vl) + real(0.5) * rho * fma_(cu, cu, cl * cl), acc[k]);
                        alpha *= real(0.5);
                    }
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

    // z += alpha d ; r += alpha s   (own elements only)
    __device__ __forceinline__ void apply_step(real alpha) {
        if (!active) return;
        real *__restrict__ zp = gz;       // z and the workspace never overlap: lets the loads of
        real *__restrict__ wp = rec;      // several stages leave before the first store
#pragma unroll 4
        for (int t = 0; t < T; ++t) {
            real *__restrict__ rp = wp + (size_t)t * RECW;
            real zz[SY], dd[SY], rr[SW], ss[SW];
#pragma unroll
            for (int m = 0; m < SY; ++m) {
                const int j = 4 * m + q, jc = j < N ? j : N - 1;
                zz[m] = zp[t * N + jc];
                dd[m] = rp[C::oY + jc];
            }
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q, rc = r < NX ? r : NX - 1;
                rr[s] = rp[C::oR + rc];
                ss[s] = rp[C::oS + rc];
            }
#pragma unroll
            for (int m = 0; m < SY; ++m) {
                const int j = 4 * m + q;
                if (j < N) zp[t * N + j] = fma_(alpha, dd[m], zz[m]);
            }
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q;
                if (r < NX) rp[C::oR + r] = fma_(alpha, ss[s], rr[s]);
            }
        }
    }

    __device__ __forceinline__ real rplus2(int &bad) {
        real acc = 0;
        for (int t = 0; t < T; ++t) {
            const real *rp = recp(t);
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q;
                if (r < NX) { real rr = rp[C::oR + r]; acc = fma_(rr, rr, acc); }
            }
#pragma unroll
            for (int m = 0; m < SY; ++m) {
                const int j = 4 * m + q;
                if (j < N) {
                    real z = gz[t * N + j];
                    bad |= !(z - z == real(0));
                    if (j >= NX) {
                        real cu = fmax_(z - uhi(t, j - NX), real(0)), cl = fmax_(ulo(t, j - NX) - z, real(0));
                        acc += fma_(cu, cu, cl * cl);
                    }
                }
            }
        }
        bad = qor(bad);
        return qsum(acc);
    }

    // lam <- lam + rho r ; lam_ineq <- max(0, .)   (AL_mpc.py:316-317)
    __device__ __forceinline__ void dual_update() {
        if (!active) return;
        for (int t = 0; t < T; ++t) {
            const real *rp = recp(t);
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                const int r = 4 * s + q;
                if (r < NX) glam[t * NX + r] = fma_(rho, rp[C::oR + r], glam[t * NX + r]);
            }
#pragma unroll
            for (int m = 0; m < SY; ++m) {
                const int j = 4 * m + q;
                if (j >= NX && j < N) {
                    const int ju = j - NX;
                    real u = gz[t * N + j];
                    real *lu = glam + T * NX + t * 2 * NU + ju, *ll = lu + NU;
                    real a = fma_(rho, u - uhi(t, ju), *lu), c = fma_(rho, ulo(t, ju) - u, *ll);
                    *lu = a < 0 ? real(0) : a;
                    *ll = c < 0 ? real(0) : c;
                }
            }
        }
    }
};

}  // namespace alqp
