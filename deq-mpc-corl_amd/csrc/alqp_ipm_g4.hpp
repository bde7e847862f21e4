// alqp_ipm_g4.hpp - the interior-point QP solve with everything ON CHIP.
//
// Same algorithm as alqp_ipm.hip (qpth/solvers/pdipm/batch_LU.py:29-244 on the MPC-structured QP of
// qpth/qp_wrapper.py:295-321; structured elimination = oracle/ipm_oracle_impl.h solver 0), different
// placement. The generic kernel streams seven 920-word vectors, F and the Schur factor of every QP through
// a workspace (25 MB of traffic per QP at (20,13,4)); here a wavefront owns ONE QP and keeps
//   * the iterate, the two directions and the right-hand side in REGISTERS, stage-owned:
//       lane = 16 g + r; stage t = 4 i + g lives in slot i of group g; lane r holds element r of the stage's
//       state / multiplier slices (xs, ys) and, in the two "control" registers c0 / c1, quarter 0 (r = j):
//       upper-bound rows (s, z), quarter 1 (r = 4+j): lower-bound rows, quarter 2 (r = 8+j): the control u_j;
//   * F (both orientations are needed: F x and F'y), the factor of the Schur complement (unit-lower inverse
//     M_m = L_m^-1 of S~_m = L D L', packed, + 1/D) and one staging vector in LDS.
// The stage-local phases (residuals, K products, right-hand sides, outputs) run on all four groups at once.
// The horizon-sequential phases (block factorisation, substitution sweeps) are TWISTED: the block-tridiagonal
// Schur complement is eliminated from both ends towards the middle block - the same arithmetic as one pass from
// the top, half the sequential depth. Lanes 0-31 run the chain that starts at block 0, lanes 32-63 the chain
// that starts at block T-1, in ONE instruction stream (only LDS addresses and a few selects differ per half);
// inside a half the two groups work redundantly - a 16-lane row is the unit of the v_fmac_*_dpp row_newbcast
// mat-vec: lane r holds row r of the matrix, the vector is spread over the row, one instruction per column.
// HBM traffic: inputs once, the best iterate when it improves.
//
// Internal conventions: the multiplier block m (m = 0..T-1) belongs to the constraint that defines x_m:
//   c_m(x) = F_{m-1} x_{m-1} - x_m[:nx] + fh_m,  fh_m = f_{m-1} (m >= 1), fh_0 = x0, F_{-1} = 0
// i.e. block m >= 1 is the reference's dynamics row block m-1 and block 0 is MINUS its initial-state block
// (so that every block has the same form; the sign is applied when vectors are read / written in the
// reference's order). Schur complement S = A Phi^-1 A' + eps:
//   S_mm = F_{m-1} P_{m-1} F_{m-1}' + P_{m,x} + eps,   S_{m,m-1} = -F_{m-1}[:, :nx] diag(P_{m-1,x})
// Block elimination, S~_m^-1 = M_m' D_m^-1 M_m (A_m = F_m[:, :nx], P_m = P_{m,x}), mid = T / 2:
//   top chain    m = 0 .. mid-1:    S~_m = S_mm - S_{m,m-1} S~_{m-1}^-1 S_{m,m-1}',  v'_m = v_m + A_{m-1} (P_{m-1} o p_{m-1})
//   bottom chain m = T-1 .. mid+1:  S~_m = S_mm - S_{m+1,m}' S~_{m+1}^-1 S_{m+1,m},  v'_m = v_m + P_m o (A_m' p_{m+1})
//   (p_m = S~_m^-1 v'_m); the middle block takes both corrections; then outwards from y_mid = p_mid:
//   top  y_m = p_m + S~_m^-1 (P_m o (A_m' y_{m+1})),   bottom  y_m = p_m + S~_m^-1 (A_{m-1} (P_{m-1} o y_{m-1}))
// The off-diagonal factor blocks are never stored (A_m is in LDS anyway): three mat-vecs per block and direction.
// LDS order of the blocks: F_t at slot t (t < mid) or mid + T-2-t (the bottom chain walks its blocks in
// ascending addresses too); factor blocks: top chain and the middle at slots 0 .. mid, bottom chain at slots
// mid+1 .. T-1 in the order it produces them - step i of either chain is (per-lane base) + i * (block size).
//
// The file is written against a small execution policy X (per-lane value types V / VI / VM and the cross-lane
// and memory primitives) so that the SAME source runs on the GPU (alqp_ipm_g4_gpu.hpp: V = real, DPP asm)
// and in a 64-lane CPU emulator (tests/emu/wave_emu.hpp: V = real[64]) - the latter is how the lane-level
// logic is tested against the reference-pinned fixtures without a GPU (tests/test_ipm_g4_emu.py).
// Control flow is wave-uniform throughout; lane predicates only appear in selects and masked accesses (the DPP
// primitives require all 64 lanes active).
#pragma once

#include <math.h>

#include "alqp_ipm_args.hpp"
#include "mi_alqp.h"

#ifndef G4_FN
#define G4_FN inline
#endif
#ifndef G4_UNROLL
#define G4_UNROLL
#endif

namespace alqp_ipm_g4 {

using alqp_ipm::IpmArgs;
using alqp_ipm::Lay;

// Chain primitives of the policy X (one instruction per term on the GPU):
//   X::row<K0, CNT>(acc, x, m)    acc    += sum_{i<CNT} bcast_{K0+i}(x) * m[i]     one accumulator, one spread vector
//   X::multi<K0, CNT>(acc, x, m)  acc[i] += bcast_{K0+i}(x) * m,  i < CNT          rank-1 update of a row set
//   X::vec<K, CNT>(acc, x, m)     acc    += sum_{i<CNT} bcast_K(x[i]) * m[i]       one source lane, CNT registers
template <class X, int K0, int CNT, class V>
G4_FN void fmac_row(V &acc, const V &x, const V *m) { if constexpr (CNT > 0) X::template row<K0, CNT>(acc, x, m); }
template <class X, int K0, int CNT, class V>
G4_FN void fmac_multi(V *acc, const V &x, const V &m) { if constexpr (CNT > 0) X::template multi<K0, CNT>(acc, x, m); }
template <class X, int K, int CNT, class V>
G4_FN void fmac_vec(V &acc, const V *x, const V *m) { if constexpr (CNT > 0) X::template vec<K, CNT>(acc, x, m); }
//   X::rank<NA, NT>(acc, x, m)    acc[i] += bcast_i(x[k]) * m[k], i < NA, k < NT   NT rank-1 updates of a row set, ONE block
// N rank-1 updates in blocks of as many terms as one block takes (X::rank_max)
template <class X, int NA, int N, class V>
G4_FN void fmac_rank(V *acc, const V *x, const V *m) {
    constexpr int B = X::rank_max(NA);
    if constexpr (N > B) {
        X::template rank<NA, B>(acc, x, m);
        fmac_rank<X, NA, N - B>(acc, x + B, m + B);
    } else if constexpr (N > 0) {
        X::template rank<NA, N>(acc, x, m);
    }
}

template <typename real, int NX, int NU, int SL, class X, bool FULLT = false>
struct Solver {
    using V = typename X::V;
    using VI = typename X::VI;
    using VM = typename X::VM;
    static constexpr int N = NX + NU;
    static constexpr int NXL = NX * (NX - 1) / 2;   // strictly-lower entries of M (packed by rows)
    static constexpr int ZC = NXL + NX;             // [NXL, NXL+NX): 1/d_j ;  ZC: a cell that holds 0
    static constexpr int MSZ = (ZC + 2) & ~1;       // factor words per stage
    static constexpr int FSZ = NX * N;
    static constexpr int TMAX = 4 * SL;
    static_assert(NX >= 1 && NX <= 16 && NU >= 1 && NU <= 4, "row layout: nx <= 16 lanes, nu <= 4 per quarter");

    struct KV { V xs[SL], ys[SL], c0[SL], c1[SL]; };   // a KKT vector (x | s | z | y) in registers

    static ALQP_HD long lds_words(int T) {
        return (long)(T - 1) * FSZ + (long)T * MSZ + 2L * T * NX + 4L * T + NX + 3;   // + NX zero words, a 1, the base shift (.hip)
    }

    const IpmArgs<real> &a;
    const Lay<real, NX, NU> L;
    const int b, T;
    const real e;
    real *w;                             // this instance's workspace slab
    real *sF, *sM, *sV, *sPx, *sPu, *sZ; // LDS (sZ: NX words that hold 0, then one that holds 1)
    const real *Cdg, *cg, *Fg, *fg, *x0g;
    VI lane, r, g, qd, j, rc;            // lane-derived indices: recomputed by refresh() at the start of every phase
    V hq;                                // h on the bound rows (quarter 0: u_hi, quarter 1: -u_lo)
    V Px[SL], Pu[SL], Dt[SL];            // 1/Phi on the state / control rows, D~ on the bound rows
    int sFt, sCt, sft;
    KV cur;
#ifdef ALQP_G4_TIMING
    long long tlast, tacc[16];   // debug build (tools/g4_timing.sh): cycles since the previous tick, attributed per phase
#endif
    VI infov;                            // first non-positive pivot (block * nx + column + 1), 0 if none; uniform
    int info;
    real sc_best, sc_mu, sc_have, sc_iter;

    G4_FN Solver(const IpmArgs<real> &a_, real *lds, int b_)
        : a(a_), L(a_.T, true), b(b_), T(a_.T), e(a_.e), info(0), sc_best(0), sc_mu(0), sc_have(0), sc_iter(0) {
        w = a.ws + (long)b * a.ws_words;
        real *p = lds;
        sF = p; p += (long)(T - 1) * FSZ;
        sM = p; p += (long)T * MSZ;
        sV = p; p += (long)T * NX;
        sPx = p; p += (long)T * NX;
        sPu = p; p += 4L * T;
        sZ = p;
        Cdg = a.Cd + (long)b * a.sC_b;
        cg = a.c ? a.c + (long)b * a.sC_b : nullptr;
        Fg = a.F + (long)b * a.sF_b;
        fg = a.f ? a.f + (long)b * a.sf_b : nullptr;
        x0g = a.x0 ? a.x0 + (long)b * NX : nullptr;
        lane = X::lane_id();
        infov = X::splati(0);
#ifdef ALQP_G4_TIMING
        for (int i = 0; i < 16; ++i) tacc[i] = 0;
        tlast = X::now();
#endif
        refresh();
    }
    // Everything derived from the lane index (masks, LDS / global offsets) is loop-invariant, and hipcc hoists all
    // of it out of the iteration loop into registers that live for the whole kernel (and spill). Laundering the
    // lane index through an empty asm at the start of a phase bounds those live ranges to the phase.
    G4_FN void refresh() {
        X::launder(lane);
        r = lane & 15; g = lane >> 4; qd = r >> 2; j = r & 3;
        rc = X::seli(r < NX, r, X::splati(NX - 1));
    }
    // phases: 0 problem load, 1 residuals, 2 factor (Dt, Pu), 3 factor blocks, 4 apply right-hand side, 5 forward sweep,
    // 6 backward sweep, 7 apply outputs, 8 K product, 9 step lengths / updates, 10 workspace + outputs
    G4_FN void tick(int ph) {
#ifdef ALQP_G4_TIMING
        const long long now = X::now();
        tacc[ph] += now - tlast;
        tlast = now;
#else
        (void)ph;
#endif
    }
    G4_FN VM mXr() const { return r < NX; }
    G4_FN VM mQ() const { return (qd < 2) & (j < NU); }
    G4_FN VM mU() const { return (qd == 2) & (j < NU); }
    G4_FN VM g0() const { return lane < 16; }
    G4_FN V sgn() const { return X::sel(qd == 0, X::splat(real(1)), X::splat(real(-1))); }
    static G4_FN VM um(bool c) { return X::splati(c ? 1 : 0) != 0; }   // a uniform condition as a lane mask

    // The two elimination chains: top = blocks 0 .. mid-1 (lanes 0-31), bottom = blocks T-1 .. mid+1 (lanes 32-63,
    // starting dl = 0 or 1 steps late so that both arrive at the middle block after `steps` steps).
    struct Twist { int mid, dl, steps, MB0; };
    G4_FN Twist twist() const {
        Twist w;
        w.mid = T / 2;
        w.dl = w.mid - (T - 1 - w.mid);
        w.steps = w.mid;
        w.MB0 = w.mid + 1 - w.dl;   // factor slot of the bottom chain's step i: MB0 + i
        return w;
    }
    G4_FN VM bottom() const { return lane >= 32; }
    G4_FN VI slotF(const VI &t) const {   // LDS slot of F_t
        const int mid = T / 2;
        return X::seli(t < mid, t, X::splati(mid + T - 2) - t);
    }

    G4_FN VI tof(int i) const { return g + 4 * i; }
    // (FULLT: the launcher saw T == 4 * SL - every slot holds a stage and the per-slot validity masks fold away, +1.4 %)
    G4_FN VM vs(int i) const {
        if constexpr (FULLT) return um(true);
        else return tof(i) < T;
    }
    G4_FN VM mx(int i) const { return vs(i) & mXr(); }
    G4_FN VM mq(int i) const { return vs(i) & mQ(); }
    G4_FN VM mu(int i) const { return vs(i) & mU(); }
    static G4_FN V zero() { return X::splat(real(0)); }
    G4_FN V keep(const VM &m, const V &v) const { return X::sel(m, v, zero()); }

    // ---- vectors in the reference's order (workspace blocks, outputs) <-> registers ----------------
    G4_FN void load_kv(const real *blk, KV &v) const {
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const VI t = tof(i);
            v.xs[i] = X::g_ld(blk, t * N + r, mx(i));
            const VI sidx = X::seli(qd == 0, t * NU + j, t * NU + j + T * NU);
            v.c0[i] = X::g_ld(blk + L.os(), sidx, mq(i)) + X::g_ld(blk, t * N + NX + j, mu(i));
            v.c1[i] = X::g_ld(blk + L.oz(), sidx, mq(i));
            const VI yidx = X::seli(t == 0, r + (T - 1) * NX, (t - 1) * NX + r);
            const V y = X::g_ld(blk + L.oy(), yidx, mx(i));
            v.ys[i] = X::sel(t == 0, -y, y);
        }
    }
    G4_FN void store_kv(real *blk, const KV &v) const {
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const VI t = tof(i);
            X::g_st(blk, t * N + r, v.xs[i], mx(i));
            X::g_st(blk, t * N + NX + j, v.c0[i], mu(i));
            const VI sidx = X::seli(qd == 0, t * NU + j, t * NU + j + T * NU);
            X::g_st(blk + L.os(), sidx, v.c0[i], mq(i));
            X::g_st(blk + L.oz(), sidx, v.c1[i], mq(i));
            const VI yidx = X::seli(t == 0, r + (T - 1) * NX, (t - 1) * NX + r);
            X::g_st(blk + L.oy(), yidx, X::sel(t == 0, -v.ys[i], v.ys[i]), mx(i));
        }
    }

    // ---- problem data: F -> LDS, cost / affine terms / bounds -> registers -----------------------
    G4_FN void load_problem() {
        const int tot = (T - 1) * FSZ;
        sFt = (int)a.sF_t; sCt = (int)a.sC_t; sft = (int)a.sf_t;
        for (int base = 0; base < tot; base += 64) {
            const VI idx = lane + base;
            const VM ok = idx < tot;
            const VI t = idx / FSZ;
            const VI el = idx - t * FSZ;
            X::lds_st(sF, slotF(t) * FSZ + el, X::g_ld(Fg, t * sFt + el, ok), ok);
        }
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const VI t = tof(i);
            Px[i] = keep(mx(i), X::rcp(Cdx(i) + e));
            X::lds_st(sPx, t * NX + r, Px[i], mx(i));
            Pu[i] = zero();
            Dt[i] = zero();
        }
        hq = zero();
        if (a.uhi && a.ulo) hq = X::g_ld(a.uhi, j, mQ() & (qd == 0)) - X::g_ld(a.ulo, j, mQ() & (qd == 1));
        X::lds_st(sM, lane * MSZ + ZC, zero(), lane < T);   // the zero cell of every factor slot
        X::lds_st(sZ, lane, X::sel(lane < NX, zero(), X::splat(real(1))), lane < NX + 1);
        X::fence();
        tick(0);
    }

    // cost diagonal / linear term / affine term of slot i's stage, re-read where they are used (L2-resident inputs:
    // registers are the scarce resource of this kernel, a few KB of reads per iteration are not)
    G4_FN V Cdx(int i) const { return X::g_ld(Cdg, tof(i) * sCt + r, mx(i)); }
    G4_FN V Cdu(int i) const { return X::g_ld(Cdg, tof(i) * sCt + NX + j, mu(i)); }
    G4_FN V cx(int i) const { return cg ? X::g_ld(cg, tof(i) * sCt + r, mx(i)) : zero(); }
    G4_FN V cu(int i) const { return cg ? X::g_ld(cg, tof(i) * sCt + NX + j, mu(i)) : zero(); }
    G4_FN V fh(int i) const {   // fh_m = f_{m-1} (m >= 1), x0 (m = 0)
        if (!(fg && x0g)) return zero();
        const VI t = tof(i);
        return X::g_ld(x0g, r, mx(i) & (t == 0)) + X::g_ld(fg, (t - 1) * sft + r, mx(i) & (t > 0));
    }

    // ---- stage shifts (cross-group moves through the LDS crossbar) -----------------------------------
    G4_FN void shift_down(const V *v, V *out) const {   // out at stage t <- v at stage t-1 (0 at stage 0)
        const VI src = (lane + 48) & 63;
        V tmp[SL];
        G4_UNROLL
        for (int i = 0; i < SL; ++i) tmp[i] = X::gather(v[i], src);
        G4_UNROLL
        for (int i = 0; i < SL; ++i) out[i] = X::sel(g == 0, i > 0 ? tmp[i > 0 ? i - 1 : 0] : zero(), tmp[i]);
    }
    G4_FN void shift_up(const V *v, V *out) const {     // out at stage t <- v at stage t+1 (0 at the last slot's end)
        const VI src = (lane + 16) & 63;
        V tmp[SL];
        G4_UNROLL
        for (int i = 0; i < SL; ++i) tmp[i] = X::gather(v[i], src);
        G4_UNROLL
        for (int i = 0; i < SL; ++i) out[i] = X::sel(g == 3, i < SL - 1 ? tmp[i < SL - 1 ? i + 1 : 0] : zero(), tmp[i]);
    }
    // control-register moves inside a row: u (quarter 2) -> quarters 0 and 1; (q0 -/+ q1) -> quarter 2
    // (row shifts on the DPP network: quarter 0 = lanes 0-3, quarter 1 = 4-7, quarter 2 = 8-11 of every 16-lane row;
    //  through ds_bpermute each of these was an exposed LDS round trip, ~95 of them per iteration)
    G4_FN V u_to_q(const V &c) const { return keep(mQ(), X::sel(qd == 0, X::template rshl<8>(c), X::template rshl<4>(c))); }
    G4_FN V q_diff_to_u(const V &c) const { return keep(mU(), X::template rshr<8>(c) - X::template rshr<4>(c)); }
    G4_FN V q_sum_to_u(const V &c) const { return keep(mU(), X::template rshr<8>(c) + X::template rshr<4>(c)); }

    // out (stage m, lanes r < nx) = F_{m-1} [x_{m-1} ; u_{m-1}]   (0 at m = 0); xs: state parts, c0: u in quarter 2.
    // The product is formed on the lanes of stage m-1 (its own x, u: no operand moves) and the RESULT moves down one stage.
    // Both F products keep TWO operand sets: a slot's matrix entries are read while the previous chain runs (hipcc reuses
    // one register set otherwise, and every chain starts with an exposed LDS round trip: 15 of them per pair of products).
    G4_FN VI f_base(int i) const { return slotF(X::mini(tof(i), X::splati(T - 2))) * FSZ; }
    G4_FN void Fx(const V *xs, const V *c0, V *out) const {
        V w[SL], fr[2][N];
        {
            const VI base = f_base(0) + rc * N;
            G4_UNROLL
            for (int k = 0; k < N; ++k) fr[0][k] = X::lds_ld(sF, base + k);
        }
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            if (i + 1 < SL) {
                const VI base = f_base(i + 1) + rc * N;
                G4_UNROLL
                for (int k = 0; k < N; ++k) fr[(i + 1) & 1][k] = X::lds_ld(sF, base + k);
            }
            V acc = zero();
            fmac_row<X, 0, NX>(acc, xs[i], fr[i & 1]);
            fmac_row<X, 8, NU>(acc, c0[i], fr[i & 1] + NX);
            w[i] = keep(mx(i) & (tof(i) < T - 1), acc);
        }
        shift_down(w, out);
    }
    // ox (lanes k < nx) = (F_t' yn)_k, ou (quarter 2) = (F_t' yn)_{nx+j}; yn = the next stage's multiplier slice
    G4_FN void FTy(const V *yn, V *ox, V *ou) const {
        V fx[NX], fu[NX];
        {
            const VI bx = f_base(0) + rc;
            G4_UNROLL
            for (int q = 0; q < NX; ++q) fx[q] = X::lds_ld(sF, bx + q * N);
        }
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const VI t = tof(i);
            {
                const VI bu = f_base(i) + NX + j;
                G4_UNROLL
                for (int q = 0; q < NX; ++q) fu[q] = X::lds_ld(sF, bu + q * N);
            }
            V ax = zero();
            fmac_row<X, 0, NX>(ax, yn[i], fx);
            if (i + 1 < SL) {
                const VI bx = f_base(i + 1) + rc;
                G4_UNROLL
                for (int q = 0; q < NX; ++q) fx[q] = X::lds_ld(sF, bx + q * N);
            }
            V au = zero();
            fmac_row<X, 0, NX>(au, yn[i], fu);
            ox[i] = keep(mx(i) & (t < T - 1), ax);
            ou[i] = keep(mu(i) & (t < T - 1), au);
        }
    }

    // ---- residuals + best iterate (batch_LU.py:86-146); rr <- -(rx, rs, rz, ry); returns "improved" ----
    G4_FN int resid(int it, KV &rr) {
        refresh();
        V yn[SL], fx_[SL], fu_[SL], Ax[SL];
        shift_up(cur.ys, yn);
        FTy(yn, fx_, fu_);
        const real *ext = a.ry_ext ? a.ry_ext + (long)b * L.ne : nullptr;
        if (!ext) Fx(cur.xs, cur.c0, Ax);
        V nxr = zero(), nzr = zero(), nyr = zero(), szs = zero();
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const VI t = tof(i);
            const V rx = keep(mx(i), Cdx(i) * cur.xs[i] + cx(i) + fx_[i] - cur.ys[i]);
            const V ru = keep(mu(i), Cdu(i) * cur.c0[i] + cu(i) + q_diff_to_u(cur.c1[i]) + fu_[i]);
            const V rs = keep(mq(i), cur.c0[i] * cur.c1[i]);
            const V rz = keep(mq(i), sgn() * u_to_q(cur.c0[i]) + cur.c0[i] - hq);
            V ry;
            if (ext) {
                const VI yidx = X::seli(t == 0, r + (T - 1) * NX, (t - 1) * NX + r);
                const V y = X::g_ld(ext, yidx, mx(i));
                ry = X::sel(t == 0, -y, y);
            } else {
                ry = keep(mx(i), Ax[i] - cur.xs[i] + fh(i));
            }
            rr.xs[i] = -rx; rr.ys[i] = -ry; rr.c0[i] = -(rs + ru); rr.c1[i] = -rz;
            nxr = nxr + rx * rx + ru * ru; nzr = nzr + rz * rz; nyr = nyr + ry * ry; szs = szs + rs;
        }
        const real sz = X::wave_sum(szs), nz2 = X::wave_sum(nzr), ny2 = X::wave_sum(nyr), nx2 = X::wave_sum(nxr);
        const real mu_ = fabs_(sz / real(L.ni));
        const real rsd = sqrt_(ny2) + sqrt_(nz2) + sqrt_(nx2) + real(L.ni) * mu_;
        const bool better = (sc_have == 0) || rsd < sc_best;
        if (better) {
            store_kv(w + L.best, cur);
            sc_best = rsd; sc_have = 1; sc_iter = real(it);
        }
        sc_mu = mu_;
        tick(1);
        return better ? 1 : 0;
    }
    static G4_FN real fabs_(real v) { return v < 0 ? -v : v; }
    static G4_FN real sqrt_(float v) { return sqrtf(v); }
    static G4_FN real sqrt_(double v) { return sqrt(v); }

    // ---- factorisation at the current (s, z) ----------------------------------------------------------
    // Per step both chains take one block each (lane r = row r of every 13 x 13 matrix; the two groups of a half alike):
    //   S = F P F' + P_x + eps  -  Z D^-1 Z',   Z = (coupling to the chain's previous block) M_prev'
    //       top: Z[r][c] = -sum_k A_{m-1}[r][k] P_{m-1}[k] M_{m-1}[c][k]; bottom: -sum_k P_m[r] A_m[k][r] M_{m+1}[c][k]
    //       (rank-1 updates of the row set, X::multi, with the previous block's M rows still in registers)
    //   S = L D L' and M = L^-1 in ONE pivot loop: the row operation that eliminates column C of S is applied to an
    //   identity alongside (X::self), so M's rows come out on the lanes that own them - no separate triangular inversion
    //   with its dependent chains.
    // The last step is the middle block: the top half forms P_x + eps + F P F' - Z D^-1 Z', the bottom half only its
    // - Z D^-1 Z'; the halves exchange and add, and both factor the same matrix.
    G4_FN void factor() {
        refresh();
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            Dt[i] = keep(mq(i), X::rcp(cur.c0[i] * X::rcp(cur.c1[i] + e) + e));
            Pu[i] = keep(mu(i), X::rcp(Cdu(i) + e + q_sum_to_u(Dt[i])));
            X::lds_st(sPu, tof(i) * 4 + j, Pu[i], mu(i));
        }
        X::fence();
        tick(2);
        const Twist tw = twist();
        V Mrow[NX];      // row r of the chain's previous M (unit lower: Mrow[k] = M[r][k] for k < r, 1 at k = r, 0 beyond)
        V dprev[NX];     // 1 / d_k of that block (uniform inside a half)
        G4_UNROLL
        for (int k = 0; k < NX; ++k) { Mrow[k] = zero(); dprev[k] = zero(); }
        for (int i = 0; i <= tw.steps; ++i) {
            const bool is_mid = i == tw.steps;
            const VM bot = bottom();
            const int mb_ = T - 1 + tw.dl - i;                                // the bottom chain's block (T: not started yet)
            const VI mv = X::seli(bot, X::splati(mb_), X::splati(i));
            const VM live = !bot | um(i >= tw.dl);
            const VM base = !bot | um(!is_mid);                               // who adds P_x + eps + F P F'
            const V pxo = X::lds_ld(sPx, X::seli(bot, X::splati((mb_ < T ? mb_ : T - 1) * NX), X::splati(i * NX)) + rc);
            const V dg = pxo + e;      // lane r: the P_x + eps part of S[r][r]; joins the pivot when column r is eliminated (ldl)
            V S[NX];
            G4_UNROLL
            for (int c = 0; c < NX; ++c) S[c] = zero();
            // rows of F_{m-1} (none for block 0), scaled by P_{m-1}
            const int st = i > 0 ? i - 1 : 0;
            const int sbt = tw.mid - tw.dl + i < T - 2 ? tw.mid - tw.dl + i : T - 2;
            const VI fb = X::seli(bot, X::splati(sbt * FSZ), X::splati(st * FSZ)) + rc * N;
            const VI pblk = X::seli(bot, X::splati(mb_ - 1), X::splati(st));
            const VM f_ok = base & live & (mv > 0);
            // lanes without an F term read their P from the zero words behind sPu: fp = 0 there (F itself is finite data)
            const VI pxb = X::seli(f_ok, pblk * NX, X::splati((int)(sZ - sPx)));
            const VI pub = X::seli(f_ok, pblk * 4, X::splati((int)(sZ - sPu)));
            // (P is fetched as ONE vector per lane row, element k on lane k, and spread by the row broadcast: a load per
            //  column would be seventeen more LDS round trips in front of the chains)
            const V pxv = X::lds_ld(sPx, pxb + rc);
            const V puv = X::lds_ld(sPu, pub + X::mini(j, X::splati(NU - 1)));
            V fr[N], fp[N];
            G4_UNROLL
            for (int k = 0; k < N; ++k) fr[k] = X::lds_ld(sF, fb + k);
            // the bottom chain's coupling columns (used after F P F'): read with this batch, under the rank-1 updates
            int zs = tw.mid + i - tw.dl - 1;
            zs = zs < 0 ? 0 : (zs > T - 2 ? T - 2 : zs);
            const VI zb = X::splati(zs * FSZ) + rc;
            V zraw[NX];
            G4_UNROLL
            for (int c = 0; c < NX; ++c) zraw[c] = X::lds_ld(sF, zb + c * N);
            X::template scale<0, NX>(fp, pxv, fr);
            X::template scale<0, NU>(fp + NX, puv, fr + NX);
            fmac_rank<X, NX, N>(S, fr, fp);   // F P F'
            tick(11);
            // coupling rows: top fp[c] = A_{m-1}[r][c] P_{m-1}[c]; bottom P_m[r] A_m[c][r] (column r of A_m; none for block T-1)
            const VM zb_ok = bot & live & (mv < T - 1);
            V zsrc[NX], Z[NX], zd[NX];
            G4_UNROLL
            for (int c = 0; c < NX; ++c) {
                zsrc[c] = X::sel(bot, keep(zb_ok, zraw[c] * pxo), fp[c]);
                Z[c] = zsrc[c];
            }
            if constexpr (NX > 1) X::template ztri<NX>(Z, Mrow, zsrc);   // Z[r][c] = -(zsrc[c] + sum_{k<c} M[c][k] zsrc[k]), M[c][k] = lane c's Mrow[k]
            G4_UNROLL
            for (int c = 0; c < NX; ++c) { zd[c] = Z[c] * dprev[c]; Z[c] = -Z[c]; }
            tick(12);
            fmac_rank<X, NX, NX>(S, Z, zd);   // S -= Z D^-1 Z'
            tick(13);
            if (is_mid) {
                const VI other = (lane + 32) & 63;
                G4_UNROLL
                for (int c = 0; c < NX; ++c) S[c] = S[c] + X::gather(S[c], other);
            }
            VI bad = X::splati(0);
            ldl<0>(S, Mrow, dprev, dg, bad);
            if (X::wave_any((bad != 0) & live)) {   // rare: record the first one (block * nx + column + 1)
                G4_UNROLL
                for (int c = 0; c < NX; ++c)
                    infov = X::seli((infov == 0) & live & (((bad >> (NX - 1 - c)) & 1) != 0), mv * NX + (c + 1), infov);
            }
            if (i < tw.dl) {   // the bottom chain has not started: it has no previous block
                G4_UNROLL
                for (int k = 0; k < NX; ++k) { Mrow[k] = keep(!bot, Mrow[k]); dprev[k] = keep(!bot, dprev[k]); }
            }
            tick(14);
            const VI mo = X::seli(bot, X::splati((tw.MB0 + i) * MSZ), X::splati(i * MSZ));
            const VM st_ok = ((lane & 16) == 0) & mXr() & live & (!bot | um(!is_mid));
            // Row r of M (entries k < r) is packed at r (r - 1) / 2. Every row lane stores its NX - 1 registers from that
            // offset in DESCENDING k under one predicate: the entries k >= r (exact zeros) land in later rows' cells, whose
            // owners write them afterwards (their k is smaller) - no per-entry predicate. Within one k the lanes' addresses
            // differ (row offsets are distinct from row 1 on; row 0 has no entries and stays out).
            if constexpr (NX > 1) X::template lds_st_desc<NX - 1>(sM, mo + ((rc * (rc - 1)) >> 1), Mrow, st_ok & (r > 0));
            const VM d_ok = ((lane & 31) == 0) & live & (!bot | um(!is_mid));   // 1/d_k is uniform inside a half: one lane stores all
            G4_UNROLL
            for (int k = 0; k < NX; ++k) X::lds_st(sM, mo + (NXL + k), dprev[k], d_ok);
            X::fence();
            tick(15);
        }
        const int it = X::firsti(infov), ib = X::lanei(infov, 32);
        info = it ? it : ib;
        tick(3);
    }
    // column C of S~ = L D L': d = S[C][C] + its P_x + eps part (uniform; a non-positive pivot is replaced by |d| and
    // flagged), L[r][C] = S[r][C] / d for r > C; the row operation row_r -= L[r][C] row_C on the trailing columns of S
    // and on the leading columns of the identity that becomes M = L^-1 (X::pivot, one block). Column C of M below the
    // diagonal is -L[:, C] at this point; M's unit diagonal is never read (nor stored) and is not formed.
    template <int C>
    G4_FN void ldl(V *S, V *Mrow, V *dinv, const V &dg, VI &bad) {
        if constexpr (C < NX) {
            const V d = X::template bcast<C>(S[C] + dg);
            bad = bad + bad + X::seli(d > zero(), X::splati(0), X::splati(1));   // a shift register of the non-positive pivots
            const V di = X::rcp(X::absv(d));
            dinv[C] = di;
            const V nl = keep(r > C, -(S[C] * di));
            Mrow[C] = nl;
            if constexpr (NX > 1) X::template pivot<C, NX>(S, Mrow, nl);
            ldl<C + 1>(S, Mrow, dinv, dg, bad);
        }
    }

    // Per-lane word offsets of a sweep step's operands, relative to (uniform) step * block size: packed-M cells (r, k),
    // k < r / (k, r), k > r (the slot's zero cell otherwise), 1/d_r, and the 13 entries of the A operand - row r of
    // A (entries k) or column r (entries k * N apart), whichever the half's chain needs in the current direction.
    struct SweepIdx { VI R[NX], C[NX], D, A[NX]; };
    G4_FN void m_offsets(SweepIdx &o, const Twist &tw) const {
        const VI hb = X::seli(bottom(), X::splati(tw.MB0 * MSZ), X::splati(0));
        G4_UNROLL
        for (int k = 0; k < NX; ++k) {
            o.R[k] = hb + X::seli((r > k) & mXr(), ((rc * (rc - 1)) >> 1) + k, X::splati(ZC));
            o.C[k] = hb + X::seli((r < k) & mXr(), X::splati(k * (k - 1) / 2) + rc, X::splati(ZC));
        }
        o.D = hb + rc + NXL;
    }
    // inwards: top rows of A_{i-1}, bottom columns of A_m (m = T-1+dl-i, slot mid + i - dl - 1)
    G4_FN void a_offsets_in(SweepIdx &o, const Twist &tw) const {
        G4_UNROLL
        for (int k = 0; k < NX; ++k)
            o.A[k] = X::seli(bottom(), X::splati((tw.mid - tw.dl - 1) * FSZ + k * N) + rc, rc * N + (k - FSZ));
    }
    // outwards: top columns of A_i, bottom rows of A_{m-1} (slot mid + i - dl)
    G4_FN void a_offsets_out(SweepIdx &o, const Twist &tw) const {
        G4_UNROLL
        for (int k = 0; k < NX; ++k)
            o.A[k] = X::seli(bottom(), X::splati((tw.mid - tw.dl) * FSZ + k) + rc * N, X::splati(k * N) + rc);
    }
    // Operands of one sweep step, loaded from LDS one step AHEAD of their use (two sets, ping-pong): a wavefront is
    // alone on its SIMD, so nothing else hides the LDS latency of a step's ~40 matrix reads.
    static constexpr int NM = NX > 1 ? NX - 1 : 1;
    struct SweepOps { V a[NX], mr[NM], mc[NM], di, pa, pb, v; VI vi; };
    G4_FN void load_factor(int i, SweepOps &o, const SweepIdx &ix, const Twist &tw) const {
        const real *Mm = sM + (long)i * MSZ;
        G4_UNROLL
        for (int k = 0; k + 1 < NX; ++k) o.mr[k] = X::lds_ld(Mm, ix.R[k]);
        G4_UNROLL
        for (int k = 1; k < NX; ++k) o.mc[k - 1] = X::lds_ld(Mm, ix.C[k]);
        o.di = X::lds_ld(Mm, ix.D);
        const int mb_ = T - 1 + tw.dl - i;
        o.vi = X::seli(bottom(), X::splati((mb_ < T ? mb_ : T - 1) * NX), X::splati(i * NX)) + rc;
        // pa = P_x where the top chain scales and 1 where the bottom chain does, pb the other way round: selected by
        // ADDRESS (the 1 is the word behind the zero words) - a select on the loaded value would wait for the whole batch
        const VI one = X::splati((int)(sZ + NX - sPx));
        o.pa = X::lds_ld(sPx, X::seli(bottom(), one, o.vi));
        o.pb = X::lds_ld(sPx, X::seli(bottom(), o.vi, one));
        o.v = X::lds_ld(sV, o.vi);
    }
    G4_FN void load_step(int i, SweepOps &o, const SweepIdx &ix, const Twist &tw) const {
        const real *Fi = sF + (long)i * FSZ;
        G4_UNROLL
        for (int k = 0; k < NX; ++k) o.a[k] = X::lds_ld(Fi, ix.A[k]);
        load_factor(i, o, ix, tw);
    }
    // the middle block: both halves read the ONE factor block at slot mid; their A operands and scalings differ as usual
    G4_FN void load_mid(SweepOps &o, const SweepIdx &ix, const Twist &tw) const {
        const real *Fi = sF + (long)tw.steps * FSZ;
        G4_UNROLL
        for (int k = 0; k < NX; ++k) o.a[k] = X::lds_ld(Fi, ix.A[k]);
        const real *Mm = sM + (long)tw.mid * MSZ;
        const VI hb = X::seli(bottom(), X::splati(tw.MB0 * MSZ), X::splati(0));
        G4_UNROLL
        for (int k = 0; k + 1 < NX; ++k) o.mr[k] = X::lds_ld(Mm, ix.R[k] - hb);
        G4_UNROLL
        for (int k = 1; k < NX; ++k) o.mc[k - 1] = X::lds_ld(Mm, ix.C[k] - hb);
        o.di = X::lds_ld(Mm, ix.D - hb);
        o.vi = rc + tw.mid * NX;
        const VI one = X::splati((int)(sZ + NX - sPx));
        o.pa = X::lds_ld(sPx, X::seli(bottom(), one, o.vi));
        o.pb = X::lds_ld(sPx, X::seli(bottom(), o.vi, one));
        o.v = X::lds_ld(sV, o.vi);
    }
    // S~_m^-1 v = M' (D^-1 (M v)) with the block's packed factor in o
    G4_FN V Sinv(const V &v, const SweepOps &o) const {
        if constexpr (NX == 1) {
            return v * o.di;
        } else {
            V a1 = v;
            fmac_row<X, 0, NX - 1>(a1, v, o.mr);
            const V a2 = a1 * o.di;
            V p = a2;
            fmac_row<X, 1, NX - 1>(p, a2, o.mc);
            return p;
        }
    }

    // ---- structured solve of the regularised system, in place: bb <- Ktilde^-1 bb -------------------------
    G4_FN void apply(KV &bb) {
        refresh();
        // r1 = bx - G' D~ wv (control rows), wv = bs / (z + eps) - bz: formed here and again for the outputs
        // (bb keeps the right-hand side until then; registers held across the sweeps are what this kernel is short of)
        V prx[SL], pru[SL], Fp[SL];
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const V wv = keep(mq(i), bb.c0[i] * X::rcp(cur.c1[i] + e) - bb.c1[i]);
            const V r1u = keep(mu(i), bb.c0[i] - q_diff_to_u(Dt[i] * wv));
            prx[i] = Px[i] * bb.xs[i];
            pru[i] = Pu[i] * r1u;
        }
        Fx(prx, pru, Fp);
        G4_UNROLL
        for (int i = 0; i < SL; ++i) X::lds_st(sV, tof(i) * NX + r, Fp[i] - prx[i] - bb.ys[i], mx(i));
        X::fence();
        tick(4);
        // inwards: step i takes block i on the top chain and block T-1+dl-i on the bottom chain. (Both loops are unrolled
        // four steps deep: the step offsets of the LDS reads become immediates, and the two operand sets alternate by a
        // compile-time index.)
        const Twist tw = twist();
        const VM bot = bottom();
        const VM st_ok = ((lane & 16) == 0) & mXr();
        const VI other = (lane + 32) & 63;
        SweepIdx ix;
        m_offsets(ix, tw);
        a_offsets_in(ix, tw);
        SweepOps so[2];
        load_factor(0, so[0], ix, tw);
        if (tw.steps > 1) load_step(1, so[1], ix, tw);
        V p = Sinv(so[0].v, so[0]);       // step 0: nothing precedes either chain's first block
        X::lds_st(sV, so[0].vi, p, st_ok & (!bot | um(tw.dl == 0)));
        V q = keep(!bot | um(tw.dl == 0), p * so[0].pa);
        for (int i0 = 1; i0 < tw.steps; i0 += 4) {
            G4_UNROLL
            for (int ii = 0; ii < 4; ++ii) {
                const int i = i0 + ii;
                if (i >= tw.steps) break;
                const SweepOps &c = so[(ii + 1) & 1];
                if (i + 1 < tw.steps) load_step(i + 1, so[ii & 1], ix, tw);
                V acc = zero();
                fmac_row<X, 0, NX>(acc, q, c.a);
                p = Sinv(c.v + c.pb * acc, c);
                X::lds_st(sV, c.vi, p, st_ok);
                q = p * c.pa;
            }
        }
        {   // the middle block takes both chains' corrections; its solution starts both outward chains
            SweepOps &c = so[0];
            load_mid(c, ix, tw);
            V acc = zero();
            fmac_row<X, 0, NX>(acc, q, c.a);
            V t = c.pb * acc;
            t = t + X::gather(t, other);
            p = Sinv(c.v + t, c);
            X::lds_st(sV, c.vi, p, st_ok & !bot);
            q = p * c.pb;
        }
        tick(5);
        // outwards from the middle: y = p + S~^-1 (coupling to the block solved before)
        a_offsets_out(ix, tw);
        load_step(tw.steps - 1, so[0], ix, tw);
        for (int i0 = tw.steps - 1; i0 >= 0; i0 -= 4) {
            G4_UNROLL
            for (int ii = 0; ii < 4; ++ii) {
                const int i = i0 - ii;
                if (i < 0) break;
                const SweepOps &c = so[ii & 1];
                if (i > 0) load_step(i - 1, so[(ii + 1) & 1], ix, tw);
                V acc = zero();
                fmac_row<X, 0, NX>(acc, q, c.a);
                const V y = c.v + Sinv(c.pa * acc, c);
                X::lds_st(sV, c.vi, y, st_ok & (!bot | um(i >= tw.dl)));
                q = y * c.pb;
            }
        }
        X::fence();
        tick(6);
        // outputs
        refresh();
        V dy[SL], dyn[SL], fx_[SL], fu_[SL];
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const VI t = tof(i);
            dy[i] = keep(mx(i), X::lds_ld(sV, X::mini(t, X::splati(T - 1)) * NX + rc));
            dyn[i] = keep(mx(i) & (t < T - 1), X::lds_ld(sV, X::mini(t + 1, X::splati(T - 1)) * NX + rc));
        }
        FTy(dyn, fx_, fu_);
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const V zi = X::rcp(cur.c1[i] + e);
            const V wv = keep(mq(i), bb.c0[i] * zi - bb.c1[i]);
            const V r1u = keep(mu(i), bb.c0[i] - q_diff_to_u(Dt[i] * wv));
            const V du = keep(mu(i), Pu[i] * (r1u - fu_[i]));
            const V dz = keep(mq(i), Dt[i] * (sgn() * u_to_q(du) + wv));
            const V ds = keep(mq(i), (bb.c0[i] - cur.c0[i] * dz) * zi);
            bb.xs[i] = keep(mx(i), Px[i] * (bb.xs[i] - fx_[i] + dy[i]));
            bb.ys[i] = dy[i];
            bb.c0[i] = ds + du;
            bb.c1[i] = dz;
        }
        X::fence();
        tick(7);
    }

    // rr <- rr - K(z, s) l   (K without regularisation)
    G4_FN void Kmul_sub(const KV &l, KV &rr) {
        refresh();
        V yn[SL], fx_[SL], fu_[SL], Ax[SL];
        shift_up(l.ys, yn);
        FTy(yn, fx_, fu_);
        Fx(l.xs, l.c0, Ax);
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const V ox = keep(mx(i), Cdx(i) * l.xs[i] + fx_[i] - l.ys[i]);
            const V ou = keep(mu(i), Cdu(i) * l.c0[i] + q_diff_to_u(l.c1[i]) + fu_[i]);
            const V oS = keep(mq(i), cur.c1[i] * l.c0[i] + cur.c0[i] * l.c1[i]);
            const V oZ = keep(mq(i), sgn() * u_to_q(l.c0[i]) + l.c0[i]);
            const V oY = keep(mx(i), Ax[i] - l.xs[i]);
            rr.xs[i] = rr.xs[i] - ox;
            rr.c0[i] = rr.c0[i] - (ou + oS);
            rr.c1[i] = rr.c1[i] - oZ;
            rr.ys[i] = rr.ys[i] - oY;
        }
        tick(8);
    }

    static G4_FN void kv_copy(KV &d, const KV &v) {
        G4_UNROLL
        for (int i = 0; i < SL; ++i) { d.xs[i] = v.xs[i]; d.ys[i] = v.ys[i]; d.c0[i] = v.c0[i]; d.c1[i] = v.c1[i]; }
    }
    static G4_FN void kv_add(KV &d, const KV &v) {
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            d.xs[i] = d.xs[i] + v.xs[i]; d.ys[i] = d.ys[i] + v.ys[i];
            d.c0[i] = d.c0[i] + v.c0[i]; d.c1[i] = d.c1[i] + v.c1[i];
        }
    }
    static G4_FN void kv_swap(KV &p, KV &q) {
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            V t;
            t = p.xs[i]; p.xs[i] = q.xs[i]; q.xs[i] = t;
            t = p.ys[i]; p.ys[i] = q.ys[i]; q.ys[i] = t;
            t = p.c0[i]; p.c0[i] = q.c0[i]; q.c0[i] = t;
            t = p.c1[i]; p.c1[i] = q.c1[i]; q.c1[i] = t;
        }
    }
    // solve_kkt (batch_LU.py:212-244): out = Ktilde^-1 rr, one refinement step against K; rr is consumed.
    // ONE apply() site (a two-trip loop): the sweeps are the bulk of the code and must not be instantiated per use.
    // `out` first keeps the right-hand side while apply() works on rr in place, then the two swap roles.
    G4_FN void solve_kkt(KV &rr, KV &out) {
        kv_copy(out, rr);
        for (int pass = 0; pass < 2; ++pass) {
            apply(rr);
            if (pass == 0) {
                kv_swap(out, rr);        // out = Ktilde^-1 r, rr = r
                Kmul_sub(out, rr);       // rr = r - K out
            } else {
                kv_add(out, rr);
            }
        }
    }

    // get_step (batch_LU.py:200-208), per instance, over the s- or z-rows held in quarters 0 / 1 of v[], dv[]
    G4_FN real get_step(const V *v, const V *dv, int &nanflag) const {
        V mn = X::splat(real(INFINITY));
        VM nf = X::never();
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const V d = dv[i];
            V s = X::sel(d == zero(), X::splat(real(1)),
                         X::sel(d < zero(), -(v[i] * X::rcp(d)), X::sel(d != d, d, X::splat(real(INFINITY)))));
            s = X::sel(mq(i), s, X::splat(real(INFINITY)));
            nf = nf | (s != s);
            mn = X::vmin(mn, s);   // (a NaN ratio is flagged through nf, not through the minimum)
        }
        if (X::wave_any(nf)) nanflag = 1;
        return X::wave_min(mn);
    }

    G4_FN real step_length(const KV &d, real scale) const {   // min(1, scale * min(get_step(z), get_step(s)))
        int nf = 0;
        real al = get_step(cur.c1, d.c1, nf);
        const real al2 = get_step(cur.c0, d.c0, nf);
        al = al2 < al ? al2 : al;
        al = scale * al;
        al = al < real(1) ? al : real(1);
        return nf ? real(NAN) : al;
    }

    // ---- the forward launch: phases selected by flags (include/mi_alqp.h) -------------------------------
    // One loop runs the initial point (k = -1: batch_LU.py:44-81) and the iterations (residuals + best iterate
    // :86-146, predictor-corrector step :153-197), so that factor() and solve_kkt() are instantiated ONCE.
    G4_FN void run_forward() {
        load_problem();
        real *sc = w + L.scal;   // {resid_best, mu, have_best, iter_best}
        const bool do_init = (a.flags & ALQP_IPM_INIT) != 0, loop = (a.flags & ALQP_IPM_LOOP) != 0;
        const bool do_resid = loop || (a.flags & ALQP_IPM_RESID), do_step = loop || (a.flags & ALQP_IPM_STEP);
        const int n_iter = loop ? a.max_iter : ((do_resid || do_step) ? 1 : 0);
        if (!do_init) {
            load_kv(w + L.cur, cur);
            sc_best = sc[0]; sc_mu = sc[1]; sc_have = sc[2]; sc_iter = sc[3];
        }
        int improved = 0;
        bool moved = do_init;
        KV rr, da, dc;
        for (int k = do_init ? -1 : 0; k < n_iter; ++k) {
            const bool is_init = k < 0;
            refresh();
            if (is_init) {
                G4_UNROLL
                for (int i = 0; i < SL; ++i) {   // Sv = Zv = 1; solve_kkt(K, Ktilde, p, 0, -h, -b): r = (-p, 0, h, b), b^ = -fh
                    cur.xs[i] = zero(); cur.ys[i] = zero();
                    cur.c0[i] = keep(mq(i), X::splat(real(1)));
                    cur.c1[i] = keep(mq(i), X::splat(real(1)));
                    rr.xs[i] = -cx(i); rr.ys[i] = -fh(i); rr.c0[i] = -cu(i); rr.c1[i] = keep(mq(i), hq);
                }
                sc_best = 0; sc_mu = 0; sc_have = 0; sc_iter = 0;
            } else {
                if (do_resid) {
                    improved |= resid(a.iter0 + k, rr);
                    if (!do_step) store_kv(w + L.rr, rr);
                }
                if (!do_step) continue;
                if (!do_resid) load_kv(w + L.rr, rr);
            }
            factor();
            G4_UNROLL
            for (int i = 0; i < SL; ++i) { da.xs[i] = zero(); da.ys[i] = zero(); da.c0[i] = zero(); da.c1[i] = zero(); }
            for (int ph = 0; ph < (is_init ? 1 : 2); ++ph) {   // 0: affine direction (or the initial point), 1: centering-corrector
                if (ph == 1) {
                    const real al = step_length(da, real(1));
                    V t3 = zero(), t4 = zero();
                    G4_UNROLL
                    for (int i = 0; i < SL; ++i) {
                        t3 = t3 + keep(mq(i), (cur.c0[i] + al * da.c0[i]) * (cur.c1[i] + al * da.c1[i]));
                        t4 = t4 + keep(mq(i), cur.c0[i] * cur.c1[i]);
                    }
                    real sig = X::wave_sum(t3) / X::wave_sum(t4);
                    sig = sig * sig * sig;
                    const real musig = sc_mu * sig;
                    G4_UNROLL
                    for (int i = 0; i < SL; ++i) {   // only the s rows: -(-mu sig + ds_aff dz_aff)
                        rr.xs[i] = zero(); rr.ys[i] = zero(); rr.c1[i] = zero();
                        rr.c0[i] = keep(mq(i), X::splat(musig) - da.c0[i] * da.c1[i]);
                    }
                }
                solve_kkt(rr, dc);
                kv_add(da, dc);
            }
            refresh();
            if (is_init) {
                V ms = X::splat(real(INFINITY)), mz = X::splat(real(INFINITY));
                G4_UNROLL
                for (int i = 0; i < SL; ++i) {
                    const V s = X::sel(mq(i), da.c0[i], X::splat(real(INFINITY)));
                    const V z = X::sel(mq(i), da.c1[i], X::splat(real(INFINITY)));
                    ms = X::sel(s < ms, s, ms);
                    mz = X::sel(z < mz, z, mz);
                }
                const real mins = X::wave_min(ms), minz = X::wave_min(mz);
                kv_copy(cur, da);
                G4_UNROLL
                for (int i = 0; i < SL; ++i) {   // positivity shift (:71-81)
                    if (mins < 0) cur.c0[i] = cur.c0[i] - keep(mq(i), X::splat(mins - real(1)));
                    if (minz < 0) cur.c1[i] = cur.c1[i] - keep(mq(i), X::splat(minz - real(1)));
                }
            } else {
                const real al = step_length(da, real(0.999));
                G4_UNROLL
                for (int i = 0; i < SL; ++i) {
                    cur.xs[i] = cur.xs[i] + al * da.xs[i]; cur.ys[i] = cur.ys[i] + al * da.ys[i];
                    cur.c0[i] = cur.c0[i] + al * da.c0[i]; cur.c1[i] = cur.c1[i] + al * da.c1[i];
                }
                moved = true;
            }
        }
        tick(9);
        refresh();
        if (moved) store_kv(w + L.cur, cur);
        if (do_init || do_resid) X::store4(sc, sc_best, sc_mu, sc_have, sc_iter);
        X::gfence();
        if (a.flags & ALQP_IPM_FINAL) {
            const real *bst = w + L.best;
            for (int base = 0; base < L.NK; base += 64) {
                const VI i = lane + base;
                const V v = X::g_ld(bst, i, i < L.NK);
                X::g_st(a.o_x + (long)b * L.nz, i, v, i < L.nz);
                X::g_st(a.o_s + (long)b * L.ni, i - L.os(), v, (i >= L.os()) & (i < L.oz()));
                X::g_st(a.o_z + (long)b * L.ni, i - L.oz(), v, (i >= L.oz()) & (i < L.oy()));
                X::g_st(a.o_y + (long)b * L.ne, i - L.oy(), v, (i >= L.oy()) & (i < L.NK));
            }
        }
        X::store_scalars(a, b, sc_best, sc_mu, (int)sc_iter, improved, info);
        tick(10);
#ifdef ALQP_G4_TIMING
        X::publish_timing(tacc);
#endif
    }

    // ---- backward of DenseQPFunction (qp.py:238-252): K at the returned lams / slacks, no regularisation ----
    G4_FN void run_backward(const real *lams, const real *slacks) {
        load_problem();
        KV rr, d0;
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const VI t = tof(i);
            const VI sidx = X::seli(qd == 0, t * NU + j, t * NU + j + T * NU);
            cur.xs[i] = zero(); cur.ys[i] = zero();
            cur.c0[i] = X::g_ld(slacks + (long)b * L.ni, sidx, mq(i));
            cur.c1[i] = X::g_ld(lams + (long)b * L.ni, sidx, mq(i));
            rr.xs[i] = -X::g_ld(a.gbar + (long)b * L.nz, t * N + r, mx(i));
            rr.c0[i] = -X::g_ld(a.gbar + (long)b * L.nz, t * N + NX + j, mu(i));
            rr.c1[i] = zero(); rr.ys[i] = zero();
        }
        factor();
        solve_kkt(rr, d0);
        G4_UNROLL
        for (int i = 0; i < SL; ++i) {
            const VI t = tof(i);
            X::g_st(a.o_x + (long)b * L.nz, t * N + r, d0.xs[i], mx(i));
            X::g_st(a.o_x + (long)b * L.nz, t * N + NX + j, d0.c0[i], mu(i));
            const VI sidx = X::seli(qd == 0, t * NU + j, t * NU + j + T * NU);
            X::g_st(a.o_z + (long)b * L.ni, sidx, d0.c1[i], mq(i));
            const VI yidx = X::seli(t == 0, r + (T - 1) * NX, (t - 1) * NX + r);
            X::g_st(a.o_y + (long)b * L.ne, yidx, X::sel(t == 0, -d0.ys[i], d0.ys[i]), mx(i));
        }
        X::store_scalars(a, b, real(0), real(0), 0, 0, info);
    }
};

}  // namespace alqp_ipm_g4
