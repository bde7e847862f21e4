// TEST INFRASTRUCTURE - a 64-lane wavefront emulator for kernels written against an execution policy
// (deq-mpc-corl_amd/csrc/alqp_ipm_g4.hpp). Every per-lane value is an array of 64; the cross-lane primitives
// follow the gfx950 semantics the GPU policy (alqp_ipm_g4_gpu.hpp) maps them to: row_newbcast inside 16-lane
// rows, ds_bpermute gathers, in-order LDS. It lets the lane-level logic of a kernel be run on the CPU against
// the reference-pinned fixtures. Not part of the product; nothing under deq-mpc-corl_amd/ includes it.
#pragma once
#include <cmath>
#include <cstring>

namespace wave_emu {

constexpr int W = 64;

template <typename T>
struct Vec {
    T v[W];
};
struct Mask {
    bool v[W];
};

#define EMU_BIN(OP)                                                                                  \
    template <typename T> inline Vec<T> operator OP(const Vec<T> &a, const Vec<T> &b) {               \
        Vec<T> o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] OP b.v[l]; return o; }                  \
    template <typename T> inline Vec<T> operator OP(const Vec<T> &a, T b) {                           \
        Vec<T> o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] OP b; return o; }                       \
    template <typename T> inline Vec<T> operator OP(T a, const Vec<T> &b) {                           \
        Vec<T> o; for (int l = 0; l < W; ++l) o.v[l] = a OP b.v[l]; return o; }
EMU_BIN(+) EMU_BIN(-) EMU_BIN(*) EMU_BIN(/)
#undef EMU_BIN
#define EMU_IBIN(OP)                                                                                 \
    inline Vec<int> operator OP(const Vec<int> &a, const Vec<int> &b) {                               \
        Vec<int> o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] OP b.v[l]; return o; }                \
    inline Vec<int> operator OP(const Vec<int> &a, int b) {                                           \
        Vec<int> o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] OP b; return o; }
EMU_IBIN(&) EMU_IBIN(>>)
#undef EMU_IBIN
template <typename T> inline Vec<T> operator-(const Vec<T> &a) {
    Vec<T> o; for (int l = 0; l < W; ++l) o.v[l] = -a.v[l]; return o; }
#define EMU_CMP(OP)                                                                                  \
    template <typename T> inline Mask operator OP(const Vec<T> &a, const Vec<T> &b) {                 \
        Mask o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] OP b.v[l]; return o; }                    \
    template <typename T> inline Mask operator OP(const Vec<T> &a, T b) {                             \
        Mask o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] OP b; return o; }
EMU_CMP(<) EMU_CMP(>) EMU_CMP(<=) EMU_CMP(>=) EMU_CMP(==) EMU_CMP(!=)
#undef EMU_CMP
inline Mask operator&(const Mask &a, const Mask &b) { Mask o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] && b.v[l]; return o; }
inline Mask operator|(const Mask &a, const Mask &b) { Mask o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] || b.v[l]; return o; }
inline Mask operator!(const Mask &a) { Mask o; for (int l = 0; l < W; ++l) o.v[l] = !a.v[l]; return o; }

template <typename real>
struct EmuX {
    using V = Vec<real>;
    using VI = Vec<int>;
    using VM = Mask;
    static VI lane_id() { VI o; for (int l = 0; l < W; ++l) o.v[l] = l; return o; }
    static V splat(real x) { V o; for (int l = 0; l < W; ++l) o.v[l] = x; return o; }
    static VI splati(int x) { VI o; for (int l = 0; l < W; ++l) o.v[l] = x; return o; }
    static VM never() { VM o; for (int l = 0; l < W; ++l) o.v[l] = false; return o; }
    static V sel(const VM &m, const V &a, const V &b) { V o; for (int l = 0; l < W; ++l) o.v[l] = m.v[l] ? a.v[l] : b.v[l]; return o; }
    static VI seli(const VM &m, const VI &a, const VI &b) { VI o; for (int l = 0; l < W; ++l) o.v[l] = m.v[l] ? a.v[l] : b.v[l]; return o; }
    static VI mini(const VI &a, const VI &b) { VI o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] < b.v[l] ? a.v[l] : b.v[l]; return o; }
    static VI maxi(const VI &a, const VI &b) { VI o; for (int l = 0; l < W; ++l) o.v[l] = a.v[l] > b.v[l] ? a.v[l] : b.v[l]; return o; }
    static V absv(const V &a) { V o; for (int l = 0; l < W; ++l) o.v[l] = std::fabs(a.v[l]); return o; }
    static real first(const V &a) { return a.v[0]; }
    static int firsti(const VI &a) { return a.v[0]; }
    static int lanei(const VI &a, int l) { return a.v[l]; }
    static V rcp(const V &a) { V o; for (int l = 0; l < W; ++l) o.v[l] = real(1) / a.v[l]; return o; }
    // row_newbcast: lane K of each 16-lane row
    template <int K> static V bcast(const V &x) { V o; for (int l = 0; l < W; ++l) o.v[l] = x.v[(l & 48) + K]; return o; }
    static V gather(const V &x, const VI &src) { V o; for (int l = 0; l < W; ++l) o.v[l] = x.v[src.v[l] & 63]; return o; }
    // broadcast-FMA chains, in the GPU blocks' order of operations
    template <int K0, int C> static void row(V &acc, const V &x, const V *m) {
        const V xs = x;   // (acc and x never share a register on the GPU: early-clobber accumulators)
        V tmp = splat(real(0));
        for (int i = 0; i < C; ++i) {
            V &d = (false && (i & 1)) ? tmp : acc;
            for (int l = 0; l < W; ++l) d.v[l] = std::fma(xs.v[(l & 48) + K0 + i], m[i].v[l], d.v[l]);
        }
        
    }
    template <int K0, int C> static void multi(V *acc, const V &x, const V &m) {
        const V xs = x, ms = m;
        for (int i = 0; i < C; ++i) for (int l = 0; l < W; ++l) acc[i].v[l] = std::fma(xs.v[(l & 48) + K0 + i], ms.v[l], acc[i].v[l]);
    }
    template <int K, int C> static void self(V *acc, const V &m) {
        const V ms = m;
        for (int i = 0; i < C; ++i) {
            const V old = acc[i];
            for (int l = 0; l < W; ++l) acc[i].v[l] = std::fma(old.v[(l & 48) + K], ms.v[l], old.v[l]);
        }
    }
    template <int K0, int C> static void scale(V *out, const V &x, const V *m) {
        const V xs = x;
        for (int i = 0; i < C; ++i) for (int l = 0; l < W; ++l) out[i].v[l] = std::fma(xs.v[(l & 48) + K0 + i], m[i].v[l], T0());
    }
    static real T0() { return real(0); }
    // the whole-matrix blocks of the GPU policy (one asm statement each there), in their order of operations
    static constexpr int rank_max(int na) { return (30 - na) / 2 < 8 ? (30 - na) / 2 : 8; }
    template <int NA, int NT> static void rank(V *acc, const V *x, const V *m) {
        for (int k = 0; k < NT; ++k) multi<0, NA>(acc, x[k], m[k]);
    }
    template <int C, int NXX> static void pivot(V *S, V *M, const V &nl) {
        const V xs = S[C], ms = nl;
        for (int i = C + 1; i < NXX; ++i)
            for (int l = 0; l < W; ++l) S[i].v[l] = std::fma(xs.v[(l & 48) + i], ms.v[l], S[i].v[l]);
        for (int i = 0; i < C; ++i) {
            const V old = M[i];
            for (int l = 0; l < W; ++l) M[i].v[l] = std::fma(old.v[(l & 48) + C], ms.v[l], old.v[l]);
        }
    }
    template <int NXX> static void ztri(V *Z, const V *M, const V *s) {
        for (int k = 0; k + 1 < NXX; ++k)
            for (int c = k + 1; c < NXX; ++c)
                for (int l = 0; l < W; ++l) Z[c].v[l] = std::fma(M[k].v[(l & 48) + c], s[k].v[l], Z[c].v[l]);
    }
    template <int K, int C> static void vec(V &acc, const V *x, const V *m) {
        if constexpr (C > 13) { vec<K, 13>(acc, x, m); vec<K, C - 13>(acc, x + 13, m + 13); }
        else {
            V tmp = splat(real(0));
            for (int i = 0; i < C; ++i) {
                V &d = (false && (i & 1)) ? tmp : acc;
                for (int l = 0; l < W; ++l) d.v[l] = std::fma(x[i].v[(l & 48) + K], m[i].v[l], d.v[l]);
            }
            
        }
    }
    // memory
    static V lds_ld(const real *p, const VI &idx) { V o; for (int l = 0; l < W; ++l) o.v[l] = p[idx.v[l]]; return o; }
    static V lds_ldu(const real *p, int idx) { return splat(p[idx]); }
    static void lds_st(real *p, const VI &idx, const V &v, const VM &m) { for (int l = 0; l < W; ++l) if (m.v[l]) p[idx.v[l]] = v.v[l]; }
    template <int CNT> static void lds_st_desc(real *p, const VI &idx0, const V *v, const VM &m) {
        for (int k = CNT - 1; k >= 0; --k) for (int l = 0; l < W; ++l) if (m.v[l]) p[idx0.v[l] + k] = v[k].v[l];
    }
    static V g_ld(const real *p, const VI &idx, const VM &m) { V o; for (int l = 0; l < W; ++l) o.v[l] = m.v[l] ? p[idx.v[l]] : real(0); return o; }
    static void g_st(real *p, const VI &idx, const V &v, const VM &m) { for (int l = 0; l < W; ++l) if (m.v[l]) p[idx.v[l]] = v.v[l]; }
    static void fence() {}
    static void gfence() {}
    static void sched_fence() {}
    static void launder(VI &) {}
    template <int N> static V rshl(const V &x) { V o; for (int l = 0; l < W; ++l) o.v[l] = ((l & 15) + N < 16) ? x.v[l + N] : real(0); return o; }
    template <int N> static V rshr(const V &x) { V o; for (int l = 0; l < W; ++l) o.v[l] = ((l & 15) >= N) ? x.v[l - N] : real(0); return o; }
    // the GPU's order: inclusive scan inside each 16-lane row (shift 1, 2, 4, 8), lane 15 of rows 0 / 2 into rows 1 / 3,
    // lane 31 into the upper half; lane 63 holds the result
    template <class OP> static real wave_reduce(const V &a, OP op, bool sum) {
        V t = a;
        for (int s = 1; s < 16; s <<= 1) {
            V n = t;
            for (int l = 0; l < W; ++l) {
                if ((l & 15) >= s) n.v[l] = op(t.v[l], t.v[l - s]);
                else if (sum) n.v[l] = op(t.v[l], real(0));
            }
            t = n;
        }
        { V n = t; for (int l = 0; l < W; ++l) { const int row = l >> 4; if (row == 1 || row == 3) n.v[l] = op(t.v[l], t.v[(row - 1) * 16 + 15]); else if (sum) n.v[l] = op(t.v[l], real(0)); } t = n; }
        { V n = t; for (int l = 0; l < W; ++l) { if (l >= 32) n.v[l] = op(t.v[l], t.v[31]); else if (sum) n.v[l] = op(t.v[l], real(0)); } t = n; }
        return t.v[63];
    }
    static real wave_sum(const V &a) { return wave_reduce(a, [](real x, real y) { return x + y; }, true); }
    static real wave_min(const V &a) { return wave_reduce(a, [](real x, real y) { return std::fmin(x, y); }, false); }
    static V vmin(const V &a, const V &b) { V o; for (int l = 0; l < W; ++l) o.v[l] = std::fmin(a.v[l], b.v[l]); return o; }
    static bool wave_any(const VM &m) { for (int l = 0; l < W; ++l) if (m.v[l]) return true; return false; }
    static void store4(real *sc, real a, real b, real c, real d) { sc[0] = a; sc[1] = b; sc[2] = c; sc[3] = d; }
    template <class A>
    static void store_scalars(const A &a, int b, real best, real mu, int iter_best, int improved, int info) {
        if (a.o_resid) a.o_resid[b] = best;
        if (a.o_mu) a.o_mu[b] = mu;
        if (a.o_iter_best) a.o_iter_best[b] = iter_best;
        if (a.o_improved) a.o_improved[b] = improved;
        if (a.o_info && info && a.o_info[b] == 0) a.o_info[b] = info;
    }
};

}  // namespace wave_emu
