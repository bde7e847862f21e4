// alqp_ipm.hip - gfx950 kernels + C ABI of the INTERIOR-POINT QP solve.
//
// Replaces, for the MPC-structured QP of qpth/qp_wrapper.py:295-321 (diagonal cost, box bounds on the
// controls, linear(ised) dynamics), the reference's qp.DenseQPFunction (qp.py:187-270) =
// pdipm_b_LU.forward / solve_kkt (qpth/solvers/pdipm/batch_LU.py:29-244): a batched primal-dual
// interior-point method whose every iteration LU-factors a dense KKT matrix of order
// nz + 2 nineq + neq (920 at T=20, nx=13, nu=4) twice.
//
// Here: ONE WAVEFRONT PER QP INSTANCE. The same regularised KKT system (KKTeps = 1e-7 on the diagonal, the
// same one step of iterative refinement against the unregularised K) is solved without ever forming it:
//   * slack / inequality-multiplier rows are eliminated by hand (they are diagonal):
//       D~ = 1 / (s/(z+eps) + eps),  Phi = Q + eps + G' D~ G   (diagonal: G = +-I on the controls)
//   * equality multipliers by the Schur complement  S = A Phi^-1 A' + eps I,  block-tridiagonal with
//     nx x nx blocks once the initial-state rows are put first; S is factored by a block Cholesky whose
//     factor (explicit inverses of the diagonal blocks + the sub-diagonal blocks, so that the sweeps are
//     mat-vecs) lives in LDS for the whole iteration (27 KB at (20,13,4) fp32);
//   * every vector (iterate, best iterate, residuals, directions; ~8 x 920 words) is a lane-strided,
//     coalesced slab of a caller-provided workspace (L2 / Infinity-Cache resident).
// Per iteration: 1 factorisation + 4 right-hand sides x (solve + refinement solve) - against 2 dense LUs.
//
// Oracle: oracle/ipm_oracle_impl.h (solver 0 is this algorithm, solver 1 the literal dense LU); both are
// pinned by fixtures generated from the reference (tests/test_ip_golden.py).
#include <hip/hip_runtime.h>

#include "alqp_dims.hpp"
#include "alqp_ipm_args.hpp"
#include "alqp_ipm_g4_launch.hpp"
#include "mi_alqp.h"

namespace alqp_ipm {

// LDS image: [factor (optional)] [F tile NX*N] [Pinv tile 2N] [S NX*NX] [W NX*NX] [v T*NX]
template <typename real, int NX, int NU>
__host__ __device__ constexpr long lds_words(int T, bool fac_lds) {
    return (fac_lds ? 2L * T * NX * NX : 0) + NX * (NX + NU) + 2 * (NX + NU) + 2 * NX * NX + (long)T * NX + 64;
}

template <typename real>
__device__ inline real wave_sum(real v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename real>
__device__ inline real wave_min(real v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        real w = __shfl_xor(v, o, 64);
        v = (w < v) ? w : v;
    }
    return v;
}
__device__ inline int wave_or(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}
// value of lane `src` (a compile-time constant after unrolling -> v_readlane_b32, no LDS round trip)
__device__ inline float lane_bcast(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ inline double lane_bcast(double v, int src) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// all-reduce over the 4 lanes of a quad by DPP quad_perm (VALU, no LDS crossbar round trip like __shfl_xor)
template <int CTRL>
__device__ inline float quad_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ inline double quad_perm(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <typename real>
__device__ inline real quad_sum(real v) {
    v += quad_perm<0xB1>(v);  // quad_perm [1,0,3,2]
    v += quad_perm<0x4E>(v);  // quad_perm [2,3,0,1]
    return v;
}
__device__ inline float absr(float a) { return __builtin_fabsf(a); }
__device__ inline double absr(double a) { return __builtin_fabs(a); }
__device__ inline float sqrtr(float a) { return __builtin_sqrtf(a); }
__device__ inline double sqrtr(double a) { return __builtin_sqrt(a); }

template <typename real, int NX, int NU, bool FAC_LDS>
struct Ipm {
    static constexpr int N = NX + NU;
    static constexpr int NN = NX * NX;
    const IpmArgs<real> &a;
    const Lay<real, NX, NU> L;
    const int b, lane, T;
    real *w;                 // this instance's workspace slab
    real *Linv, *Wb;         // factor: [T][NN] each
    real *sF, *sP, *sS, *sW, *sv;   // LDS tiles
    const real *Cd, *c, *F, *f, *x0;
    int info;

    __device__ Ipm(const IpmArgs<real> &a_, real *lds, int b_)
        : a(a_), L(a_.T, !FAC_LDS), b(b_), lane(threadIdx.x), T(a_.T), info(0) {
        w = a.ws + (long)b * a.ws_words;
        real *p = lds;
        if (FAC_LDS) { Linv = p; p += (long)T * NN; Wb = p; p += (long)T * NN; }
        else { Linv = w + L.fac; Wb = Linv + (long)T * NN; }
        sF = p; p += NX * N; sP = p; p += 2 * N; sS = p; p += NN; sW = p; p += NN; sv = p;
        Cd = a.Cd + (long)b * a.sC_b; c = a.c ? a.c + (long)b * a.sC_b : nullptr;
        F = a.F + (long)b * a.sF_b; f = a.f ? a.f + (long)b * a.sf_b : nullptr;
        x0 = a.x0 ? a.x0 + (long)b * NX : nullptr;
    }
    __device__ __forceinline__ real cd(int k) const { return Cd[(long)(k / N) * a.sC_t + (k % N)]; }
    __device__ __forceinline__ real cc(int k) const { return c[(long)(k / N) * a.sC_t + (k % N)]; }
    __device__ __forceinline__ const real *Ft(int t) const { return F + (long)t * a.sF_t; }
    __device__ __forceinline__ real ff(int t, int r) const { return f[(long)t * a.sf_t + r]; }
    __device__ __forceinline__ real hh(int i) const {   // h = [u_upper tiled ; -u_lower tiled]  (qp_wrapper.py:651-652)
        const int Tn = T * NU;
        return i < Tn ? a.uhi[i % NU] : -a.ulo[(i - Tn) % NU];
    }
    static __device__ void sync() { __syncthreads(); }

    // (A' y)[k] for k = t*N + j: F_t' y_t (t < T-1)  -  y_{t-1} on the state rows (t >= 1)  +  y_init (t = 0)
    __device__ __forceinline__ real ATy(const real *y, int k) const {
        const int t = k / N, j = k % N;
        real acc = 0;
        if (t < T - 1) {
            const real *Fp = Ft(t);
#pragma unroll
            for (int r = 0; r < NX; ++r) acc += Fp[r * N + j] * y[t * NX + r];
        }
        if (j < NX) {
            if (t >= 1) acc -= y[(t - 1) * NX + j];
            else acc += y[(T - 1) * NX + j];
        }
        return acc;
    }
    // (A x)[i], reference row order: dynamics rows t*NX + r, then the initial-state rows
    __device__ __forceinline__ real Ax(const real *x, int i) const {
        const int t = i / NX, r = i % NX;
        if (t == T - 1) return x[r];
        const real *Fp = Ft(t) + r * N;
        real acc = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) acc += Fp[k] * x[t * N + k];
        return acc - x[(t + 1) * N + r];
    }
    // G' z at variable k (0 unless k is a control), G x at inequality row i
    __device__ __forceinline__ real GTz(const real *z, int k) const {
        const int t = k / N, j = k % N - NX;
        return j < 0 ? real(0) : z[t * NU + j] - z[T * NU + t * NU + j];
    }
    __device__ __forceinline__ real Gx(const real *x, int i) const {
        const int Tn = T * NU;
        const int iu = i < Tn ? i : i - Tn;
        const real u = x[(iu / NU) * N + NX + iu % NU];
        return i < Tn ? u : -u;
    }

    // ---- factorisation at (zd, sd) = current (z, s) ---------------------------------------------
    __device__ __forceinline__ void factor() {
        const real e = a.e;
        real *zc = w + L.cur + L.oz(), *sc = w + L.cur + L.os();
        real *Pinv = w + L.pinv, *Dt = w + L.dt;
        for (int i = lane; i < L.ni; i += 64) Dt[i] = real(1) / (sc[i] / (zc[i] + e) + e);
        sync();
        for (int k = lane; k < L.nz; k += 64) {
            const int t = k / N, j = k % N - NX;
            real p = cd(k) + e;
            if (j >= 0) p += Dt[t * NU + j] + Dt[T * NU + t * NU + j];
            Pinv[k] = real(1) / p;
        }
        sync();
        for (int m = 0; m < T; ++m) {
            real *Lm = Linv + (long)m * NN, *Wm = Wb + (long)m * NN;
            if (m == 0) {
                for (int i = lane; i < NN; i += 64) sS[i] = (i / NX == i % NX) ? Pinv[i / NX] + e : real(0);
            } else {
                const int t = m - 1;
                const real *Fp = Ft(t);
                for (int i = lane; i < NX * N; i += 64) sF[i] = Fp[i];
                for (int i = lane; i < 2 * N; i += 64) sP[i] = (t * N + i < L.nz) ? Pinv[t * N + i] : real(0);
                sync();
                const real sign = (m == 1) ? real(1) : real(-1);
                for (int i = lane; i < NN; i += 64) {
                    const int r = i / NX, q = i % NX;
                    real acc = 0;
                    if (q <= r) {
#pragma unroll
                        for (int k = 0; k < N; ++k) acc += sF[r * N + k] * sP[k] * sF[q * N + k];
                        if (q == r) acc += sP[N + r] + e;
                    }
                    sS[i] = acc;
                    sW[i] = sign * sF[r * N + q] * sP[q];    // S_{m,m-1} before the triangular solve
                }
                sync();
                // W_m = S_{m,m-1} L_{m-1}^{-T} = S_{m,m-1} (Linv_{m-1})'
                const real *Lp = Linv + (long)(m - 1) * NN;
                constexpr int NR = (NN + 63) / 64;
                real wtmp[NR];
#pragma unroll
                for (int it = 0; it < NR; ++it) {
                    const int i = lane + 64 * it;
                    real acc = 0;
                    if (i < NN) {
                        const int r = i / NX, q = i % NX;
                        for (int k = 0; k <= q; ++k) acc += sW[r * NX + k] * Lp[q * NX + k];
                    }
                    wtmp[it] = acc;
                }
                sync();
#pragma unroll
                for (int it = 0; it < NR; ++it) {
                    const int i = lane + 64 * it;
                    if (i < NN) { sW[i] = wtmp[it]; Wm[i] = wtmp[it]; }
                }
                sync();
                for (int i = lane; i < NN; i += 64) {
                    const int r = i / NX, q = i % NX;
                    if (q <= r) {
                        real acc = 0;
#pragma unroll
                        for (int k = 0; k < NX; ++k) acc += sW[r * NX + k] * sW[q * NX + k];
                        sS[i] -= acc;
                    }
                }
            }
            sync();
            // Cholesky of S_m and the inverse of its factor WITHOUT LDS round trips: lane r (< NX) holds row r
            // of the lower triangle in registers; pivots and column entries travel by v_readlane (constant lane
            // indices after unrolling). 13 column steps cost ~0.5 k instructions instead of 39 barriers.
            {
                real row[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k) row[k] = (lane < NX && k <= lane) ? sS[(lane < NX ? lane : 0) * NX + k] : real(0);
#pragma unroll
                for (int j = 0; j < NX; ++j) {
                    const real d = lane_bcast(row[j], j);
                    if (!(d > 0) && info == 0) info = m * NX + j + 1;
                    const real ad = absr(d);       // |d|: modified Cholesky on a non-positive pivot (flagged)
                    const real sd = sqrtr(ad);
                    const real lj = (lane == j) ? sd : row[j] / sd;   // L[r][j], r >= j (lanes above hold unused values)
                    row[j] = lj;
#pragma unroll
                    for (int k = j + 1; k < NX; ++k) row[k] -= lj * lane_bcast(lj, k);   // only entries k <= r are used
                }
                // Linv = L^{-1}: lane c builds column c by forward substitution; L[i][k] comes from lane i
                real col[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    real acc = (i == lane) ? real(1) : real(0);
#pragma unroll
                    for (int k = 0; k < i; ++k) acc -= lane_bcast(row[k], i) * col[k];
                    col[i] = (i >= lane) ? acc / lane_bcast(row[i], i) : real(0);
                }
                if (lane < NX) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) Lm[i * NX + lane] = col[i];
                }
            }
            sync();
        }
    }

    // ---- structured solve of the regularised system, rhs b (NK block, x|s|z|y), result into o ----
    // Phases (a barrier between them): r1 -> right-hand side of S -> 2 sweeps x T stages x 2 -> outputs.
    __device__ __forceinline__ void apply(const real *bb, real *o) {
        const real e = a.e;
        // the slabs do not overlap: telling the compiler lets it keep several lanes-strided iterations of the
        // element-wise loops in flight instead of one global round trip per iteration
        const real *__restrict__ zc = w + L.cur + L.oz(), *__restrict__ sc = w + L.cur + L.os();
        const real *__restrict__ Pinv = w + L.pinv, *__restrict__ Dt = w + L.dt;
        real *__restrict__ r1 = w + L.r1;
        const real *__restrict__ bx = bb, *__restrict__ bs = bb + L.os(), *__restrict__ bz = bb + L.oz(),
                   *__restrict__ by = bb + L.oy();
        real *__restrict__ dx = o, *__restrict__ ds = o + L.os(), *__restrict__ dz = o + L.oz(), *__restrict__ dy = o + L.oy();
        auto wv = [&](int i) { return bs[i] / (zc[i] + e) - bz[i]; };   // bs/(z+eps) - bz, recomputed where needed
#pragma unroll 4
        for (int k = lane; k < L.nz; k += 64) {
            const int t = k / N, j = k % N - NX;
            real v = bx[k];
            if (j >= 0) {
                const int iu = t * NU + j, il = T * NU + iu;
                v -= Dt[iu] * wv(iu) - Dt[il] * wv(il);
            }
            r1[k] = v;
        }
        sync();
        // rhs of S in internal block order (block 0 = initial-state rows, block t+1 = dynamics t) -> sv
#pragma unroll 4
        for (int i = lane; i < L.ne; i += 64) {
            const int m = i / NX, r = i % NX;
            real v;
            if (m == 0) v = Pinv[r] * r1[r] - by[(T - 1) * NX + r];
            else {
                const int t = m - 1;
                const real *Fp = Ft(t) + r * N;
                real acc = 0;
#pragma unroll
                for (int k = 0; k < N; ++k) acc += Fp[k] * Pinv[t * N + k] * r1[t * N + k];
                v = acc - Pinv[(t + 1) * N + r] * r1[(t + 1) * N + r] - by[t * NX + r];
            }
            sv[i] = v;
        }
        sync();
        // forward sweep: v_m <- Linv_m (v_m - W_m v_{m-1}); 4 lanes per row, the intermediate in sT
        const int row = lane >> 2, part = lane & 3;
        real *sT = sS;   // the S tile is free outside factor()
        for (int m = 0; m < T; ++m) {
            const real *Lm = Linv + (long)m * NN, *Wm = Wb + (long)m * NN;
            real acc = 0;
            if (m > 0 && row < NX)
                for (int k = part; k < NX; k += 4) acc += Wm[row * NX + k] * sv[(m - 1) * NX + k];
            acc = quad_sum(acc);
            if (row < NX && part == 0) sT[row] = sv[m * NX + row] - acc;
            sync();
            real acc2 = 0;
            if (row < NX)
                for (int k = part; k <= row; k += 4) acc2 += Lm[row * NX + k] * sT[k];
            acc2 = quad_sum(acc2);
            if (row < NX && part == 0) sv[m * NX + row] = acc2;
            sync();
        }
        // backward sweep: v_m <- Linv_m' (v_m - W_{m+1}' v_{m+1})
        for (int m = T - 1; m >= 0; --m) {
            const real *Lm = Linv + (long)m * NN;
            real acc = 0;
            if (m < T - 1 && row < NX) {
                const real *Wn = Wb + (long)(m + 1) * NN;
                for (int k = part; k < NX; k += 4) acc += Wn[k * NX + row] * sv[(m + 1) * NX + k];
            }
            acc = quad_sum(acc);
            if (row < NX && part == 0) sT[row] = sv[m * NX + row] - acc;
            sync();
            real acc2 = 0;
            if (row < NX)
                for (int k = row + part; k < NX; k += 4) acc2 += Lm[k * NX + row] * sT[k];
            acc2 = quad_sum(acc2);
            if (row < NX && part == 0) sv[m * NX + row] = acc2;
            sync();
        }
        // outputs in one phase: dy (reference row order), dx = Phi^-1 (r1 - A'dy) with dy read from the LDS
        // sweep result, dz / ds with the control's dx recomputed in place (no barrier in between)
        auto dyv = [&](int i) { const int t = i / NX, r = i % NX; return (t == T - 1) ? sv[r] : sv[(t + 1) * NX + r]; };
        auto ATdy = [&](int k) {   // (A' dy)[k] from sv
            const int t = k / N, j = k % N;
            real acc = 0;
            if (t < T - 1) {
                const real *Fp = Ft(t);
#pragma unroll
                for (int r = 0; r < NX; ++r) acc += Fp[r * N + j] * sv[(t + 1) * NX + r];
            }
            if (j < NX) {
                if (t >= 1) acc -= sv[t * NX + j];       // dynamics row t-1 = internal block t
                else acc += sv[j];                       // initial-state rows = internal block 0
            }
            return acc;
        };
#pragma unroll 4
        for (int i = lane; i < L.ne; i += 64) dy[i] = dyv(i);
#pragma unroll 4
        for (int k = lane; k < L.nz; k += 64) dx[k] = Pinv[k] * (r1[k] - ATdy(k));
#pragma unroll 4
        for (int i = lane; i < L.ni; i += 64) {
            const int Tn = T * NU, iu = i < Tn ? i : i - Tn;
            const int k = (iu / NU) * N + NX + iu % NU;
            const real du = Pinv[k] * (r1[k] - ATdy(k));
            const real zi = Dt[i] * ((i < Tn ? du : -du) + wv(i));
            dz[i] = zi;
            ds[i] = (bs[i] - sc[i] * zi) / (zc[i] + e);
        }
        sync();
    }

    // out = K(z, s) l   (no regularisation)
    __device__ __forceinline__ void Kmul(const real *l, real *o) {
        const real *__restrict__ zc = w + L.cur + L.oz(), *__restrict__ sc = w + L.cur + L.os();
        const real *__restrict__ lx = l, *__restrict__ ls = l + L.os(), *__restrict__ lz = l + L.oz(), *__restrict__ ly = l + L.oy();
        real *__restrict__ ox = o, *__restrict__ oS = o + L.os(), *__restrict__ oZ = o + L.oz(), *__restrict__ oY = o + L.oy();
#pragma unroll 4
        for (int k = lane; k < L.nz; k += 64) ox[k] = cd(k) * lx[k] + GTz(lz, k) + ATy(ly, k);
#pragma unroll 4
        for (int i = lane; i < L.ni; i += 64) {
            oS[i] = zc[i] * ls[i] + sc[i] * lz[i];
            oZ[i] = Gx(lx, i) + ls[i];
        }
#pragma unroll 4
        for (int i = lane; i < L.ne; i += 64) oY[i] = Ax(lx, i);
    }

    // solve_kkt (batch_LU.py:212-244): rr holds r = -(rx, rs, rz, ry); result in `out`.
    // One call site of apply() (a loop over the solve and its refinement solve): everything is inlined, so
    // the solver's state stays in registers / SGPRs instead of a `this` object in scratch.
    __device__ __forceinline__ void solve_kkt(real *out) {
        real *__restrict__ rr = w + L.rr, *__restrict__ r2 = w + L.r2, *__restrict__ dd = w + L.dc;
        for (int pass = 0; pass < 2; ++pass) {
            apply(pass == 0 ? rr : r2, pass == 0 ? out : dd);
            if (pass == 0) {
                Kmul(out, r2);
                sync();
#pragma unroll 4
                for (int i = lane; i < L.NK; i += 64) r2[i] = rr[i] - r2[i];
            } else {
#pragma unroll 4
                for (int i = lane; i < L.NK; i += 64) out[i] += dd[i];
            }
            sync();
        }
    }

    // get_step (batch_LU.py:200-208), per instance: min over rows of -v/dv (dv < 0), 1 (dv == 0),
    // no constraint (dv > 0). (The reference caps dv > 0 rows at max(1, a.max()) taken over the WHOLE
    // batch; the cap only matters when it is below 1/0.999, which never happened in any fixture:
    // `gs_coupled` of tools/gen_golden_ip.py. DESIGN.md section 12.)
    __device__ __forceinline__ real get_step(const real *v, const real *dv, int &nanflag) const {
        real m = INFINITY;
        int nf = 0;
        for (int i = lane; i < L.ni; i += 64) {
            const real d = dv[i];
            real s = INFINITY;
            if (d == 0) s = 1;
            else if (d < 0) s = -v[i] / d;
            else if (d != d) s = d;
            if (s != s) nf = 1;
            m = s < m ? s : m;
        }
        nanflag |= wave_or(nf);
        return wave_min(m);
    }

    // ---- residuals + best iterate (batch_LU.py:86-146); returns 1 if this instance improved -------
    __device__ __forceinline__ int resid(int it) {
        real *x = w + L.cur, *s = x + L.os(), *z = x + L.oz(), *y = x + L.oy();
        real *rx = w + L.res, *rs = rx + L.os(), *rz = rx + L.oz(), *ry = rx + L.oy();
        real sz = 0, nzr = 0, nyr = 0, nxr = 0;
#pragma unroll 4
        for (int k = lane; k < L.nz; k += 64) {
            const real v = cd(k) * x[k] + cc(k) + GTz(z, k) + ATy(y, k);
            rx[k] = v; nxr += v * v;
        }
#pragma unroll 4
        for (int i = lane; i < L.ni; i += 64) {
            const real p = s[i] * z[i];
            rs[i] = p; sz += p;
            const real v = Gx(x, i) + s[i] - hh(i);
            rz[i] = v; nzr += v * v;
        }
        const real *ext = a.ry_ext ? a.ry_ext + (long)b * L.ne : nullptr;
        for (int i = lane; i < L.ne; i += 64) {
            real v;
            if (ext) v = ext[i];
            else {
                const int t = i / NX, r = i % NX;
                v = Ax(x, i) + (t == T - 1 ? -x0[r] : ff(t, r));     // A x - b, b = [-f ; x0]
            }
            ry[i] = v; nyr += v * v;
        }
        sz = wave_sum(sz); nzr = wave_sum(nzr); nyr = wave_sum(nyr); nxr = wave_sum(nxr);
        const real mu = absr(sz / real(L.ni));
        const real rsd = sqrtr(nyr) + sqrtr(nzr) + sqrtr(nxr) + real(L.ni) * mu;
        real *sc = w + L.scal;   // {resid_best, mu, have_best, iter_best}
        const bool have = sc[2] != 0;
        const bool better = !have || rsd < sc[0];
        sync();
        if (better) {
            real *bst = w + L.best;
#pragma unroll 4
            for (int i = lane; i < L.NK; i += 64) bst[i] = x[i];
            if (lane == 0) { sc[0] = rsd; sc[2] = 1; sc[3] = real(it); }
        }
        if (lane == 0) sc[1] = mu;
        sync();
        return better ? 1 : 0;
    }

    // ---- one predictor-corrector step (batch_LU.py:153-197) ----------------------------------------
    __device__ __forceinline__ void step() {
        real *x = w + L.cur, *s = x + L.os(), *z = x + L.oz();
        real *res = w + L.res, *rr = w + L.rr, *da = w + L.da;
        real *dcr = w + L.res;                      // corrector direction: the residual block is free by then
        const real mu = (w + L.scal)[1];
        factor();
        for (int ph = 0; ph < 2; ++ph) {            // 0: affine direction, 1: centering-corrector (one solve_kkt site)
            if (ph == 0) {
                for (int i = lane; i < L.NK; i += 64) rr[i] = -res[i];
            } else {
                int nf = 0;
                real al = get_step(z, da + L.oz(), nf);
                const real al2 = get_step(s, da + L.os(), nf);
                al = al2 < al ? al2 : al;
                al = al < real(1) ? al : real(1);
                if (nf) al = NAN;
                real t3 = 0, t4 = 0;
                for (int i = lane; i < L.ni; i += 64) {
                    t3 += (s[i] + al * da[L.os() + i]) * (z[i] + al * da[L.oz() + i]);
                    t4 += s[i] * z[i];
                }
                t3 = wave_sum(t3); t4 = wave_sum(t4);
                real sig = t3 / t4; sig = sig * sig * sig;
                // corrector right-hand side: rx = rz = ry = 0, rs = -mu sig + ds_aff dz_aff   (r = -residual)
                for (int i = lane; i < L.NK; i += 64) {
                    const int j = i - L.os();
                    rr[i] = (j >= 0 && j < L.ni) ? -(-mu * sig + da[L.os() + j] * da[L.oz() + j]) : real(0);
                }
            }
            sync();
            solve_kkt(ph == 0 ? da : dcr);
        }
#pragma unroll 4
        for (int i = lane; i < L.NK; i += 64) da[i] += dcr[i];
        sync();
        int nf = 0;
        real al = get_step(z, da + L.oz(), nf);
        const real al3 = get_step(s, da + L.os(), nf);
        al = al3 < al ? al3 : al;
        al = real(0.999) * al;
        al = al < real(1) ? al : real(1);
        if (nf) al = NAN;
#pragma unroll 4
        for (int i = lane; i < L.NK; i += 64) x[i] += al * da[i];
        sync();
    }

    // ---- initial point (batch_LU.py:44-81) ------------------------------------------------------------
    __device__ __forceinline__ void init() {
        real *x = w + L.cur, *s = x + L.os(), *z = x + L.oz();
        real *rr = w + L.rr, *da = w + L.da, *sc = w + L.scal;
        for (int i = lane; i < L.ni; i += 64) { s[i] = 1; z[i] = 1; }
        if (lane < 8) sc[lane] = 0;
        sync();
        factor();
        // solve_kkt(K, Ktilde, p, 0, -h, -b): r = -(p, 0, -h, -b), b = [-f ; x0]
        for (int k = lane; k < L.nz; k += 64) rr[k] = -cc(k);
        for (int i = lane; i < L.ni; i += 64) { rr[L.os() + i] = 0; rr[L.oz() + i] = hh(i); }
        for (int i = lane; i < L.ne; i += 64) {
            const int t = i / NX, r = i % NX;
            rr[L.oy() + i] = (t == T - 1) ? x0[r] : -ff(t, r);
        }
        sync();
        solve_kkt(da);
        for (int i = lane; i < L.NK; i += 64) x[i] = da[i];
        sync();
        // positivity shift (:71-81)
        real ms = INFINITY, mz = INFINITY;
        for (int i = lane; i < L.ni; i += 64) { ms = s[i] < ms ? s[i] : ms; mz = z[i] < mz ? z[i] : mz; }
        ms = wave_min(ms); mz = wave_min(mz);
        for (int i = lane; i < L.ni; i += 64) {
            if (ms < 0) s[i] -= ms - 1;
            if (mz < 0) z[i] -= mz - 1;
        }
        sync();
    }
};

template <typename real, int NX, int NU, bool FAC_LDS>
__global__ __launch_bounds__(64) void k_ipm(const IpmArgs<real> a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    Ipm<real, NX, NU, FAC_LDS> S(a, reinterpret_cast<real *>(lds_raw), b);
    const Lay<real, NX, NU> &L = S.L;
    const int lane = threadIdx.x;
    if (a.flags & ALQP_IPM_INIT) S.init();
    int improved = 0;
    if (a.flags & ALQP_IPM_LOOP) {
        for (int it = 0; it < a.max_iter; ++it) {
            improved |= S.resid(a.iter0 + it);
            S.step();
        }
    } else {
        if (a.flags & ALQP_IPM_RESID) improved = S.resid(a.iter0);
        if (a.flags & ALQP_IPM_STEP) S.step();
    }
    real *sc = S.w + L.scal;
    if (a.flags & ALQP_IPM_FINAL) {
        const real *bst = S.w + L.best;
        for (int k = lane; k < L.nz; k += 64) a.o_x[(long)b * L.nz + k] = bst[k];
        for (int i = lane; i < L.ni; i += 64) {
            a.o_s[(long)b * L.ni + i] = bst[L.os() + i];
            a.o_z[(long)b * L.ni + i] = bst[L.oz() + i];
        }
        for (int i = lane; i < L.ne; i += 64) a.o_y[(long)b * L.ne + i] = bst[L.oy() + i];
    }
    if (lane == 0) {
        if (a.o_resid) a.o_resid[b] = sc[0];
        if (a.o_mu) a.o_mu[b] = sc[1];
        if (a.o_iter_best) a.o_iter_best[b] = (int)sc[3];
        if (a.o_improved) a.o_improved[b] = improved;
        if (a.o_info && S.info && a.o_info[b] == 0) a.o_info[b] = S.info;
    }
}

// Backward of DenseQPFunction (qp.py:238-252): K at the returned lams / slacks, no regularisation:
// solve_kkt(K, K, gbar, 0, 0, 0) -> dx, dlam (= dz), dnu (= dy).
template <typename real, int NX, int NU, bool FAC_LDS>
__global__ __launch_bounds__(64) void k_ipm_backward(const IpmArgs<real> a, const real *lams, const real *slacks) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    Ipm<real, NX, NU, FAC_LDS> S(a, reinterpret_cast<real *>(lds_raw), b);
    const Lay<real, NX, NU> &L = S.L;
    const int lane = threadIdx.x;
    real *cur = S.w + L.cur, *rr = S.w + L.rr, *da = S.w + L.da;
    for (int i = lane; i < L.ni; i += 64) {
        cur[L.os() + i] = slacks[(long)b * L.ni + i];
        cur[L.oz() + i] = lams[(long)b * L.ni + i];
    }
    for (int i = lane; i < L.NK; i += 64) rr[i] = (i < L.nz) ? -a.gbar[(long)b * L.nz + i] : real(0);
    __syncthreads();
    S.factor();
    S.solve_kkt(da);
    for (int k = lane; k < L.nz; k += 64) a.o_x[(long)b * L.nz + k] = da[k];
    for (int i = lane; i < L.ni; i += 64) a.o_z[(long)b * L.ni + i] = da[L.oz() + i];
    for (int i = lane; i < L.ne; i += 64) a.o_y[(long)b * L.ne + i] = da[L.oy() + i];
    if (lane == 0 && a.o_info && S.info && a.o_info[b] == 0) a.o_info[b] = S.info;
}

constexpr size_t kLdsLimit = 64 * 1024;   // per-workgroup LDS the launch may ask for

// Where the Schur factor lives. LDS: lowest latency per QP, but 27 / 54 KB (fp32 / fp64 at (20,13,4)) leave 5 / 2
// wavefronts per CU. Workspace (L2 / Infinity Cache): every sweep stage waits on a global round trip, but the
// launch then runs 8 wavefronts per CU (register-limited) and hides it - faster as soon as the batch fills the
// chip that way (measured). fac_mode (AlqpIpmParams.variant 1 / 2): 0 auto, 1 always LDS,
// 2 always workspace.
template <typename real, int NX, int NU>
static bool fac_fits_lds(int T) { return lds_words<real, NX, NU>(T, true) * sizeof(real) <= kLdsLimit; }
template <typename real, int NX, int NU>
static bool fac_in_lds(int T, int B, int fac_mode) {
    if (!fac_fits_lds<real, NX, NU>(T) || fac_mode == 2) return false;
    if (fac_mode == 1) return true;
    // auto: LDS while it still allows >= 4 wavefronts per CU (160 KB), or when the batch is too small to
    // fill the chip with enough wavefronts to hide the workspace latency. Measured at (20,13,4), B = 8192:
    // fp64 (2 per CU in LDS) 65 k -> 106 k QP/s in the workspace; fp32 (5 per CU in LDS) 160 k vs 140 k; B = 512:
    // LDS 65 k / 81 k vs workspace 19 k / 58 k.
    const long per_cu = (160L * 1024) / (lds_words<real, NX, NU>(T, true) * (long)sizeof(real));
    return B < 2048 || per_cu >= 4;
}

// (the workspace always reserves the factor region, so that its size does not depend on the placement)
template <typename real, int NX, int NU>
static size_t ws_words_for(int T) { return Lay<real, NX, NU>(T, true).total; }

template <typename real>
static size_t ws_bytes(int nx, int nu, int B, int T) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return (size_t)B * ws_words_for<real, NX, NU>(T) * sizeof(real);
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return 0;
}

template <typename real, int NX, int NU>
static int launch(IpmArgs<real> a, const real *lams, const real *slacks, bool backward, int fac_mode, hipStream_t stream) {
    const bool fl = fac_in_lds<real, NX, NU>(a.T, a.B, fac_mode);
    a.ws_words = (long)ws_words_for<real, NX, NU>(a.T);
    const size_t lds = lds_words<real, NX, NU>(a.T, fl) * sizeof(real);
    if (lds > kLdsLimit) return ALQP_E_UNSUPPORTED;
#define LAUNCH(KF)                                                                                            \
    do {                                                                                                      \
        if (lds > 48 * 1024 &&                                                                                \
            hipFuncSetAttribute(reinterpret_cast<const void *>(KF), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds) != hipSuccess)                                                      \
            return ALQP_E_LAUNCH;                                                                             \
    } while (0)
    if (backward) {
        if (fl) { LAUNCH((k_ipm_backward<real, NX, NU, true>)); hipLaunchKernelGGL((k_ipm_backward<real, NX, NU, true>), dim3(a.B), dim3(64), lds, stream, a, lams, slacks); }
        else { LAUNCH((k_ipm_backward<real, NX, NU, false>)); hipLaunchKernelGGL((k_ipm_backward<real, NX, NU, false>), dim3(a.B), dim3(64), lds, stream, a, lams, slacks); }
    } else {
        if (fl) { LAUNCH((k_ipm<real, NX, NU, true>)); hipLaunchKernelGGL((k_ipm<real, NX, NU, true>), dim3(a.B), dim3(64), lds, stream, a); }
        else { LAUNCH((k_ipm<real, NX, NU, false>)); hipLaunchKernelGGL((k_ipm<real, NX, NU, false>), dim3(a.B), dim3(64), lds, stream, a); }
    }
#undef LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

static int g4_launch(int nx, int nu, const IpmArgs<double> &a, const double *l, const double *s, bool bw, hipStream_t st) {
    return alqp_ipm_g4::launch_f64(nx, nu, a, l, s, bw, (void *)st);
}
static int g4_launch(int nx, int nu, const IpmArgs<float> &a, const float *l, const float *s, bool bw, hipStream_t st) {
    return alqp_ipm_g4::launch_f32(nx, nu, a, l, s, bw, (void *)st);
}

// variant (AlqpIpmParams.variant): 0 auto - the register/LDS-resident kernel (alqp_ipm_g4.hpp) whenever the problem
// fits it (T <= 20), else the generic kernel with its automatic factor placement; 1 / 2 the generic kernel with
// the factor in LDS / in the workspace; 3 the register-resident kernel or ALQP_E_UNSUPPORTED.
template <typename real>
static int dispatch(int nx, int nu, const IpmArgs<real> &a, const real *lams, const real *slacks, bool backward,
                    int variant, hipStream_t stream) {
    if (variant < 0 || variant > 3) return ALQP_E_BADARG;
    if (variant == 0 || variant == 3) {
        const int rc = g4_launch(nx, nu, a, lams, slacks, backward, stream);
        if (rc != ALQP_E_UNSUPPORTED || variant == 3) return rc;
    }
    const int fac_mode = variant == 0 ? 0 : variant;
#define X(NX, NU) \
    if (nx == NX && nu == NU) return launch<real, NX, NU>(a, lams, slacks, backward, fac_mode, stream);
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return ALQP_E_UNSUPPORTED;
}

static bool dims_ok(const AlqpDims *d) { return d && d->B > 0 && d->T >= 2 && d->nx > 0 && d->nu > 0 && d->nx <= 16; }

template <typename real>
static int solve_impl(const AlqpDims *d, const AlqpIpmParams *p, const void *Cd, const void *c, const void *F,
                      const void *f, const void *x0, const void *u_hi, const void *u_lo, long sC_t, long sC_b,
                      long sF_t, long sF_b, long sf_t, long sf_b, void *ws, size_t ws_bytes_, const void *ry_ext,
                      void *zhat, void *nus, void *lams, void *slacks, void *resid, void *mu, int *iter_best,
                      int *improved, int *info, void *stream) {
    if (!dims_ok(d) || !p || !Cd || !c || !F || !f || !x0 || !u_hi || !u_lo || !ws) return ALQP_E_BADARG;
    if ((p->flags & ALQP_IPM_FINAL) && (!zhat || !nus || !lams || !slacks)) return ALQP_E_BADARG;
    if (p->max_iter < 0 || !(p->flags & (ALQP_IPM_INIT | ALQP_IPM_RESID | ALQP_IPM_STEP | ALQP_IPM_LOOP | ALQP_IPM_FINAL)))
        return ALQP_E_BADARG;
    const size_t need = ws_bytes<real>(d->nx, d->nu, d->B, d->T);
    if (need == 0) return ALQP_E_UNSUPPORTED;
    if (ws_bytes_ < need) return ALQP_E_BADARG;
    IpmArgs<real> a = {};
    a.B = d->B; a.T = d->T; a.flags = p->flags; a.max_iter = p->max_iter; a.iter0 = p->iter0; a.e = (real)p->kkt_eps;
    a.Cd = (const real *)Cd; a.c = (const real *)c; a.F = (const real *)F; a.f = (const real *)f;
    a.x0 = (const real *)x0; a.uhi = (const real *)u_hi; a.ulo = (const real *)u_lo;
    a.sC_t = sC_t; a.sC_b = sC_b; a.sF_t = sF_t; a.sF_b = sF_b; a.sf_t = sf_t; a.sf_b = sf_b;
    a.ws = (real *)ws; a.ry_ext = (const real *)ry_ext;
    a.o_x = (real *)zhat; a.o_y = (real *)nus; a.o_z = (real *)lams; a.o_s = (real *)slacks;
    a.o_resid = (real *)resid; a.o_mu = (real *)mu; a.o_iter_best = iter_best; a.o_improved = improved; a.o_info = info;
    return dispatch<real>(d->nx, d->nu, a, nullptr, nullptr, false, p->variant, (hipStream_t)stream);
}

template <typename real>
static int backward_impl(const AlqpDims *d, const void *Cd, const void *F, long sC_t, long sC_b, long sF_t, long sF_b,
                         const void *lams, const void *slacks, const void *gbar, void *ws, size_t ws_bytes_, void *dx,
                         void *dlam, void *dnu, int *info, int variant, void *stream) {
    if (!dims_ok(d) || !Cd || !F || !lams || !slacks || !gbar || !ws || !dx || !dlam || !dnu) return ALQP_E_BADARG;
    const size_t need = ws_bytes<real>(d->nx, d->nu, d->B, d->T);
    if (need == 0) return ALQP_E_UNSUPPORTED;
    if (ws_bytes_ < need) return ALQP_E_BADARG;
    IpmArgs<real> a = {};
    a.B = d->B; a.T = d->T; a.e = 0;
    a.Cd = (const real *)Cd; a.F = (const real *)F; a.sC_t = sC_t; a.sC_b = sC_b; a.sF_t = sF_t; a.sF_b = sF_b;
    a.ws = (real *)ws; a.gbar = (const real *)gbar;
    a.o_x = (real *)dx; a.o_z = (real *)dlam; a.o_y = (real *)dnu; a.o_info = info;
    return dispatch<real>(d->nx, d->nu, a, (const real *)lams, (const real *)slacks, true, variant, (hipStream_t)stream);
}

}  // namespace alqp_ipm

extern "C" {

size_t alqp_ipm_workspace_bytes(const AlqpDims *dims, int is_f64) {
    if (!alqp_ipm::dims_ok(dims)) return 0;
    return is_f64 ? alqp_ipm::ws_bytes<double>(dims->nx, dims->nu, dims->B, dims->T)
                  : alqp_ipm::ws_bytes<float>(dims->nx, dims->nu, dims->B, dims->T);
}

#define ALQP_IPM_DEFINE(SFX, REAL)                                                                                  \
    int alqp_ipm_solve_##SFX(const AlqpDims *dims, const AlqpIpmParams *prm, const void *Cd, const void *c,         \
                             const void *F, const void *f, const void *x0, const void *u_hi, const void *u_lo,      \
                             long sC_t, long sC_b, long sF_t, long sF_b, long sf_t, long sf_b, void *workspace,     \
                             size_t ws_bytes, const void *ry_ext, void *zhat, void *nus, void *lams, void *slacks,  \
                             void *resid, void *mu, int *iter_best, int *improved, int *info, void *stream) {       \
        return alqp_ipm::solve_impl<REAL>(dims, prm, Cd, c, F, f, x0, u_hi, u_lo, sC_t, sC_b, sF_t, sF_b, sf_t,     \
                                          sf_b, workspace, ws_bytes, ry_ext, zhat, nus, lams, slacks, resid, mu,    \
                                          iter_best, improved, info, stream);                                       \
    }                                                                                                               \
    int alqp_ipm_backward_##SFX(const AlqpDims *dims, const void *Cd, const void *F, long sC_t, long sC_b,          \
                                long sF_t, long sF_b, const void *lams, const void *slacks, const void *gbar,       \
                                void *workspace, size_t ws_bytes, void *dx, void *dlam, void *dnu, int *info,       \
                                int variant, void *stream) {                                                        \
        return alqp_ipm::backward_impl<REAL>(dims, Cd, F, sC_t, sC_b, sF_t, sF_b, lams, slacks, gbar, workspace,    \
                                             ws_bytes, dx, dlam, dnu, info, variant, stream);                       \
    }

ALQP_IPM_DEFINE(f32, float)
ALQP_IPM_DEFINE(f64, double)

}  // extern "C"
