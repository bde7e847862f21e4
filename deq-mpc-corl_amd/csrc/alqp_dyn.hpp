// alqp_dyn.hpp - dynamics models that can be inlined into the solver kernels (the nonlinear
// fused solve, alqp_solve_nonlin): one RK4 step x+ = f(x, u) of the reference's generated
// per-robot packages, evaluated with dual numbers so that the same code gives the value
// (NT = 0: line-search candidates, true residuals) and the Jacobian (NT = n tangents).
//   pendulum1l : deqmpc/my_envs/pendulum1l/src/generated_dynamics.c:55-140
//   cartpole1l : deqmpc/my_envs/cartpole1l/src/generated_dynamics.c (model: DESIGN.md section 9)
// The same models back the provider kernels (alqp_dyn_*), parity-tested against the compiled
// reference code (tests/test_dynamics_provider.py).
#pragma once
#include <hip/hip_runtime.h>

#include "alqp_team.hpp"  // fma_

namespace alqp {

template <typename real, int NT>
struct Dual {
    real v;
    real d[NT > 0 ? NT : 1];
};
template <typename real, int NT>
__device__ __forceinline__ Dual<real, NT> dconst(real v) {
    Dual<real, NT> r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = 0;
    return r;
}
template <typename real, int NT>
__device__ __forceinline__ Dual<real, NT> daxpy(Dual<real, NT> a, real s, Dual<real, NT> b) {  // a + s b
    Dual<real, NT> r;
    r.v = fma_(s, b.v, a.v);
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = fma_(s, b.d[i], a.d[i]);
    return r;
}
template <typename real, int NT>
__device__ __forceinline__ Dual<real, NT> dmul(Dual<real, NT> a, Dual<real, NT> b) {
    Dual<real, NT> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = fma_(a.d[i], b.v, a.v * b.d[i]);
    return r;
}
template <typename real, int NT>
__device__ __forceinline__ void dsincos(Dual<real, NT> a, Dual<real, NT> &sn, Dual<real, NT> &cs) {
    const real s = sin(a.v), c = cos(a.v);
    sn.v = s;
    cs.v = c;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        sn.d[i] = c * a.d[i];
        cs.d[i] = -s * a.d[i];
    }
}

// ---- pendulum1l: x = (theta, omega), u = tau; theta'' = 4 tau - 19.62 sin(theta) ----------------
template <typename real>
struct DynPendulum1l {
    static constexpr int NX = 2, NU = 1, ID = 1;
    template <int NT>
    __device__ __forceinline__ static Dual<real, NT> acc(Dual<real, NT> th, Dual<real, NT> ta) {
        Dual<real, NT> sn, cs;
        dsincos(th, sn, cs);
        return daxpy(daxpy(dconst<real, NT>(0), real(4), ta), real(-19.62), sn);
    }
    template <int NT>
    __device__ __forceinline__ static void step(const Dual<real, NT> (&z)[3], real h, Dual<real, NT> (&xn)[2]) {
        using D = Dual<real, NT>;
        const D th = z[0], om = z[1], ta = z[2];
        const real hh = real(0.5) * h, two = real(2), h6 = h / real(6);
        const D k1o = acc(th, ta);
        const D om2 = daxpy(om, hh, k1o), k2o = acc(daxpy(th, hh, om), ta);
        const D om3 = daxpy(om, hh, k2o), k3o = acc(daxpy(th, hh, om2), ta);
        const D om4 = daxpy(om, h, k3o), k4o = acc(daxpy(th, h, om3), ta);
        xn[0] = daxpy(th, h6, daxpy(daxpy(om, two, om2), real(1), daxpy(om4, two, om3)));
        xn[1] = daxpy(om, h6, daxpy(daxpy(k1o, two, k2o), real(1), daxpy(k4o, two, k3o)));
    }
};

// ---- cartpole1l: x = (cart x, theta, xdot, thetadot), u = force on the cart (tau = (u, 0)) ----
//      M(th) q'' = tau - (mb sin th th'^2, 0) + (0, 9.81 mb sin th), M = [[ma, -mb cos th], [-mb cos th, md]].
//      V = 1: my_envs/cartpole1l (ma, mb, md) = (11, 1, 2); V = 2: my_envs/cartpole1l_v2 (0.7, 0.1, 0.05)
template <typename real, int V = 1>
struct DynCartpole1l {
    static constexpr int NX = 4, NU = 1, ID = V == 1 ? 2 : 4;
    static constexpr double MA = V == 1 ? 11.0 : 0.7, MB = V == 1 ? 1.0 : 0.1, MD = V == 1 ? 2.0 : 0.05;
    template <int NT>
    __device__ __forceinline__ static void acc(Dual<real, NT> th, Dual<real, NT> thd, Dual<real, NT> t0, Dual<real, NT> &xdd,
                                               Dual<real, NT> &thdd) {
        using D = Dual<real, NT>;
        D sn, cs;
        dsincos(th, sn, cs);
        const D r0 = daxpy(t0, real(-MB), dmul(sn, dmul(thd, thd)));  // tau0 - mb sin(th) thd^2
        const D r1 = daxpy(dconst<real, NT>(0), real(9.81 * MB), sn);  // 9.81 mb sin(th)
        const D det = daxpy(dconst<real, NT>(real(MA * MD)), real(-MB * MB), dmul(cs, cs));
        D idet;
        idet.v = real(1) / det.v;
#pragma unroll
        for (int i = 0; i < NT; ++i) idet.d[i] = -det.d[i] * idet.v * idet.v;
        const D zero = dconst<real, NT>(0);
        xdd = dmul(idet, daxpy(daxpy(zero, real(MB), dmul(cs, r1)), real(MD), r0));    // M^-1 = [[md, mb c], [mb c, ma]] / det
        thdd = dmul(idet, daxpy(daxpy(zero, real(MB), dmul(cs, r0)), real(MA), r1));
    }
    template <int NT>
    __device__ __forceinline__ static void step(const Dual<real, NT> (&z)[5], real h, Dual<real, NT> (&xn)[4]) {
        using D = Dual<real, NT>;
        const D px = z[0], th = z[1], xd = z[2], thd = z[3], t0 = z[4];
        const real hh = real(0.5) * h, two = real(2), h6 = h / real(6);
        D k1xd, k1td, k2xd, k2td, k3xd, k3td, k4xd, k4td;
        acc(th, thd, t0, k1xd, k1td);
        const D k2x = daxpy(xd, hh, k1xd), k2t = daxpy(thd, hh, k1td);
        acc(daxpy(th, hh, thd), k2t, t0, k2xd, k2td);
        const D k3x = daxpy(xd, hh, k2xd), k3t = daxpy(thd, hh, k2td);
        acc(daxpy(th, hh, k2t), k3t, t0, k3xd, k3td);
        const D k4x = daxpy(xd, h, k3xd), k4t = daxpy(thd, h, k3td);
        acc(daxpy(th, h, k3t), k4t, t0, k4xd, k4td);
        xn[0] = daxpy(px, h6, daxpy(daxpy(xd, two, k2x), real(1), daxpy(k4x, two, k3x)));
        xn[1] = daxpy(th, h6, daxpy(daxpy(thd, two, k2t), real(1), daxpy(k4t, two, k3t)));
        xn[2] = daxpy(xd, h6, daxpy(daxpy(k1xd, two, k2xd), real(1), daxpy(k4xd, two, k3xd)));
        xn[3] = daxpy(thd, h6, daxpy(daxpy(k1td, two, k2td), real(1), daxpy(k4td, two, k3td)));
    }
};

// ---- cartpole2l: q = (cart x, th1, th2 relative to link 1), x = (q, q'), u = force on the cart -----
//      M(q) q'' = tau - h(q, q') + G(q)   (model: oracle/dyn_oracle.c)
template <typename real>
struct DynCartpole2l {
    static constexpr int NX = 6, NU = 1, ID = 3;
    template <int NT>
    __device__ __forceinline__ static Dual<real, NT> neg(Dual<real, NT> a) {
        return daxpy(dconst<real, NT>(0), real(-1), a);
    }
    template <int NT>
    __device__ __forceinline__ static Dual<real, NT> lin2(real a, Dual<real, NT> x, real b, Dual<real, NT> y) {  // a x + b y
        return daxpy(daxpy(dconst<real, NT>(0), a, x), b, y);
    }
    template <int NT>
    __device__ __forceinline__ static void acc(const Dual<real, NT> (&q)[3], const Dual<real, NT> (&w)[3],
                                               const Dual<real, NT> (&tau)[3], Dual<real, NT> (&a)[3]) {
        using D = Dual<real, NT>;
        D s1, c1, s2, c2, s12, c12;
        dsincos(q[1], s1, c1);
        dsincos(q[2], s2, c2);
        dsincos(daxpy(q[1], real(1), q[2]), s12, c12);
        const D m11 = dconst<real, NT>(real(12)), m12 = neg(lin2(real(2), c1, real(1), c12)), m13 = neg(c12);
        const D m22 = daxpy(dconst<real, NT>(real(5)), real(2), c2), m23 = daxpy(dconst<real, NT>(real(2)), real(1), c2);
        const D m33 = dconst<real, NT>(real(2));
        const D w12 = daxpy(w[1], real(1), w[2]);
        const D r0 = daxpy(tau[0], real(-1), lin2(real(2), dmul(s1, dmul(w[1], w[1])), real(1), dmul(s12, dmul(w12, w12))));
        const D r1 = daxpy(daxpy(tau[1], real(1), dmul(s2, dmul(w[2], daxpy(w[2], real(2), w[1])))), real(9.81),
                           lin2(real(2), s1, real(1), s12));
        const D r2 = daxpy(daxpy(tau[2], real(-1), dmul(s2, dmul(w[1], w[1]))), real(9.81), s12);
        // symmetric 3x3 solve by cofactors
        const D A00 = daxpy(dmul(m22, m33), real(-1), dmul(m23, m23));
        const D A01 = daxpy(dmul(m13, m23), real(-1), dmul(m12, m33));
        const D A02 = daxpy(dmul(m12, m23), real(-1), dmul(m13, m22));
        const D A11 = daxpy(dmul(m11, m33), real(-1), dmul(m13, m13));
        const D A12 = daxpy(dmul(m12, m13), real(-1), dmul(m11, m23));
        const D A22 = daxpy(dmul(m11, m22), real(-1), dmul(m12, m12));
        const D det = daxpy(daxpy(dmul(m11, A00), real(1), dmul(m12, A01)), real(1), dmul(m13, A02));
        D idet;
        idet.v = real(1) / det.v;
#pragma unroll
        for (int i = 0; i < NT; ++i) idet.d[i] = -det.d[i] * idet.v * idet.v;
        a[0] = dmul(idet, daxpy(daxpy(dmul(A00, r0), real(1), dmul(A01, r1)), real(1), dmul(A02, r2)));
        a[1] = dmul(idet, daxpy(daxpy(dmul(A01, r0), real(1), dmul(A11, r1)), real(1), dmul(A12, r2)));
        a[2] = dmul(idet, daxpy(daxpy(dmul(A02, r0), real(1), dmul(A12, r1)), real(1), dmul(A22, r2)));
    }
    // full interface of the generated package: x[6], tau[3]
    template <int NT>
    __device__ __forceinline__ static void step_full(const Dual<real, NT> (&x)[6], const Dual<real, NT> (&tau)[3], real h,
                                                     Dual<real, NT> (&xn)[6]) {
        using D = Dual<real, NT>;
        const real hh = real(0.5) * h, two = real(2), h6 = h / real(6);
        D q[3], w[3], k1w[3], k2q[3], k2w[3], k3q[3], k3w[3], k4q[3], k4w[3], tq[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            q[i] = x[i];
            w[i] = x[3 + i];
        }
        acc(q, w, tau, k1w);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            tq[i] = daxpy(q[i], hh, w[i]);
            k2q[i] = daxpy(w[i], hh, k1w[i]);
        }
        acc(tq, k2q, tau, k2w);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            tq[i] = daxpy(q[i], hh, k2q[i]);
            k3q[i] = daxpy(w[i], hh, k2w[i]);
        }
        acc(tq, k3q, tau, k3w);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            tq[i] = daxpy(q[i], h, k3q[i]);
            k4q[i] = daxpy(w[i], h, k3w[i]);
        }
        acc(tq, k4q, tau, k4w);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            xn[i] = daxpy(q[i], h6, daxpy(daxpy(w[i], two, k2q[i]), real(1), daxpy(k4q[i], two, k3q[i])));
            xn[3 + i] = daxpy(w[i], h6, daxpy(daxpy(k1w[i], two, k2w[i]), real(1), daxpy(k4w[i], two, k3w[i])));
        }
    }
    // MPC interface: z = (x[6], u), tau = (u, 0, 0)
    template <int NT>
    __device__ __forceinline__ static void step(const Dual<real, NT> (&z)[7], real h, Dual<real, NT> (&xn)[6]) {
        Dual<real, NT> x[6], tau[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = z[i];
        tau[0] = z[6];
        tau[1] = dconst<real, NT>(0);
        tau[2] = dconst<real, NT>(0);
        step_full<NT>(x, tau, h, xn);
    }
};

// value only: xn[NX] = f(z[0..N))
template <typename Dyn, typename real>
__device__ __forceinline__ void dyn_value(const real (&z)[Dyn::NX + Dyn::NU], real h, real (&xn)[Dyn::NX]) {
    constexpr int N = Dyn::NX + Dyn::NU;
    Dual<real, 0> zd[N], xd[Dyn::NX];
#pragma unroll
    for (int j = 0; j < N; ++j) zd[j].v = z[j];
    Dyn::template step<0>(zd, h, xd);
#pragma unroll
    for (int i = 0; i < Dyn::NX; ++i) xn[i] = xd[i].v;
}
// value and Jacobian J[NX][N] = d f / d z
template <typename Dyn, typename real>
__device__ __forceinline__ void dyn_value_jac(const real (&z)[Dyn::NX + Dyn::NU], real h, real (&xn)[Dyn::NX],
                                              real (&J)[Dyn::NX][Dyn::NX + Dyn::NU]) {
    constexpr int N = Dyn::NX + Dyn::NU;
    Dual<real, N> zd[N], xd[Dyn::NX];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        zd[j] = dconst<real, N>(z[j]);
        zd[j].d[j] = 1;
    }
    Dyn::template step<N>(zd, h, xd);
#pragma unroll
    for (int i = 0; i < Dyn::NX; ++i) {
        xn[i] = xd[i].v;
#pragma unroll
        for (int j = 0; j < N; ++j) J[i][j] = xd[i].d[j];
    }
}

// value and THIS LANE's share of the Jacobian: the four lanes of a quad split the n tangents,
// lane q carries columns k = 4 i + q (i < NTL): Jl[r][i] = d f_r / d z_{4i+q} (zero past n)
template <typename Dyn, typename real>
__device__ __forceinline__ void dyn_value_jac_split(const real (&z)[Dyn::NX + Dyn::NU], real h, int q, real (&xn)[Dyn::NX],
                                                    real (&Jl)[Dyn::NX][(Dyn::NX + Dyn::NU + 3) / 4]) {
    constexpr int N = Dyn::NX + Dyn::NU, NTL = (N + 3) / 4;
    Dual<real, NTL> zd[N], xd[Dyn::NX];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        zd[j] = dconst<real, NTL>(z[j]);
        zd[j].d[j >> 2] = ((j & 3) == q) ? real(1) : real(0);
    }
    Dyn::template step<NTL>(zd, h, xd);
#pragma unroll
    for (int i = 0; i < Dyn::NX; ++i) {
        xn[i] = xd[i].v;
#pragma unroll
        for (int j = 0; j < NTL; ++j) Jl[i][j] = xd[i].d[j];
    }
}

}  // namespace alqp
