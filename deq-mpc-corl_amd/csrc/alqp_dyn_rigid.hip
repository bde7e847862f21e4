// alqp_dyn_rigid.hip - dynamics + Jacobian providers of the reference's torch-coded robots:
//   RexQuadrotor     deqmpc/rex_quadrotor.py:98-144      x = (r, MRP m, body velocity v, body rate w) [12], u [4]
//   FlyingCartpole   deqmpc/flying_cartpole2d.py:81-148  x = (r, m, theta, v, w, theta') [14],          u [4]
// One classical RK4 step of length h and its exact Jacobian d x+ / d (x, u), written straight into the solver's
// operands: x+ [K][nx] and F = [A | B] [K][nx][nx+nu] - what the reference obtains by replicating the state nx times
// and calling torch.autograd.grad on a TorchScript RK4 (rex_quadrotor.py:136-144: nx forward + backward passes of
// ~30 small tensor kernels each, per Newton step).
//
// PARITY UNPINNED. Both reference files import rk4, mrp2quat, quatrot, w2pdotkinematics_mrp, ... from a module
// `rexquad_utils` that is NOT in the reference tree, so they cannot run, and the tree holds no
// outputs of them. The three missing helpers are restated from their standard definitions (modified Rodrigues
// parameters, scalar-first unit quaternions):
//   mrp2quat(p)                 = [(1 - |p|^2), 2 p] / (1 + |p|^2)
//   quatrot(q, v)               = v + q0 t + qv x t,  t = 2 qv x v                  (rotation of v by q)
//   w2pdotkinematics_mrp(p, w)  = 1/4 [(1 - |p|^2) w + 2 p x w + 2 (p.w) p]
// Everything else follows the reference's lines (cited at each step). What IS checked (tests/test_dynamics_rigid.py):
// the kernels against a torch restatement of the same equations (oracle/rigid_py.py) in value, the Jacobian against
// autograd of that restatement and against central differences, and physical invariants (hover is an equilibrium,
// rotation by q preserves norms, |m| small-angle limit of the MRP kinematics).
//
// Layout: LPP lanes per point (4-16 in fp32, 16-32 in fp64); each lane carries the value and 1/LPP of the nx + nu tangents as
// dual numbers (lane q: columns LPP i + q), so a point's 16 (18) directional derivatives cost 2-5 tangent slots per
// lane instead of a 17-wide dual number per thread.
#include <hip/hip_runtime.h>

#include "alqp_dyn.hpp"
#include "mi_alqp.h"

namespace alqp_rigid {

using alqp::Dual;

template <typename real>
struct Params {   // AlqpRigidParams in the kernel's precision
    real mass, J[9], Jinv[9], g[3], motor_dist, kf, bf, km, act_scale, u_hover, pend_L, ss[12], bf_force;
};

// ---- dual-number algebra (value + NT tangents) -------------------------------------------------------
template <typename real, int NT>
struct D {
    real v, d[NT > 0 ? NT : 1];
};
#define RIGID_FN template <typename real, int NT> __device__ __forceinline__
RIGID_FN D<real, NT> cst(real v) { D<real, NT> r; r.v = v; for (int i = 0; i < NT; ++i) r.d[i] = 0; return r; }
RIGID_FN D<real, NT> operator+(D<real, NT> a, D<real, NT> b) { D<real, NT> r; r.v = a.v + b.v; for (int i = 0; i < NT; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
RIGID_FN D<real, NT> operator-(D<real, NT> a, D<real, NT> b) { D<real, NT> r; r.v = a.v - b.v; for (int i = 0; i < NT; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
RIGID_FN D<real, NT> operator-(D<real, NT> a) { D<real, NT> r; r.v = -a.v; for (int i = 0; i < NT; ++i) r.d[i] = -a.d[i]; return r; }
RIGID_FN D<real, NT> operator*(D<real, NT> a, D<real, NT> b) { D<real, NT> r; r.v = a.v * b.v; for (int i = 0; i < NT; ++i) r.d[i] = alqp::fma_(a.d[i], b.v, a.v * b.d[i]); return r; }
RIGID_FN D<real, NT> operator*(real s, D<real, NT> a) { D<real, NT> r; r.v = s * a.v; for (int i = 0; i < NT; ++i) r.d[i] = s * a.d[i]; return r; }
RIGID_FN D<real, NT> operator+(D<real, NT> a, real s) { a.v += s; return a; }
RIGID_FN D<real, NT> inv(D<real, NT> a) { D<real, NT> r; r.v = real(1) / a.v; const real m = -r.v * r.v; for (int i = 0; i < NT; ++i) r.d[i] = m * a.d[i]; return r; }
RIGID_FN void sincos_(D<real, NT> a, D<real, NT> &sn, D<real, NT> &cs) {
    const real s = sin(a.v), c = cos(a.v);
    sn.v = s; cs.v = c;
    for (int i = 0; i < NT; ++i) { sn.d[i] = c * a.d[i]; cs.d[i] = -s * a.d[i]; }
}
template <typename real, int NT>
struct V3 {
    D<real, NT> x, y, z;
};
RIGID_FN V3<real, NT> operator+(V3<real, NT> a, V3<real, NT> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RIGID_FN V3<real, NT> operator-(V3<real, NT> a, V3<real, NT> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RIGID_FN V3<real, NT> operator*(D<real, NT> s, V3<real, NT> a) { return {s * a.x, s * a.y, s * a.z}; }
RIGID_FN V3<real, NT> operator*(real s, V3<real, NT> a) { return {s * a.x, s * a.y, s * a.z}; }
RIGID_FN V3<real, NT> cross(V3<real, NT> a, V3<real, NT> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RIGID_FN D<real, NT> dot(V3<real, NT> a, V3<real, NT> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RIGID_FN V3<real, NT> vcst(real x, real y, real z) { return {cst<real, NT>(x), cst<real, NT>(y), cst<real, NT>(z)}; }
// M v for a constant 3x3 matrix (row-major)
RIGID_FN V3<real, NT> matvec(const real *M, V3<real, NT> v) {
    return {M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z};
}
// rotation of v by the unit quaternion (q0, qv)   [rexquad_utils.quatrot: restated, see the header]
RIGID_FN V3<real, NT> quatrot(D<real, NT> q0, V3<real, NT> qv, V3<real, NT> v) {
    const V3<real, NT> t = real(2) * cross(qv, v);
    return v + q0 * t + cross(qv, t);
}

// ---- the rigid body both robots share: time derivative of (r, m, v, w) --------------------------------
// FLY = false: RexQuadrotor_dynamics.dynamics_ (rex_quadrotor.py:113-127), forces (:53-68), moments (:70-85)
// FLY = true : FlyingCartpole_dynamics.dynamics_ (flying_cartpole2d.py:107-130), forces (:51-58), moments (:60-75)
template <typename real, int NT, bool FLY>
__device__ __forceinline__ void body_rates(const Params<real> &P, V3<real, NT> m, V3<real, NT> v, V3<real, NT> w,
                                           const D<real, NT> (&u)[4], V3<real, NT> &pdot, V3<real, NT> &mdot,
                                           V3<real, NT> &vdot, V3<real, NT> &wdot, D<real, NT> &q0, V3<real, NT> &qv) {
    using Dn = D<real, NT>;
    Dn us[4];
    for (int i = 0; i < 4; ++i) us[i] = P.act_scale * (u[i] + P.u_hover);   // u = act_scale * u (:114) / act_scale * (u + u_hover) (:112)
    // q = mrp2quat(m)  [restated]
    const Dn n2 = dot(m, m);
    const Dn id = inv(n2 + real(1));
    q0 = (cst<real, NT>(1) - n2) * id;
    qv = (real(2) * id) * m;
    // forces: thrust along body z + gravity rotated into the body frame by mrp2quat(-m) (+ the motors' bias force, quadrotor)
    const V3<real, NT> qvi = {-qv.x, -qv.y, -qv.z};
    V3<real, NT> F = quatrot(q0, qvi, vcst<real, NT>(P.mass * P.g[0], P.mass * P.g[1], P.mass * P.g[2]));
    F.z = F.z + P.kf * (us[0] + us[1] + us[2] + us[3]) + P.bf_force;
    // moments: yaw torque from the rotor drag, roll / pitch from the arm cross products
    V3<real, NT> tau = {cst<real, NT>(0), cst<real, NT>(0), P.km * (us[0] - us[1] + us[2] - us[3])};
    for (int i = 0; i < 4; ++i) {
        const Dn fz = P.kf * us[i] + (FLY ? real(0) : P.bf);
        // cross(L ss_i, (0, 0, fz)) = L (ss_y fz, -ss_x fz, 0)
        tau.x = tau.x + (P.motor_dist * P.ss[3 * i + 1]) * fz;
        tau.y = tau.y - (P.motor_dist * P.ss[3 * i]) * fz;
    }
    // kinematics and Newton-Euler
    const Dn mw = dot(m, w);
    mdot = real(0.25) * ((cst<real, NT>(1) - n2) * w + real(2) * cross(m, w) + (real(2) * mw) * m);   // w2pdotkinematics_mrp [restated]
    pdot = quatrot(q0, qv, v);                                            // pdot = quatrot(q, v)
    vdot = (real(1) / P.mass) * F - cross(w, v);                          // vdot = F / m - w x v
    wdot = matvec<real, NT>(P.Jinv, tau - cross(w, matvec<real, NT>(P.J, w)));   // wdot = Jinv (tau - w x J w)
}

template <typename real>
struct DynRexQuadrotor {
    static constexpr int NX = 12, NU = 4;
    template <int NT>
    __device__ __forceinline__ static void deriv(const Params<real> &P, const D<real, NT> (&x)[NX], const D<real, NT> (&u)[NU],
                                                 D<real, NT> (&xd)[NX]) {
        V3<real, NT> m = {x[3], x[4], x[5]}, v = {x[6], x[7], x[8]}, w = {x[9], x[10], x[11]}, pd, md, vd, wd, qv;
        D<real, NT> q0;
        body_rates<real, NT, false>(P, m, v, w, u, pd, md, vd, wd, q0, qv);
        xd[0] = pd.x; xd[1] = pd.y; xd[2] = pd.z; xd[3] = md.x; xd[4] = md.y; xd[5] = md.z;
        xd[6] = vd.x; xd[7] = vd.y; xd[8] = vd.z; xd[9] = wd.x; xd[10] = wd.y; xd[11] = wd.z;
    }
};
template <typename real>
struct DynFlyingCartpole {
    static constexpr int NX = 14, NU = 4;
    template <int NT>
    __device__ __forceinline__ static void deriv(const Params<real> &P, const D<real, NT> (&x)[NX], const D<real, NT> (&u)[NU],
                                                 D<real, NT> (&xd)[NX]) {
        // state: [r, m, theta, v, w, theta_dot] (flying_cartpole2d.py:95-105)
        V3<real, NT> m = {x[3], x[4], x[5]}, v = {x[7], x[8], x[9]}, w = {x[10], x[11], x[12]}, pd, md, vd, wd, qv;
        D<real, NT> q0;
        body_rates<real, NT, true>(P, m, v, w, u, pd, md, vd, wd, q0, qv);
        // the inverted pendulum (:124-127): x_ddot = quatrot(q, vdot)[0]; theta_ddot = (g_z sin(theta) + x_ddot cos(theta)) / L
        D<real, NT> sn, cs;
        sincos_(x[6], sn, cs);
        const D<real, NT> xdd = quatrot(q0, qv, vd).x;
        const D<real, NT> thdd = (real(1) / P.pend_L) * (P.g[2] * sn + xdd * cs);
        xd[0] = pd.x; xd[1] = pd.y; xd[2] = pd.z; xd[3] = md.x; xd[4] = md.y; xd[5] = md.z; xd[6] = x[13];
        xd[7] = vd.x; xd[8] = vd.y; xd[9] = vd.z; xd[10] = wd.x; xd[11] = wd.y; xd[12] = wd.z; xd[13] = thdd;
    }
};

// classical RK4 (rex_quadrotor.py:98-107 / flying_cartpole2d.py:79-89), with the weighted sum accumulated as the stages
// come (three state-sized dual arrays live instead of six: the tangents are what fills the register file)
template <typename Dyn, typename real, int NT>
__device__ __forceinline__ void rk4(const Params<real> &P, const D<real, NT> (&x)[Dyn::NX], const D<real, NT> (&u)[Dyn::NU], real h,
                                    D<real, NT> (&xn)[Dyn::NX]) {
    constexpr int NX = Dyn::NX;
    D<real, NT> k[NX], y[NX];
    const real h2 = real(0.5) * h, h6 = h / real(6);
    Dyn::template deriv<NT>(P, x, u, k);
    for (int i = 0; i < NX; ++i) { xn[i] = k[i]; y[i] = x[i] + h2 * k[i]; }
    Dyn::template deriv<NT>(P, y, u, k);
    for (int i = 0; i < NX; ++i) { xn[i] = xn[i] + real(2) * k[i]; y[i] = x[i] + h2 * k[i]; }
    Dyn::template deriv<NT>(P, y, u, k);
    for (int i = 0; i < NX; ++i) { xn[i] = xn[i] + real(2) * k[i]; y[i] = x[i] + h * k[i]; }
    Dyn::template deriv<NT>(P, y, u, k);
    for (int i = 0; i < NX; ++i) xn[i] = x[i] + h6 * (xn[i] + k[i]);
}

// one point per group of LPP lanes: lane q carries the tangents of columns LPP i + q (the value is computed by every
// lane of the group; the tangents are the bulk of the arithmetic). LPP = 4 / 16 in fp32, 16 / 32 in fp64 (quadrotor / flying cartpole; register budget: no scratch).
template <typename Dyn, typename real, int LPP>
__global__ __launch_bounds__(64) void k_dyn_rigid(long K, Params<real> P, const real *x, const real *u, real h, real *xnext, real *J) {
    constexpr int NX = Dyn::NX, NU = Dyn::NU, N = NX + NU, NTL = (N + LPP - 1) / LPP;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long k = tid / LPP;
    const int q = (int)(tid % LPP);
    if (k >= K) return;
    if (J) {
        D<real, NTL> xd[NX], ud[NU], xn[NX];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            D<real, NTL> v = cst<real, NTL>(j < NX ? x[k * NX + j] : u[k * NU + (j - NX)]);
            v.d[j / LPP] = ((j % LPP) == q) ? real(1) : real(0);
            if (j < NX) xd[j] = v; else ud[j - NX] = v;
        }
        rk4<Dyn, real, NTL>(P, xd, ud, h, xn);
        real *Jk = J + k * NX * N;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int s = 0; s < NTL; ++s)
                if (LPP * s + q < N) Jk[i * N + LPP * s + q] = xn[i].d[s];
        }
        if (xnext && q == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xnext[k * NX + i] = xn[i].v;
        }
    } else if (q == 0) {
        D<real, 0> xd[NX], ud[NU], xn[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) xd[j].v = x[k * NX + j];
#pragma unroll
        for (int j = 0; j < NU; ++j) ud[j].v = u[k * NU + j];
        rk4<Dyn, real, 0>(P, xd, ud, h, xn);
#pragma unroll
        for (int i = 0; i < NX; ++i) xnext[k * NX + i] = xn[i].v;
    }
}

template <typename real>
static Params<real> convert(const AlqpRigidParams *p, bool fly) {
    Params<real> P;
    P.mass = (real)p->mass;
    for (int i = 0; i < 9; ++i) { P.J[i] = (real)p->J[i]; P.Jinv[i] = (real)p->Jinv[i]; }
    for (int i = 0; i < 3; ++i) P.g[i] = (real)p->g[i];
    P.motor_dist = (real)p->motor_dist; P.kf = (real)p->kf; P.bf = (real)p->bf; P.km = (real)p->km;
    P.act_scale = (real)p->act_scale; P.u_hover = (real)p->u_hover; P.pend_L = (real)p->pend_L;
    for (int i = 0; i < 12; ++i) P.ss[i] = (real)p->ss[i];
    P.bf_force = fly ? real(0) : (real)p->bf_force;   // Bf = (0, 0, 4 bf) (rex_quadrotor.py:31-32); none in the flying cartpole's forces
    return P;
}

template <typename Dyn, typename real>
static int launch(long K, const AlqpRigidParams *p, const void *x, const void *u, double h, void *xn, void *J, void *stream, bool fly) {
    if (K < 0 || !p || !x || !u || (!xn && !J)) return ALQP_E_BADARG;
    if (K == 0) return 0;
    constexpr int NN = Dyn::NX + Dyn::NU;
    constexpr int LPP = sizeof(real) == 8 ? (NN > 16 ? 32 : 16) : (NN > 16 ? 16 : 4);   // lanes per point: no scratch in any instance
    const long threads = (long)LPP * K;
    hipLaunchKernelGGL((k_dyn_rigid<Dyn, real, LPP>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, (hipStream_t)stream, K,
                       convert<real>(p, fly), (const real *)x, (const real *)u, (real)h, (real *)xn, (real *)J);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

}  // namespace alqp_rigid

extern "C" {
int alqp_dyn_rexquadrotor_f32(long K, const AlqpRigidParams *p, const void *x, const void *u, double h, void *xnext, void *J, void *stream) {
    return alqp_rigid::launch<alqp_rigid::DynRexQuadrotor<float>, float>(K, p, x, u, h, xnext, J, stream, false);
}
int alqp_dyn_rexquadrotor_f64(long K, const AlqpRigidParams *p, const void *x, const void *u, double h, void *xnext, void *J, void *stream) {
    return alqp_rigid::launch<alqp_rigid::DynRexQuadrotor<double>, double>(K, p, x, u, h, xnext, J, stream, false);
}
int alqp_dyn_flyingcartpole_f32(long K, const AlqpRigidParams *p, const void *x, const void *u, double h, void *xnext, void *J, void *stream) {
    return alqp_rigid::launch<alqp_rigid::DynFlyingCartpole<float>, float>(K, p, x, u, h, xnext, J, stream, true);
}
int alqp_dyn_flyingcartpole_f64(long K, const AlqpRigidParams *p, const void *x, const void *u, double h, void *xnext, void *J, void *stream) {
    return alqp_rigid::launch<alqp_rigid::DynFlyingCartpole<double>, double>(K, p, x, u, h, xnext, J, stream, true);
}
}
