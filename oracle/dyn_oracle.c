/* dyn_oracle.c - CPU restatement of the reference's pendulum1l dynamics provider.
 * TEST INFRASTRUCTURE: only tests/ and tools/ may load this (see oracle/oracle_py.py).
 *
 * The reference ships the provider as CasADi-generated straight-line code
 * (deqmpc/my_envs/pendulum1l/src/generated_dynamics.c:55-140 eval_forward_dynamics,
 *  generated_derivatives.c:52-222 eval_forward_derivatives), bound by dynamics.cpp:13-47 as
 *  dynamics(q, qdot, tau, h) -> (q_next, qdot_next) and
 *  derivatives(q, qdot, tau, h) -> (dq'/dq, dq'/dqdot, dq'/dtau, dqdot'/dq, dqdot'/dqdot, dqdot'/dtau).
 * Read as mathematics the generated code is one classical RK4 step of length h of
 *      theta'' = 4 tau - 2 * 9.81 * sin(theta)          (generated_dynamics.c:63-72: a5=4, a8=-2, a9=9.81)
 * i.e. a point-mass pendulum with m l^2 = 1/4 and g/l = 19.62, and the derivatives are the exact
 * Jacobian of that step. This file states it that way (stage derivatives propagated by the chain
 * rule); it is pinned against outputs of the compiled reference code
 * (oracle/_ref/libpendulum1l_casadi.so, tests/golden/dyn_pendulum1l.npz, tools/gen_dyn_golden.py).
 */
#include <math.h>

#define KT 4.0     /* 1 / (m l^2) */
#define KG 19.62   /* 2 * 9.81 */

/* value and tangents (w.r.t. theta0, omega0, tau) of f(theta, omega) = (omega, KT tau - KG sin theta) */
typedef struct { double v, d[3]; } dual;

static dual d_add(dual a, dual b) { dual r; r.v = a.v + b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static dual d_scale(double s, dual a) { dual r; r.v = s * a.v; for (int i = 0; i < 3; ++i) r.d[i] = s * a.d[i]; return r; }
static dual d_acc(dual th, dual tau) {
    dual r;
    const double s = sin(th.v), c = cos(th.v);
    r.v = KT * tau.v - KG * s;
    for (int i = 0; i < 3; ++i) r.d[i] = KT * tau.d[i] - KG * c * th.d[i];
    return r;
}

/* one point: x = (theta, omega), u = tau -> xn[2], A[2][2] = dxn/dx, B[2] = dxn/du */
void dyn_pendulum1l_point(double theta, double omega, double tau, double h, double *xn, double *A, double *B) {
    dual th = {theta, {1, 0, 0}}, om = {omega, {0, 1, 0}}, ta = {tau, {0, 0, 1}};
    /* k1 */
    dual k1t = om, k1o = d_acc(th, ta);
    /* k2 at x + h/2 k1 */
    dual th2 = d_add(th, d_scale(0.5 * h, k1t)), om2 = d_add(om, d_scale(0.5 * h, k1o));
    dual k2t = om2, k2o = d_acc(th2, ta);
    /* k3 at x + h/2 k2 */
    dual th3 = d_add(th, d_scale(0.5 * h, k2t)), om3 = d_add(om, d_scale(0.5 * h, k2o));
    dual k3t = om3, k3o = d_acc(th3, ta);
    /* k4 at x + h k3 */
    dual th4 = d_add(th, d_scale(h, k3t)), om4 = d_add(om, d_scale(h, k3o));
    dual k4t = om4, k4o = d_acc(th4, ta);
    dual st = d_add(d_add(k1t, d_scale(2.0, k2t)), d_add(d_scale(2.0, k3t), k4t));
    dual so = d_add(d_add(k1o, d_scale(2.0, k2o)), d_add(d_scale(2.0, k3o), k4o));
    dual tn = d_add(th, d_scale(h / 6.0, st)), on = d_add(om, d_scale(h / 6.0, so));
    xn[0] = tn.v; xn[1] = on.v;
    A[0] = tn.d[0]; A[1] = tn.d[1]; A[2] = on.d[0]; A[3] = on.d[1];
    B[0] = tn.d[2]; B[1] = on.d[2];
}

/* K points: x[K][2], u[K][1] -> xn[K][2], A[K][2][2], B[K][2][1] */
void dyn_pendulum1l(long K, const double *x, const double *u, double h, double *xn, double *A, double *B) {
    for (long i = 0; i < K; ++i)
        dyn_pendulum1l_point(x[2 * i], x[2 * i + 1], u[i], h, xn + 2 * i, A + 4 * i, B + 2 * i);
}


/* ---- cartpole1l ---------------------------------------------------------------------------
 * deqmpc/my_envs/cartpole1l/src/generated_dynamics.c (eval_forward_dynamics) and
 * generated_derivatives.c, bound like the pendulum (cartpole1l/src/dynamics.cpp:13-47):
 * q = (cart position x, pole angle theta), theta = 0 upright (my_envs/cartpole.py:1-3),
 * tau = (force on the cart, torque on the pole; the environment only drives tau[0],
 * my_envs/dynamics.py:54-56). Read as mathematics the generated code is one RK4 step of
 *      M(theta) q'' = tau - (sin(theta) theta'^2, 0) + (0, 9.81 sin(theta)),
 *      M(theta) = [[11, -cos(theta)], [-cos(theta), 2]]
 * (cart 10 kg + pole 1 kg, m l = 1, pole inertia about the joint 2, no friction; identified from
 * the compiled reference code and pinned against its outputs, tests/golden/dyn_cartpole1l.npz).
 */
typedef struct { double v, d[6]; } dual6;
static dual6 c6(double v) { dual6 r; r.v = v; for (int i = 0; i < 6; ++i) r.d[i] = 0; return r; }
static dual6 a6(dual6 a, dual6 b) { dual6 r; r.v = a.v + b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static dual6 s6(double s, dual6 a) { dual6 r; r.v = s * a.v; for (int i = 0; i < 6; ++i) r.d[i] = s * a.d[i]; return r; }
static dual6 m6(dual6 a, dual6 b) { dual6 r; r.v = a.v * b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
static dual6 sin6(dual6 a) { dual6 r; double c = cos(a.v); r.v = sin(a.v); for (int i = 0; i < 6; ++i) r.d[i] = c * a.d[i]; return r; }
static dual6 cos6(dual6 a) { dual6 r; double s = -sin(a.v); r.v = cos(a.v); for (int i = 0; i < 6; ++i) r.d[i] = s * a.d[i]; return r; }
static dual6 inv6(dual6 a) { dual6 r; r.v = 1.0 / a.v; for (int i = 0; i < 6; ++i) r.d[i] = -a.d[i] * r.v * r.v; return r; }

/* The two cartpole1l packages are the same model with different constants:
 *      M(theta) = [[ma, -mb cos], [-mb cos, md]],  M q'' = tau - (mb sin(theta) theta'^2, 0) + (0, 9.81 mb sin(theta))
 *   cartpole1l    : ma = 11,  mb = 1,   md = 2     (cart 10 kg + pole 1 kg, m l = 1)
 *   cartpole1l_v2 : ma = 0.7, mb = 0.1, md = 0.05  (cart 0.5 kg + pole 0.2 kg at l = 0.5: m l = 0.1, m l^2 = 0.05;
 *                   deqmpc/my_envs/cartpole1l_v2/src/generated_dynamics.c - a package the reference ships but does not
 *                   import, my_envs/cartpole.py:35; identified from its compiled code like the first one and pinned
 *                   against its outputs, tests/golden/dyn_cartpole1l_v2.npz) */
typedef struct { double ma, mb, md; } cart_par;
static const cart_par CART_V1 = {11.0, 1.0, 2.0}, CART_V2 = {0.7, 0.1, 0.05};

/* accelerations (xdd, thdd) at (theta, thetad, tau) */
static void cart_acc(cart_par p, dual6 th, dual6 thd, dual6 t0, dual6 t1, dual6 *xdd, dual6 *thdd) {
    dual6 sn = sin6(th), cs = cos6(th);
    dual6 r0 = a6(t0, s6(-p.mb, m6(sn, m6(thd, thd))));            /* tau0 - mb sin(th) thd^2 */
    dual6 r1 = a6(t1, s6(9.81 * p.mb, sn));                         /* tau1 + 9.81 mb sin(th)  */
    dual6 idet = inv6(a6(c6(p.ma * p.md), s6(-p.mb * p.mb, m6(cs, cs)))); /* det M = ma md - mb^2 cos^2 */
    *xdd = m6(idet, a6(s6(p.md, r0), s6(p.mb, m6(cs, r1))));        /* M^-1 = [[md, mb c],[mb c, ma]] / det */
    *thdd = m6(idet, a6(s6(p.mb, m6(cs, r0)), s6(p.ma, r1)));
}

/* one point: q[2], qd[2], tau[2] -> qn[2], qdn[2], J[4][6] = d(qn, qdn)/d(q, qd, tau) */
static void dyn_cartpole1l_point_p(cart_par p, const double *q, const double *qd, const double *tau, double h, double *xn, double *J) {
    dual6 x = c6(q[0]), th = c6(q[1]), xd = c6(qd[0]), thd = c6(qd[1]), t0 = c6(tau[0]), t1 = c6(tau[1]);
    x.d[0] = 1; th.d[1] = 1; xd.d[2] = 1; thd.d[3] = 1; t0.d[4] = 1; t1.d[5] = 1;
    dual6 k1x = xd, k1t = thd, k1xd, k1td;
    cart_acc(p, th, thd, t0, t1, &k1xd, &k1td);
    dual6 k2x = a6(xd, s6(0.5 * h, k1xd)), k2t = a6(thd, s6(0.5 * h, k1td)), k2xd, k2td;
    cart_acc(p, a6(th, s6(0.5 * h, k1t)), k2t, t0, t1, &k2xd, &k2td);
    dual6 k3x = a6(xd, s6(0.5 * h, k2xd)), k3t = a6(thd, s6(0.5 * h, k2td)), k3xd, k3td;
    cart_acc(p, a6(th, s6(0.5 * h, k2t)), k3t, t0, t1, &k3xd, &k3td);
    dual6 k4x = a6(xd, s6(h, k3xd)), k4t = a6(thd, s6(h, k3td)), k4xd, k4td;
    cart_acc(p, a6(th, s6(h, k3t)), k4t, t0, t1, &k4xd, &k4td);
    dual6 o[4];
    o[0] = a6(x, s6(h / 6.0, a6(a6(k1x, s6(2.0, k2x)), a6(s6(2.0, k3x), k4x))));
    o[1] = a6(th, s6(h / 6.0, a6(a6(k1t, s6(2.0, k2t)), a6(s6(2.0, k3t), k4t))));
    o[2] = a6(xd, s6(h / 6.0, a6(a6(k1xd, s6(2.0, k2xd)), a6(s6(2.0, k3xd), k4xd))));
    o[3] = a6(thd, s6(h / 6.0, a6(a6(k1td, s6(2.0, k2td)), a6(s6(2.0, k3td), k4td))));
    for (int i = 0; i < 4; ++i) {
        xn[i] = o[i].v;
        for (int j = 0; j < 6; ++j) J[6 * i + j] = o[i].d[j];
    }
}

void dyn_cartpole1l_point(const double *q, const double *qd, const double *tau, double h, double *xn, double *J) {
    dyn_cartpole1l_point_p(CART_V1, q, qd, tau, h, xn, J);
}

/* K points: x[K][4] = (q, qd), tau[K][2] -> xn[K][4], J[K][4][6] */
void dyn_cartpole1l(long K, const double *x, const double *tau, double h, double *xn, double *J) {
    for (long i = 0; i < K; ++i) dyn_cartpole1l_point_p(CART_V1, x + 4 * i, x + 4 * i + 2, tau + 2 * i, h, xn + 4 * i, J + 24 * i);
}
void dyn_cartpole1l_v2(long K, const double *x, const double *tau, double h, double *xn, double *J) {
    for (long i = 0; i < K; ++i) dyn_cartpole1l_point_p(CART_V2, x + 4 * i, x + 4 * i + 2, tau + 2 * i, h, xn + 4 * i, J + 24 * i);
}


/* ---- cartpole2l ---------------------------------------------------------------------------
 * deqmpc/my_envs/cartpole2l/src/generated_dynamics.c / generated_derivatives.c (the nx = 6 case
 * of CartpoleDynamics, my_envs/cartpole.py:30-32): q = (cart x, th1, th2), th1 from upright, th2
 * RELATIVE to link 1. Read as mathematics: one RK4 step of  M(q) q'' = tau - h(q, q') + G(q),
 *   M = [[12, -(2 c1 + c12), -c12], [., 5 + 2 c2, 2 + c2], [., ., 2]]     (c1 = cos th1, c12 = cos(th1+th2), c2 = cos th2)
 *   h = (2 s1 w1^2 + s12 (w1 + w2)^2,  -s2 w2 (2 w1 + w2),  s2 w1^2)       (Christoffel terms of M)
 *   G = (0, 9.81 (2 s1 + s12), 9.81 s12)
 * (cart 10 kg, two 1 kg links of length 1 with the masses at the tips... as identified from the
 * compiled reference code: M^-1 from the response to tau at several poses, G from tau = 0, h from the
 * mass matrix; pinned against its outputs, tests/golden/dyn_cartpole2l.npz).
 */
#define NT9 9
typedef struct { double v, d[NT9]; } dual9;
static dual9 c9(double v) { dual9 r; r.v = v; for (int i = 0; i < NT9; ++i) r.d[i] = 0; return r; }
static dual9 a9(dual9 a, dual9 b) { dual9 r; r.v = a.v + b.v; for (int i = 0; i < NT9; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static dual9 s9(double s, dual9 a) { dual9 r; r.v = s * a.v; for (int i = 0; i < NT9; ++i) r.d[i] = s * a.d[i]; return r; }
static dual9 m9(dual9 a, dual9 b) { dual9 r; r.v = a.v * b.v; for (int i = 0; i < NT9; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
static dual9 sin9(dual9 a) { dual9 r; double c = cos(a.v); r.v = sin(a.v); for (int i = 0; i < NT9; ++i) r.d[i] = c * a.d[i]; return r; }
static dual9 cos9(dual9 a) { dual9 r; double s = -sin(a.v); r.v = cos(a.v); for (int i = 0; i < NT9; ++i) r.d[i] = s * a.d[i]; return r; }
static dual9 inv9(dual9 a) { dual9 r; r.v = 1.0 / a.v; for (int i = 0; i < NT9; ++i) r.d[i] = -a.d[i] * r.v * r.v; return r; }

static void cart2_acc(const dual9 *q, const dual9 *w, const dual9 *tau, dual9 *acc) {
    dual9 s1 = sin9(q[1]), c1 = cos9(q[1]), s2 = sin9(q[2]), c2 = cos9(q[2]);
    dual9 t12 = a9(q[1], q[2]), s12 = sin9(t12), c12 = cos9(t12);
    dual9 m11 = c9(12.0), m12 = s9(-1.0, a9(s9(2.0, c1), c12)), m13 = s9(-1.0, c12);
    dual9 m22 = a9(c9(5.0), s9(2.0, c2)), m23 = a9(c9(2.0), c2), m33 = c9(2.0);
    dual9 w12 = a9(w[1], w[2]);
    dual9 r0 = a9(tau[0], s9(-1.0, a9(s9(2.0, m9(s1, m9(w[1], w[1]))), m9(s12, m9(w12, w12)))));
    dual9 r1 = a9(a9(tau[1], m9(s2, m9(w[2], a9(s9(2.0, w[1]), w[2])))), s9(9.81, a9(s9(2.0, s1), s12)));
    dual9 r2 = a9(a9(tau[2], s9(-1.0, m9(s2, m9(w[1], w[1])))), s9(9.81, s12));
    /* symmetric 3x3 solve by cofactors */
    dual9 A00 = a9(m9(m22, m33), s9(-1.0, m9(m23, m23)));
    dual9 A01 = a9(m9(m13, m23), s9(-1.0, m9(m12, m33)));
    dual9 A02 = a9(m9(m12, m23), s9(-1.0, m9(m13, m22)));
    dual9 A11 = a9(m9(m11, m33), s9(-1.0, m9(m13, m13)));
    dual9 A12 = a9(m9(m12, m13), s9(-1.0, m9(m11, m23)));
    dual9 A22 = a9(m9(m11, m22), s9(-1.0, m9(m12, m12)));
    dual9 idet = inv9(a9(a9(m9(m11, A00), m9(m12, A01)), m9(m13, A02)));
    acc[0] = m9(idet, a9(a9(m9(A00, r0), m9(A01, r1)), m9(A02, r2)));
    acc[1] = m9(idet, a9(a9(m9(A01, r0), m9(A11, r1)), m9(A12, r2)));
    acc[2] = m9(idet, a9(a9(m9(A02, r0), m9(A12, r1)), m9(A22, r2)));
}

/* one point: x[6] = (q, qd), tau[3] -> xn[6], J[6][9] = d xn / d(q, qd, tau) */
void dyn_cartpole2l_point(const double *x, const double *tau, double h, double *xn, double *J) {
    dual9 q[3], w[3], ta[3], k1q[3], k1w[3], k2q[3], k2w[3], k3q[3], k3w[3], k4q[3], k4w[3], tq[3], tw[3];
    for (int i = 0; i < 3; ++i) {
        q[i] = c9(x[i]); q[i].d[i] = 1;
        w[i] = c9(x[3 + i]); w[i].d[3 + i] = 1;
        ta[i] = c9(tau[i]); ta[i].d[6 + i] = 1;
    }
    for (int i = 0; i < 3; ++i) k1q[i] = w[i];
    cart2_acc(q, w, ta, k1w);
    for (int i = 0; i < 3; ++i) { tq[i] = a9(q[i], s9(0.5 * h, k1q[i])); tw[i] = a9(w[i], s9(0.5 * h, k1w[i])); k2q[i] = tw[i]; }
    cart2_acc(tq, tw, ta, k2w);
    for (int i = 0; i < 3; ++i) { tq[i] = a9(q[i], s9(0.5 * h, k2q[i])); tw[i] = a9(w[i], s9(0.5 * h, k2w[i])); k3q[i] = tw[i]; }
    cart2_acc(tq, tw, ta, k3w);
    for (int i = 0; i < 3; ++i) { tq[i] = a9(q[i], s9(h, k3q[i])); tw[i] = a9(w[i], s9(h, k3w[i])); k4q[i] = tw[i]; }
    cart2_acc(tq, tw, ta, k4w);
    for (int i = 0; i < 3; ++i) {
        dual9 oq = a9(q[i], s9(h / 6.0, a9(a9(k1q[i], s9(2.0, k2q[i])), a9(s9(2.0, k3q[i]), k4q[i]))));
        dual9 ow = a9(w[i], s9(h / 6.0, a9(a9(k1w[i], s9(2.0, k2w[i])), a9(s9(2.0, k3w[i]), k4w[i]))));
        xn[i] = oq.v; xn[3 + i] = ow.v;
        for (int j = 0; j < 9; ++j) { J[9 * i + j] = oq.d[j]; J[9 * (3 + i) + j] = ow.d[j]; }
    }
}

void dyn_cartpole2l(long K, const double *x, const double *tau, double h, double *xn, double *J) {
    for (long i = 0; i < K; ++i) dyn_cartpole2l_point(x + 6 * i, tau + 3 * i, h, xn + 6 * i, J + 54 * i);
}
