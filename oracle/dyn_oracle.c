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

/* accelerations (xdd, thdd) at (theta, thetad, tau) */
static void cart_acc(dual6 th, dual6 thd, dual6 t0, dual6 t1, dual6 *xdd, dual6 *thdd) {
    dual6 sn = sin6(th), cs = cos6(th);
    dual6 r0 = a6(t0, s6(-1.0, m6(sn, m6(thd, thd))));   /* tau0 - sin(th) thd^2 */
    dual6 r1 = a6(t1, s6(9.81, sn));                      /* tau1 + 9.81 sin(th)  */
    dual6 idet = inv6(a6(c6(22.0), s6(-1.0, m6(cs, cs)))); /* det M = 22 - cos^2   */
    *xdd = m6(idet, a6(s6(2.0, r0), m6(cs, r1)));          /* M^-1 = [[2, c],[c, 11]] / det */
    *thdd = m6(idet, a6(m6(cs, r0), s6(11.0, r1)));
}

/* one point: q[2], qd[2], tau[2] -> qn[2], qdn[2], J[4][6] = d(qn, qdn)/d(q, qd, tau) */
void dyn_cartpole1l_point(const double *q, const double *qd, const double *tau, double h, double *xn, double *J) {
    dual6 x = c6(q[0]), th = c6(q[1]), xd = c6(qd[0]), thd = c6(qd[1]), t0 = c6(tau[0]), t1 = c6(tau[1]);
    x.d[0] = 1; th.d[1] = 1; xd.d[2] = 1; thd.d[3] = 1; t0.d[4] = 1; t1.d[5] = 1;
    dual6 k1x = xd, k1t = thd, k1xd, k1td;
    cart_acc(th, thd, t0, t1, &k1xd, &k1td);
    dual6 k2x = a6(xd, s6(0.5 * h, k1xd)), k2t = a6(thd, s6(0.5 * h, k1td)), k2xd, k2td;
    cart_acc(a6(th, s6(0.5 * h, k1t)), k2t, t0, t1, &k2xd, &k2td);
    dual6 k3x = a6(xd, s6(0.5 * h, k2xd)), k3t = a6(thd, s6(0.5 * h, k2td)), k3xd, k3td;
    cart_acc(a6(th, s6(0.5 * h, k2t)), k3t, t0, t1, &k3xd, &k3td);
    dual6 k4x = a6(xd, s6(h, k3xd)), k4t = a6(thd, s6(h, k3td)), k4xd, k4td;
    cart_acc(a6(th, s6(h, k3t)), k4t, t0, t1, &k4xd, &k4td);
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

/* K points: x[K][4] = (q, qd), tau[K][2] -> xn[K][4], J[K][4][6] */
void dyn_cartpole1l(long K, const double *x, const double *tau, double h, double *xn, double *J) {
    for (long i = 0; i < K; ++i) dyn_cartpole1l_point(x + 4 * i, x + 4 * i + 2, tau + 2 * i, h, xn + 4 * i, J + 24 * i);
}
