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
