// TEST INFRASTRUCTURE - runs deq-mpc-corl_amd/csrc/alqp_ipm_g4.hpp (the register/LDS-resident interior-point
// kernel) in the 64-lane CPU emulator, one "wavefront" per QP, through the same argument block as the HIP launch.
// Built by tests/emu/build.py into tests/emu/libipm_g4_emu.so; used by tests/test_ipm_g4_emu.py only.
#include <cstdlib>
#include <vector>

#include "wave_emu.hpp"
#include "alqp_dims.hpp"
#include "alqp_ipm_g4.hpp"

using alqp_ipm::IpmArgs;

template <typename real, int NX, int NU, bool FULLT>
static int run_as(const IpmArgs<real> &a0, const real *lams, const real *slacks, int backward) {
    constexpr int SL = 5;
    using S = alqp_ipm_g4::Solver<real, NX, NU, SL, wave_emu::EmuX<real>, FULLT>;
    if (a0.T > S::TMAX || a0.T < 2) return -2;
    IpmArgs<real> a = a0;
    a.ws_words = alqp_ipm::Lay<real, NX, NU>(a.T, true).total;
    std::vector<real> lds(S::lds_words(a.T));
    for (int b = 0; b < a.B; ++b) {
        for (auto &v : lds) v = real(NAN);   // nothing may depend on what LDS held before
        S s(a, lds.data(), b);
        if (backward) s.run_backward(lams, slacks);
        else s.run_forward();
    }
    return 0;
}
// like the HIP launcher: the full horizon (T = 4 * SL) has its own instantiation
template <typename real, int NX, int NU>
static int run(const IpmArgs<real> &a, const real *lams, const real *slacks, int backward) {
    return a.T == 20 ? run_as<real, NX, NU, true>(a, lams, slacks, backward) : run_as<real, NX, NU, false>(a, lams, slacks, backward);
}

template <typename real>
static int dispatch(int nx, int nu, const IpmArgs<real> &a, const real *lams, const real *slacks, int backward) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return run<real, NX, NU>(a, lams, slacks, backward);
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return -2;
}

template <typename real>
static size_t ws_words(int nx, int nu, int T) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return (size_t)alqp_ipm::Lay<real, NX, NU>(T, true).total;
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return 0;
}

#define DEFINE(SFX, REAL)                                                                                           \
    extern "C" size_t emu_ipm_ws_words_##SFX(int nx, int nu, int T) { return ws_words<REAL>(nx, nu, T); }           \
    extern "C" int emu_ipm_solve_##SFX(int B, int T, int nx, int nu, int flags, int max_iter, int iter0,            \
                                       double kkt_eps, const REAL *Cd, const REAL *c, const REAL *F, const REAL *f, \
                                       const REAL *x0, const REAL *uhi, const REAL *ulo, long sC_t, long sC_b,      \
                                       long sF_t, long sF_b, long sf_t, long sf_b, REAL *ws, const REAL *ry_ext,    \
                                       REAL *zhat, REAL *nus, REAL *lams, REAL *slacks, REAL *resid, REAL *mu,      \
                                       int *iter_best, int *improved, int *info) {                                  \
        IpmArgs<REAL> a = {};                                                                                       \
        a.B = B; a.T = T; a.flags = flags; a.max_iter = max_iter; a.iter0 = iter0; a.e = (REAL)kkt_eps;             \
        a.Cd = Cd; a.c = c; a.F = F; a.f = f; a.x0 = x0; a.uhi = uhi; a.ulo = ulo;                                  \
        a.sC_t = sC_t; a.sC_b = sC_b; a.sF_t = sF_t; a.sF_b = sF_b; a.sf_t = sf_t; a.sf_b = sf_b;                   \
        a.ws = ws; a.ry_ext = ry_ext; a.o_x = zhat; a.o_y = nus; a.o_z = lams; a.o_s = slacks;                      \
        a.o_resid = resid; a.o_mu = mu; a.o_iter_best = iter_best; a.o_improved = improved; a.o_info = info;        \
        return dispatch<REAL>(nx, nu, a, nullptr, nullptr, 0);                                                      \
    }                                                                                                               \
    extern "C" int emu_ipm_backward_##SFX(int B, int T, int nx, int nu, const REAL *Cd, const REAL *F, long sC_t,   \
                                          long sC_b, long sF_t, long sF_b, const REAL *lams, const REAL *slacks,    \
                                          const REAL *gbar, REAL *ws, REAL *dx, REAL *dlam, REAL *dnu, int *info) { \
        IpmArgs<REAL> a = {};                                                                                       \
        a.B = B; a.T = T; a.e = 0; a.Cd = Cd; a.F = F; a.sC_t = sC_t; a.sC_b = sC_b; a.sF_t = sF_t; a.sF_b = sF_b;  \
        a.ws = ws; a.gbar = gbar; a.o_x = dx; a.o_z = dlam; a.o_y = dnu; a.o_info = info;                           \
        return dispatch<REAL>(nx, nu, a, lams, slacks, 1);                                                          \
    }

DEFINE(f64, double)
DEFINE(f32, float)
