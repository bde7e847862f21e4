// Argument block and workspace layout shared by the two interior-point kernels (alqp_ipm.hip: the
// size-generic one-wavefront-per-QP kernel; alqp_ipm_g4.hpp: the register/LDS-resident kernel) and by the
// CPU wave emulator of the latter (tests/emu). Plain C++: no HIP types.
#pragma once

#ifndef ALQP_HD
#ifdef __HIPCC__
#define ALQP_HD __host__ __device__
#else
#define ALQP_HD
#endif
#endif

namespace alqp_ipm {

template <typename real>
struct IpmArgs {
    int B, T;
    int flags;       // ALQP_IPM_*
    int max_iter;    // iterations done by THIS launch when ALQP_IPM_LOOP is set
    int iter0;       // index of the first iteration of this launch (for iter_best)
    real e;          // KKTeps (0 in the backward solve)
    const real *Cd, *c, *F, *f, *x0, *uhi, *ulo;
    long sC_t, sC_b, sF_t, sF_b, sf_t, sf_b;   // element strides (stage, instance) of Cd/c, F, f
    real *ws;        // [B][ws_words]
    long ws_words;
    const real *ry_ext;   // nullable [B][T*nx]: equality residual supplied by the caller (true dynamics)
    const real *gbar;     // backward: [B][T*n]
    real *o_x, *o_y, *o_z, *o_s;   // outputs: zhat/nus/lams/slacks (or dx / dnu / dlam / - in backward)
    real *o_resid, *o_mu;
    int *o_iter_best, *o_improved, *o_info;
};

// Per-instance workspace slab. Both kernels keep what has to survive between the launches of one solve
// (reference exit mode: one RESID and one STEP launch per iteration) at the SAME offsets: the iterate `cur`,
// the best iterate `best`, the negated residual `rr`, the scalars `scal`; all NK blocks are in the reference's
// order x | s | z | y with its row orderings (include/mi_alqp.h). The rest is private to the generic kernel.
template <typename real, int NX, int NU>
struct Lay {
    static constexpr int N = NX + NU;
    int T, nz, ni, ne, NK;
    long cur, best, res, da, dc, rr, r2, pinv, dt, wv, r1, scal, fac, total;
    ALQP_HD Lay(int T_, bool fac_in_ws) : T(T_) {
        nz = T * N; ni = 2 * T * NU; ne = T * NX; NK = nz + 2 * ni + ne;
        long o = 0;
        cur = o; o += NK; best = o; o += NK; res = o; o += NK; da = o; o += NK; dc = o; o += NK;
        rr = o; o += NK; r2 = o; o += NK; pinv = o; o += nz; dt = o; o += ni; wv = o; o += ni; r1 = o; o += nz;
        scal = o; o += 8; fac = o;
        if (fac_in_ws) o += 2L * T * NX * NX;
        total = (o + 15) & ~15L;
    }
    // offsets inside an NK block, reference order (x, s, z, y)
    ALQP_HD int ox() const { return 0; }
    ALQP_HD int os() const { return nz; }
    ALQP_HD int oz() const { return nz + ni; }
    ALQP_HD int oy() const { return nz + 2 * ni; }
};

}  // namespace alqp_ipm
