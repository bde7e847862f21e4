// alqp_kernels.hip - gfx950 kernels + the C ABI of include/mi_alqp.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -I../../include alqp_kernels.hip -o libmi_alqp.so
// No torch types anywhere: plain device pointers in, kernel launches on the caller's stream.
// The file can be compiled as one translation unit (default) or, to build in parallel, as
// three objects: -DALQP_PART=1 (team kernels, helper kernels, C ABI), -DALQP_PART=2 (quad
// kernels fp32), -DALQP_PART=3 (quad kernels fp64). See build.sh.
#include <hip/hip_runtime.h>

#include "alqp_team.hpp"
#include "alqp_quad.hpp"
#include "mi_alqp.h"
#include "alqp_dims.hpp"

#ifndef ALQP_PART
#define ALQP_PART 0
#endif
#define ALQP_BUILD_MAIN (ALQP_PART == 0 || ALQP_PART == 1)
#define ALQP_BUILD_QUAD (ALQP_PART == 0 || ALQP_PART == 2 || ALQP_PART == 3)
#define ALQP_QUAD_F32 (ALQP_PART == 0 || ALQP_PART == 2)
#define ALQP_QUAD_F64 (ALQP_PART == 0 || ALQP_PART == 3)

#include <hip/hip_cooperative_groups.h>

namespace alqp {

// Kernel arguments are kept lean on purpose: every pointer is two SGPRs for the whole
// kernel, and SGPR spills (v_writelane/v_readlane) showed up in the hot loops otherwise.
template <typename real>
struct SolveArgs {
    int B, T;
    int al_iter, max_newton, n_ls, flags;
    real rho_scale;
    const real *Qd, *q, *F, *c, *x0, *ulo, *uhi;
    long sb_u, st_u;
    real *z, *lam, *rho, *phi, *rnorm2;
    int *info;
    unsigned char *status;
    real *factor;
    const double *skip;  // nullable: *skip != 0 -> the launch does nothing (device-side loop exit)
    real dyn_h;          // nonlinear fused solve: step length of the inlined dynamics model
    int stagger;         // quad solve: start offset between the four wavefronts of a CU, units of ~1024 clocks (0: none)
    // ALQP_EXIT_IN_KERNEL (cooperative launch): the reference's batch-global exit test of the Newton loop inside the launch
    double exit_tol;
    int *newton_counts;    // [al_iter] executed Newton steps per AL iteration
    double *exit_scratch;  // arrival counter, then [2][gridDim.x] per-workgroup partial sums (ping-pong), then a time-out flag
};

// sum of one value per workgroup over the whole (cooperatively launched) grid, the same bits in every lane of every
// workgroup: partials in workgroup order, 64 interleaved chains, xor butterfly. Every workgroup must call it.
__device__ inline double grid_sum_ordered(double block_val, double *scratch, int &phase) {
    // scratch: one arrival counter (8 bytes, zeroed by the host before the launch), then [2][gridDim.x] partials (ping-pong).
    // Only the partials and the counter are shared between workgroups and all of them are touched with agent-scope
    // atomics, so no cache write-back / invalidate (a full fence) is needed: a workgroup's own records are its own.
    // Co-residency of the grid is guaranteed by the cooperative launch, so the spin cannot starve anyone.
    double *p = scratch + 1 + (size_t)(phase & 1) * gridDim.x;
    unsigned *cnt = reinterpret_cast<unsigned *>(scratch);
    double *timed_out = scratch + 1 + 2 * (size_t)gridDim.x;
    if (threadIdx.x == 0) {
        // Ordering: the partial is an agent-scope store, the arrival an agent-scope RELEASE add (the partial is visible to
        // whoever observes the count), the consumer side one ACQUIRE load after the spin (its later loads of the partials
        // cannot be satisfied from before the count was seen). The spin itself polls relaxed (an acquire per poll is 2-3x
        // slower per hop, MI355X_MICROARCH.md "Invalid forms").
        __hip_atomic_store(p + blockIdx.x, block_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(phase + 1) * gridDim.x;
        // Bounded spin: ~1 s of wall clock (s_memrealtime ticks at 100 MHz), and none at all once a previous phase has
        // timed out - every wavefront reaches an exit even if an arrival never shows up; the caller then sees NaN, never
        // takes the early exit and reports -1 Newton steps, which the host turns into an error.
        if (__hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0.0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool late = false;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull ||
                    __hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) { late = true; break; }
            }
            if (late) __hip_atomic_store(timed_out, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else (void)__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_s_barrier();   // one wavefront per workgroup: re-converges the lanes behind lane 0's spin
    if (__hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) {
        ++phase;
        return __builtin_nan("");
    }
    double s = 0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 64) s += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off, 64);
    ++phase;
    return s;
}
// An instance's term of the batch norm the Newton loop exits on (al_utils.py:552). One whose factorisation hit a
// non-positive pivot counts as +inf, so that the exit does not fire: the reference's norm holds the undefined residuals
// of such instances and ran the full 4 steps in both recorded batches (tests/test_failure_path.py).
template <typename real>
__device__ inline double exit_term(real r2, int info) { return info ? (double)INFINITY : (double)r2; }
__device__ inline bool grid_barrier_timed_out(const double *scratch) {
    return __hip_atomic_load(scratch + 1 + 2 * (size_t)gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0;
}
// one value per lane group (teams / quads: `leader` marks one lane per instance) -> the workgroup's sum, fixed order
__device__ inline double wave_sum_leaders(double v, bool leader) {
    double s = leader ? v : 0.0;
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

template <typename real>
struct TraceArgs {
    real *g, *d, *phi, *phi_prev;
    int *k, *accept;
};

template <typename real>
struct StepArgs {
    int B, T;
    const real *z, *xnext, *F, *x0, *lam, *rho, *Qd, *q, *ulo, *uhi;
    long sb_u, st_u;
    real *d_out, *g_out, *factor;
    int *info;
    const real *obs;   // nullable [B][T][nobs][3]: obstacle centres (Obstacle_MPC)
    int nobs;
    real obs_r2;
    int no_init;       // state-estimator row set (AlqpObstacles.state_estimator)
};

template <typename real>
struct BwdArgs {
    int B, T;
    const real *factor, *F, *rho, *z_final, *gbar;
    real *q_grad, *Qd_grad;
};

// A VGPR zero the compiler cannot see through: keeps LDS addresses "divergent", so that
// wave-uniform operand reads stay 16-byte vector ds_reads instead of being scalarised.
__device__ inline unsigned opaque_zero() {
    unsigned z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

#if ALQP_BUILD_MAIN
#ifdef ALQP_PHASE_TIMING
__device__ unsigned long long g_team_cycles[8];   // debug build only (tools/team_timing.py)
#endif
// ---- fused LinDx solve -------------------------------------------------------------
// OCC: wavefronts per SIMD the register allocation is capped for. 2 (256 registers) lets a CU hold the 8 teams its
// LDS has room for, but costs 80 B (fp32) / 252 B (fp64) of scratch per lane at (13,4); 1 (no cap) has no scratch.
// Measured (profiles/r02/experiments): fp64 is faster uncapped at every batch the team kernels see (B = 200: 1.27 ->
// 1.22 ms, B = 2048: 3.43 -> 2.85 ms); fp32 only while there is at most one wavefront per SIMD anyway (B = 200: 0.84 ->
// 0.80 ms, B = 1024: 0.91 -> 0.87 ms; B = 2048: 1.07 against 1.71 ms). dispatch_solve() picks accordingly.
// The TRACE instantiations (tests only) are never capped: with their extra live pointers the capped fp64 builds spill ~70-150
// registers, and this hipcc (ROCm 7.2) can place such a spill store at the head of a divergent loop's exit block IN FRONT OF the
// `s_or_b64 exec` that re-enables the lanes - the store then runs with EXEC = 0 and the reload returns stale scratch. Seen once
// (k_solve_lin<double,12,4,true,2>: the zs base offset spilled after the `for (e = li; e < T*N; e += G)` load loop, every row
// of the returned z one repeated value); the uncapped builds have no spills (checked per kernel: tools/spill_report.py).
template <typename real, int NX, int NU, bool TRACE, int OCC = TRACE ? 1 : 2>
__global__ __launch_bounds__(64, OCC) void k_solve_lin(SolveArgs<real> a, TraceArgs<real> tr) {
    if (a.skip && *a.skip != 0.0) return;  // block-uniform, before any barrier
    using C = Cfg<real, NX, NU>;
    constexpr int G = C::G, N = C::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    real *smem = reinterpret_cast<real *>(smem_raw);
    const int lane = threadIdx.x, team = lane / G, li = lane % G;
    const int b_raw = blockIdx.x * C::QPW + team;
    const bool active = b_raw < a.B;
    const int b = active ? b_raw : a.B - 1;
    const int T = a.T, M = C::M(T), neq = T * NX;

    Team<real, NX, NU> tm;
#ifdef ALQP_PHASE_TIMING
    for (int i = 0; i < 8; ++i) tm.tacc[i] = 0;
    tm.stamp(-1);
#endif
    tm.init(smem + (size_t)team * C::team_words(T) + opaque_zero(), li, team * G, T, b);
    tm.gQd = a.Qd + (size_t)b * T * N;
    tm.gq = a.q + (size_t)b * T * N;
    tm.gF = a.F + (size_t)b * (T - 1) * NX * N;
    tm.gc = a.c + (size_t)b * (T - 1) * NX;
    tm.gx0 = a.x0 + (size_t)b * NX;
    tm.gulo = a.ulo + (size_t)b * a.sb_u;
    tm.guhi = a.uhi + (size_t)b * a.sb_u;
    tm.st_u = a.st_u;

    real *gz = a.z + (size_t)b * T * N;
    tm.lams = a.lam + (size_t)b * M;  // multipliers stay in global memory (L2), updated in place
    for (int e = li; e < T * N; e += G) tm.zs[e] = gz[e];
    tm.rho = a.rho[b];
    real phi_prev = a.phi[b];
    wave_sync();
    tm.residual_sweep();

    int step_id = 0;
    const bool ref_exit = (a.flags & ALQP_EXIT_IN_KERNEL) != 0;   // grid-uniform
    int gphase = 0;
    for (int it = 0; it < a.al_iter; ++it) {
        if (a.flags & ALQP_INIT_MERIT) {
            real p1[1];
            tm.template merit_candidates<1>(p1, true);
            phi_prev = p1[0];
        }
        double nrm_old = 0;
        int n_done = 0;
        if (ref_exit) {   // ||r+|| over the whole batch at the start of the Newton loop (al_utils.py:486)
            const real r0 = tm.rplus2();
            nrm_old = sqrt(grid_sum_ordered(wave_sum_leaders(exit_term(r0, tm.info), active && li == 0), a.exit_scratch, gphase));
        }
        for (int st = 0; st < a.max_newton; ++st, ++step_id) {
            real *tg = nullptr;
            if constexpr (TRACE) tg = (tr.g && active) ? tr.g + ((size_t)step_id * a.B + b) * T * N : nullptr;
            tm.forward_sweep(tg);
#ifdef ALQP_PHASE_TIMING
            tm.stamp(-1);
#endif
            tm.backward_sweep();
#ifdef ALQP_PHASE_TIMING
            tm.stamp(4);
#endif
            if constexpr (TRACE) {
                if (tr.d && active) {
                    real *td = tr.d + ((size_t)step_id * a.B + b) * T * N;
                    for (int e = li; e < T * N; e += G) td[e] = tm.ds[e];
                }
            }
            real ph[20];
            tm.template merit_candidates<20>(ph, false);
#ifdef ALQP_PHASE_TIMING
            tm.stamp(5);
#endif
            // first argmin, a NaN wins like torch.min (al_utils.py:634)
            int kbest = 0;
            real best = ph[0];
            if (a.n_ls == 20) {
#pragma unroll
                for (int k = 1; k < 20; ++k) {
                    if (!(best != best) && (ph[k] != ph[k] || ph[k] < best)) {
                        best = ph[k];
                        kbest = k;
                    }
                }
            } else if constexpr (C::RB * C::NXP >= 20) {
                // fewer candidates (tests): go through LDS (the Ft region is free here)
                // instead of 20 hoisted lane masks
                if (li == 0) {
#pragma unroll
                    for (int k = 0; k < 20; ++k) tm.Ft[k] = ph[k];
                }
                wave_sync();
                for (int k = 1; k < a.n_ls; ++k) {
                    real v = tm.Ft[k];
                    if (!(best != best) && (v != v || v < best)) {
                        best = v;
                        kbest = k;
                    }
                }
                wave_sync();
                if (li < 20) tm.Ft[li] = 0;  // restore the constant zeros of the SYRK operand
                wave_sync();
            } else {
#pragma unroll
                for (int k = 1; k < 20; ++k) {
                    if (k < a.n_ls && !(best != best) && (ph[k] != ph[k] || ph[k] < best)) {
                        best = ph[k];
                        kbest = k;
                    }
                }
            }
            const bool acc = best < phi_prev;
            if constexpr (TRACE) {
                if (active && li == 0) {
                    if (tr.phi)
#pragma unroll
                        for (int k = 0; k < 20; ++k)
                            if (k < a.n_ls) tr.phi[((size_t)step_id * a.n_ls + k) * a.B + b] = ph[k];
                    if (tr.phi_prev) tr.phi_prev[(size_t)step_id * a.B + b] = phi_prev;
                    if (tr.k) tr.k[(size_t)step_id * a.B + b] = kbest;
                    if (tr.accept) tr.accept[(size_t)step_id * a.B + b] = acc ? 1 : 0;
                }
            }
            const real alpha = acc ? real(1) / real(1 << kbest) : real(0);
            for (int e = li; e < T * N; e += G) tm.zs[e] += alpha * tm.ds[e];
            for (int e = li; e < neq; e += G) tm.req[e] += alpha * tm.seq[e];
            wave_sync();
#ifdef ALQP_PHASE_TIMING
            tm.stamp(6);
#endif
            phi_prev = best;  // merit <- new_merit even when rejected (al_utils.py:569)
            if (ref_exit) {   // al_utils.py:551-564, the same test alqp_exit_test takes between launches
                const real r1 = tm.rplus2();
                const double nw = sqrt(grid_sum_ordered(wave_sum_leaders(exit_term(r1, tm.info), active && li == 0), a.exit_scratch, gphase));
                ++n_done;
                if (nw < a.exit_tol || fabs(nrm_old - nw) / nw < a.exit_tol) break;
                nrm_old = nw;
            }
        }
        if (ref_exit && a.newton_counts && blockIdx.x == 0 && lane == 0) a.newton_counts[it] = grid_barrier_timed_out(a.exit_scratch) ? -1 : n_done;
        if (a.flags & ALQP_DUAL_UPDATE) {
            if (active) tm.dual_update();  // in-place on global lam: padding teams must not touch it
            tm.rho *= a.rho_scale;
        }
    }

    const real rn2 = tm.rplus2();
    int bad = 0;
    for (int e = li; e < T * N; e += G) {
        real v = tm.zs[e];
        bad |= !(v - v == real(0));
    }
    bad = team_or<G>(bad);
#ifdef ALQP_PHASE_TIMING
    tm.stamp(7);
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_team_cycles[i], tm.tacc[i]);
#endif
    if (active) {
        for (int e = li; e < T * N; e += G) gz[e] = tm.zs[e];
        if ((a.flags & ALQP_SAVE_FACTOR) && a.factor) {
            real *gf = a.factor + (size_t)b * T * C::XT;
            for (int e = li; e < T * C::XT; e += G) gf[e] = tm.Xp[e];
        }
        if (li == 0) {
            a.rho[b] = tm.rho;
            a.phi[b] = phi_prev;
            if (a.rnorm2) a.rnorm2[b] = rn2;
            if (a.info && tm.info && a.info[b] == 0) a.info[b] = tm.info;  // sticky: first failure of the solve
            if (a.status) a.status[b] = bad ? 0 : 1;
        }
    }
}

#endif  // ALQP_BUILD_MAIN

#if ALQP_BUILD_QUAD
// line-search merits accumulated inside the backward sweep (1) or by a pass of their own (0)
#ifndef ALQP_FUSE_LS
#define ALQP_FUSE_LS 1
#endif
// ---- fused LinDx solve, quad variant (4 lanes per instance, HBM workspace) --------------
#ifdef ALQP_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[10];
#define QSTAMP(b) qd.stamp(b)
#else
#define QSTAMP(b)
#endif
// Dyn = NoDyn: affine dynamics from the caller's F, c (alqp_solve_lin). Dyn = a model of alqp_dyn.hpp:
// the nonlinear solve with that model inlined (alqp_solve_nonlin): every Newton step re-linearises
// on the device, the line search and the dual update use the true dynamics; a.F then points at the
// F region of the workspace (behind the records) and a.c is unused.
template <typename real, int NX, int NU, bool TRACE, class Dyn = NoDyn>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_solve_lin_quad(SolveArgs<real> a, TraceArgs<real> tr, real *ws) {
    using C = QCfg<real, NX, NU>;
    constexpr bool NL = Dyn::ID != 0;
    // fp64 is short of registers already: its line search keeps a pass of its own
#ifndef ALQP_FUSE_LS_F64
#define ALQP_FUSE_LS_F64 1
#endif
    constexpr bool FUSE_LS = ALQP_FUSE_LS && (sizeof(real) == 4 || ALQP_FUSE_LS_F64) && !NL;
    constexpr int N = C::N;
    if (a.skip && *a.skip != 0.0) return;  // wave-uniform
    const int lane = threadIdx.x, qi = lane >> 2;
    const int b_raw = blockIdx.x * 16 + qi;
    const bool active = b_raw < a.B;
    {
        // Phase shift between the four wavefronts of a CU (one per SIMD, all running the same sweeps): the wave
        // on SIMD s starts s * stagger * ~1024 clocks late. Started together they march in lock step and queue
        // on the CU's vector-memory pipeline in their load phases while it idles in their arithmetic phases;
        // a fifth of a sweep apart, one wave's memory phase runs under the others' panels (+5 % at the headline
        // size, delay included; shifting whole XCDs instead buys nothing: the contention is inside the CU).
        // a.stagger is set by the host (quad_stagger(): only when the grid fills the SIMDs and the launch is long
        // enough to amortise the delay); flags bits 24-31 / 20-23 override it for experiments.
        int stag = a.stagger;
        const int mode = (a.flags >> 20) & 0xf;
        if ((a.flags >> 24) & 0xff) stag = (a.flags >> 24) & 0xff;
        if (stag > 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID: SIMD id in bits 5:4
            int simd = (hw >> 4) & 3;
            if (mode == 1) simd = simd * 16 + ((hw >> 8) & 0xf);   // all 64 (SIMD, CU-in-array) pairs apart
            if (mode == 2) simd &= 1;                              // two groups per CU
            if (mode == 3) simd = blockIdx.x & 1;                  // two groups of XCDs
            if (mode == 4) simd = (simd & 1) ^ (blockIdx.x & 1);   // two groups, mixed over SIMDs and XCDs
            if (mode == 5) simd = (hw >> 8) & 3;                   // four groups of CUs, a CU's SIMDs together
            for (int i = 0; i < simd * stag; ++i) __builtin_amdgcn_s_sleep(16);
        }
    }
#ifdef ALQP_ALIAS_ALL
    // experiment only (tools/phase_timing.sh -DALQP_ALIAS_ALL): every instance of an XCD-sized group
    // works on the same data, so the launch runs out of L2: what remains is the on-chip time
    const int b = (active ? b_raw : a.B - 1) % ALQP_ALIAS_ALL;
#else
    const int b = active ? b_raw : a.B - 1;
#endif
    const int T = a.T, M = C::M(T);

    Quad<real, NX, NU> qd;
    __shared__ real w_lds[QCfg<real, NX, NU>::WLDS_WORDS];   // fp64: the W panel (WPanel); one word otherwise
    qd.wl = w_lds + threadIdx.x;
    qd.q = lane & 3;
    qd.T = T;
    qd.active = active;
    qd.gQd = a.Qd + (size_t)b * T * N;
    qd.gq = a.q + (size_t)b * T * N;
    qd.gF = a.F + (size_t)b * (T - 1) * NX * N;
    qd.gc = a.c + (size_t)b * (T - 1) * NX;
    qd.gx0 = a.x0 + (size_t)b * NX;
    qd.gulo = a.ulo + (size_t)b * a.sb_u;
    qd.guhi = a.uhi + (size_t)b * a.sb_u;
    qd.st_u = a.st_u;
    qd.gz = a.z + (size_t)b * T * N;
    qd.glam = a.lam + (size_t)b * M;
    qd.rec = ws + C::rec_base(b, T);
    qd.gFw = const_cast<real *>(qd.gF);
    qd.dyn_h = a.dyn_h;
    qd.rho = a.rho[b];
    qd.info = 0;
    real phi_prev = a.phi[b];
#ifdef ALQP_PHASE_TIMING
    for (int i = 0; i < 10; ++i) qd.tacc[i] = 0;
#endif
    QSTAMP(-1);
    // the residual pre-pass is only needed when no forward sweep will run before r is used
    if (!(a.flags & ALQP_WS_PRIMED)) {
        if constexpr (NL) qd.stage_in(false, false);
        else qd.stage_in(a.max_newton == 0 || a.al_iter == 0 || (a.flags & ALQP_EXIT_IN_KERNEL) ||
                         (!C::PHI0_FWD && (a.flags & ALQP_INIT_MERIT)));
    }
    const bool ref_exit = (a.flags & ALQP_EXIT_IN_KERNEL) != 0;   // grid-uniform
    int gphase = 0;

    int step_id = 0;
    bool pend = false;  // a chosen step not yet applied (the next forward sweep applies it)
    real alpha_pend = 0;
    int bad = 0;
    real rn2 = 0, phi_next = 0;
    for (int it = 0; it < a.al_iter; ++it) {
        // starting merit of the iteration (al_utils.py:481): iterations > 0 get it from iter_end() of
        // the previous one; the first gets it from its first forward sweep, or from a pass of its
        // own when the launch has no Newton step
        bool phi_from_forward = false;
        if (a.flags & ALQP_INIT_MERIT) {
            if (it > 0) {
                phi_prev = phi_next;
            } else if (C::PHI0_FWD && a.max_newton > 0 && !ref_exit) {
                phi_from_forward = true;
            } else {
                if constexpr (NL) qd.template linearize<Dyn>(real(0), false);  // true residuals for the merit
                real p1[1];
                qd.template merit_candidates<1>(p1, true);
                phi_prev = p1[0];
            }
        }
        pend = false;
        double nrm_old = 0;
        int n_done = 0;
        if (ref_exit) {   // ||r+|| over the whole batch at the start of the Newton loop (al_utils.py:486)
            int bad0 = 0;
            const real r0 = it == 0 ? qd.rplus2(bad0) : rn2;   // later iterations: from iter_end() of the previous one
            nrm_old = sqrt(grid_sum_ordered(wave_sum_leaders(exit_term(r0, qor(qd.info)), active && qd.q == 0), a.exit_scratch, gphase));
        }
        for (int st = 0; st < a.max_newton; ++st, ++step_id) {
            real *tg = nullptr;
            if constexpr (TRACE) tg = (tr.g && active) ? tr.g + ((size_t)step_id * a.B + b) * T * N : nullptr;
            QSTAMP(9);  // everything between Newton steps
            if constexpr (NL) {
                qd.template linearize<Dyn>(alpha_pend, pend);
                pend = false;
            }
            qd.forward(tg, alpha_pend, pend, (phi_from_forward && st == 0) ? &phi_prev : nullptr);
            real ph[20];
            qd.template backward<FUSE_LS>(ph);
            QSTAMP(-1);
            if constexpr (TRACE) {
                if (tr.d && active) {
                    real *td = tr.d + ((size_t)step_id * a.B + b) * T * N;
                    for (int t = 0; t < T; ++t)
                        for (int j = qd.q; j < N; j += 4) td[t * N + j] = qd.recp(t)[C::oY + C::pn(j)];
                }
            }
            if constexpr (NL) qd.template merit_nonlin<Dyn>(ph);
            else if constexpr (!FUSE_LS) qd.template merit_candidates<20>(ph, false);
            QSTAMP(6);  // line-search candidates
            int kbest = 0;
            real best = ph[0];
#pragma unroll
            for (int k = 1; k < 20; ++k) {
                if (k < a.n_ls && !(best != best) && (ph[k] != ph[k] || ph[k] < best)) {
                    best = ph[k];
                    kbest = k;
                }
            }
            const bool acc = best < phi_prev;
            if constexpr (TRACE) {
                if (active && qd.q == 0) {
                    if (tr.phi)
#pragma unroll
                        for (int k = 0; k < 20; ++k)
                            if (k < a.n_ls) tr.phi[((size_t)step_id * a.n_ls + k) * a.B + b] = ph[k];
                    if (tr.phi_prev) tr.phi_prev[(size_t)step_id * a.B + b] = phi_prev;
                    if (tr.k) tr.k[(size_t)step_id * a.B + b] = kbest;
                    if (tr.accept) tr.accept[(size_t)step_id * a.B + b] = acc ? 1 : 0;
                }
            }
            const real alpha = acc ? real(1) / real(1 << kbest) : real(0);
            pend = true;  // applied by the next forward sweep or by iter_end()
            alpha_pend = alpha;
            QSTAMP(7);  // pick
            phi_prev = best;  // merit <- new_merit even when rejected (al_utils.py:569)
            if (ref_exit) {
                // apply the step now (what the end of a one-step launch does), ||r+||^2 at the new iterate, then the
                // reference's batch-global test (al_utils.py:551-564; alqp_exit_test between launches otherwise)
                int bad1 = 0;
                real ph_unused = 0, r1 = 0;
                qd.template iter_end<Dyn>(alpha_pend, pend, false, (real)a.rho_scale, false, ph_unused, r1, bad1);
                pend = false;
                const double nw = sqrt(grid_sum_ordered(wave_sum_leaders(exit_term(r1, qor(qd.info)), active && qd.q == 0), a.exit_scratch, gphase));
                ++n_done;
                if (nw < a.exit_tol || fabs(nrm_old - nw) / nw < a.exit_tol) break;
                nrm_old = nw;
            }
        }
        if (ref_exit && a.newton_counts && blockIdx.x == 0 && lane == 0) a.newton_counts[it] = grid_barrier_timed_out(a.exit_scratch) ? -1 : n_done;
        bad = 0;
        qd.template iter_end<Dyn>(alpha_pend, pend, (a.flags & ALQP_DUAL_UPDATE) != 0, (real)a.rho_scale, it + 1 == a.al_iter,
                    phi_next, rn2, bad);
        pend = false;
    }
    if (a.al_iter <= 0) {
        rn2 = qd.rplus2(bad);
        qd.stage_out();
    }
#ifdef ALQP_PHASE_TIMING
    QSTAMP(9);
    if (lane == 0)
        for (int i = 0; i < 10; ++i) atomicAdd(&g_phase_cycles[i], qd.tacc[i]);
#endif
    if (active && qd.q == 0) {
        a.rho[b] = qd.rho;
        a.phi[b] = phi_prev;
        if (a.rnorm2) a.rnorm2[b] = rn2;
        if (a.info && qd.info && a.info[b] == 0) a.info[b] = qd.info;  // sticky: first failure of the solve
        if (a.status) a.status[b] = bad ? 0 : 1;
    }
}

// ---- backward of the implicit layer, quad variant: the factor is the workspace a previous
//      quad solve left behind (per-stage lower triangles of L) ---------------------------------
template <typename real, int NX, int NU>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_backward_quad(BwdArgs<real> a, real *ws) {
    using C = QCfg<real, NX, NU>;
    constexpr int N = C::N;
    const int lane = threadIdx.x, qi = lane >> 2;
    const int b_raw = blockIdx.x * 16 + qi;
    const bool active = b_raw < a.B;
    const int b = active ? b_raw : a.B - 1;
    const int T = a.T;
    Quad<real, NX, NU> qd;
    __shared__ real w_lds[QCfg<real, NX, NU>::WLDS_WORDS];   // fp64: the W panel (WPanel); one word otherwise
    qd.wl = w_lds + threadIdx.x;
    qd.q = lane & 3;
    qd.T = T;
    qd.active = active;
    qd.gF = a.F + (size_t)b * (T - 1) * NX * N;
    qd.gQd = nullptr; qd.gq = nullptr; qd.gc = nullptr; qd.gx0 = nullptr; qd.gulo = nullptr; qd.guhi = nullptr;
    qd.st_u = 0;
    qd.gz = nullptr; qd.glam = nullptr;
    qd.rec = ws + C::rec_base(b, T);
    qd.rho = a.rho[b];
    qd.info = 0;
    qd.solve_forward(a.gbar + (size_t)b * T * N);
    real unused[20];
    qd.template backward<false>(unused);
    if (active) {
        const real *zf = a.z_final + (size_t)b * T * N;
        real *qg = a.q_grad + (size_t)b * T * N;
        real *Qg = a.Qd_grad + (size_t)b * T * N;
        for (int t = 0; t < T; ++t)
            for (int j = qd.q; j < N; j += 4) {
                const real w = qd.recp(t)[C::oY + C::pn(j)];
                qg[t * N + j] = w;
                Qg[t * N + j] = w * zf[t * N + j];
            }
    }
}

// ---- one Newton direction (nonlinear-caller mode), quad variant: 16 instances per wavefront, the
//      factor streamed through (and left in) the workspace records, like the fused quad solve. The
//      caller evaluated dx_jac -> (xnext = f(z), F) in PyTorch (al_utils.py:233-248); the sweeps run on the
//      linearisation F at z with the true residual z_{t+1}[x] - xnext_t.
template <typename real, int NX, int NU>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_newton_step_quad(StepArgs<real> a, real *ws) {
    using C = QCfg<real, NX, NU>;
    constexpr int N = C::N, SW = C::SW;
    const int lane = threadIdx.x, qi = lane >> 2;
    const int b_raw = blockIdx.x * 16 + qi;
    const bool active = b_raw < a.B;
    const int b = active ? b_raw : a.B - 1;
    const int T = a.T, M = C::M(T) + T * a.nobs;   // obstacle rows behind each stage's bound rows
    Quad<real, NX, NU> qd;
    __shared__ real w_lds[QCfg<real, NX, NU>::WLDS_WORDS];   // fp64: the W panel (WPanel); one word otherwise
    qd.wl = w_lds + threadIdx.x;
    qd.q = lane & 3;
    qd.T = T;
    qd.active = active;
    qd.gQd = a.Qd + (size_t)b * T * N;
    qd.gq = a.q + (size_t)b * T * N;
    qd.gF = a.F + (size_t)b * (T - 1) * NX * N;
    qd.gc = nullptr;
    qd.gx0 = a.x0 + (size_t)b * NX;
    qd.gulo = a.ulo + (size_t)b * a.sb_u;
    qd.guhi = a.uhi + (size_t)b * a.sb_u;
    qd.st_u = a.st_u;
    qd.gz = const_cast<real *>(a.z) + (size_t)b * T * N;        // read-only here
    qd.glam = const_cast<real *>(a.lam) + (size_t)b * M;        // read-only here
    qd.rec = ws + C::rec_base(b, T);
    qd.gFw = const_cast<real *>(qd.gF);
    qd.dyn_h = 0;
    qd.rho = a.rho[b];
    qd.info = 0;
    qd.nobs = a.nobs;
    qd.gobs = a.nobs > 0 ? a.obs + (size_t)b * T * a.nobs * 3 : nullptr;
    qd.obs_r2 = a.obs_r2;
    qd.no_init = a.no_init != 0;
    // no copy-in: the forward sweep reads the caller's arrays itself and takes r_t = z_{t+1}[x] - xnext_t
    const real *gxn = a.xnext + (size_t)b * (T - 1) * NX;
    real *tg = (a.g_out && active) ? a.g_out + (size_t)b * T * N : nullptr;
    qd.template forward<true>(tg, real(0), false, nullptr, gxn);
    real unused[20];
    qd.template backward<false>(unused, a.d_out + (size_t)b * T * N);
    if (active) {
        if (qd.q == 0 && a.info && qd.info && a.info[b] == 0) a.info[b] = qd.info;
    }
}

#endif  // ALQP_BUILD_QUAD

#if ALQP_BUILD_MAIN
// ---- one Newton direction (nonlinear-caller mode) ----------------------------------
template <typename real, int NX, int NU>
__global__ __launch_bounds__(64) void k_newton_step(StepArgs<real> a) {
    using C = Cfg<real, NX, NU>;
    constexpr int G = C::G, N = C::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    real *smem = reinterpret_cast<real *>(smem_raw);
    const int lane = threadIdx.x, team = lane / G, li = lane % G;
    const int b_raw = blockIdx.x * C::QPW + team;
    const bool active = b_raw < a.B;
    const int b = active ? b_raw : a.B - 1;
    const int T = a.T, M = C::M(T) + T * a.nobs;

    Team<real, NX, NU> tm;
    tm.init(smem + (size_t)team * C::team_words(T) + opaque_zero(), li, team * G, T, b);
    if (a.nobs > 0) { tm.gobs = a.obs + (size_t)b * T * a.nobs * 3; tm.nobs = a.nobs; tm.obs_r2 = a.obs_r2; }
    tm.no_init = a.no_init != 0;
    tm.gQd = a.Qd + (size_t)b * T * N;
    tm.gq = a.q + (size_t)b * T * N;
    tm.gF = a.F + (size_t)b * (T - 1) * NX * N;
    tm.gc = nullptr;
    tm.gx0 = a.x0 + (size_t)b * NX;
    tm.gulo = a.ulo + (size_t)b * a.sb_u;
    tm.guhi = a.uhi + (size_t)b * a.sb_u;
    tm.st_u = a.st_u;
    tm.gxnext = a.xnext + (size_t)b * (T - 1) * NX;
    const real *gz = a.z + (size_t)b * T * N;
    tm.lams = const_cast<real *>(a.lam) + (size_t)b * M;  // read-only here
    for (int e = li; e < T * N; e += G) tm.zs[e] = gz[e];
    tm.rho = a.rho[b];
    wave_sync();
    tm.forward_sweep((active && a.g_out) ? a.g_out + (size_t)b * T * N : nullptr);
    tm.backward_sweep();
    if (active) {
        real *gd = a.d_out + (size_t)b * T * N;
        for (int e = li; e < T * N; e += G) gd[e] = tm.ds[e];
        if (a.factor) {
            real *gf = a.factor + (size_t)b * T * C::XT;
            for (int e = li; e < T * C::XT; e += G) gf[e] = tm.Xp[e];
        }
        if (li == 0 && a.info && tm.info && a.info[b] == 0) a.info[b] = tm.info;
    }
}

// ---- backward of the implicit layer -------------------------------------------------
template <typename real, int NX, int NU>
__global__ __launch_bounds__(64) void k_backward(BwdArgs<real> a) {
    using C = Cfg<real, NX, NU>;
    constexpr int G = C::G, N = C::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    real *smem = reinterpret_cast<real *>(smem_raw);
    const int lane = threadIdx.x, team = lane / G, li = lane % G;
    const int b_raw = blockIdx.x * C::QPW + team;
    const bool active = b_raw < a.B;
    const int b = active ? b_raw : a.B - 1;
    const int T = a.T;

    Team<real, NX, NU> tm;
    tm.init(smem + (size_t)team * C::team_words(T) + opaque_zero(), li, team * G, T, b);
    tm.gF = a.F + (size_t)b * (T - 1) * NX * N;
    const real *gf = a.factor + (size_t)b * T * C::XT;
    const real *gg = a.gbar + (size_t)b * T * N;
    for (int e = li; e < T * C::XT; e += G) tm.Xp[e] = gf[e];
    for (int e = li; e < T * N; e += G) tm.ds[e] = -gg[e];
    tm.rho = a.rho[b];
    wave_sync();
    tm.forward_solve_only();
    tm.backward_sweep();
    if (active) {
        const real *zf = a.z_final + (size_t)b * T * N;
        real *qg = a.q_grad + (size_t)b * T * N;
        real *Qg = a.Qd_grad + (size_t)b * T * N;
        for (int e = li; e < T * N; e += G) {
            real w = tm.ds[e];
            qg[e] = w;
            Qg[e] = w * zf[e];
        }
    }
}

// ---- size-generic helper kernels (one wavefront per instance) ----------------------
template <typename real>
__device__ inline real wave_sum(real v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename real>
struct AuxArgs {
    int B, T, nx, nu, K, n_ls;
    const real *zc, *xnext, *x0, *lam, *rho, *Qd, *q, *ulo, *uhi;
    long sb_u, st_u;
    real *phi, *rnorm2;
    // pick
    const real *phi_all, *d;
    real *phi_prev, *z;
    int *k_out, *accept_out;
    // dual
    real *lam_io, *rho_io;
    real rho_scale;
    // obstacle rows (nullable): centres [B][T][nobs][3], radius^2
    const real *obs;
    int nobs;
    real obs_r2;
    int no_init;   // state-estimator row set: the initial-state rows (row block T-1) do not exist
};

// c_k = r^2 - |x_t[0:3] - o_k|^2 for obstacle row e = t*nobs + k of instance b (al_utils.py:313-323)
template <typename real>
__device__ inline real obs_row(const AuxArgs<real> &a, const real *z, int b, int e) {
    const int t = e / a.nobs, n = a.nx + a.nu;
    const real *o = a.obs + ((size_t)b * a.T * a.nobs + e) * 3;
    const real d0 = z[t * n] - o[0], d1 = z[t * n + 1] - o[1], d2 = z[t * n + 2] - o[2];
    return a.obs_r2 - (d0 * d0 + d1 * d1 + d2 * d2);
}

// merit of candidate kk for instance b (al_utils.py:73-77), block = (kk, b)
template <typename real>
__global__ __launch_bounds__(64) void k_merit(AuxArgs<real> a) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x % a.B, kk = blockIdx.x / a.B;
    const int T = a.T, nx = a.nx, nu = a.nu, n = nx + nu, neq = T * nx;
    const real *z = a.zc + ((size_t)kk * a.B + b) * T * n;
    const real *xn = a.xnext + ((size_t)kk * a.B + b) * (T - 1) * nx;
    const int nit = 2 * nu + a.nobs;   // inequality rows per stage
    const real *lam = a.lam + (size_t)b * (neq + T * nit);
    const real *Qd = a.Qd + (size_t)b * T * n, *q = a.q + (size_t)b * T * n;
    const real *ulo = a.ulo + (size_t)b * a.sb_u, *uhi = a.uhi + (size_t)b * a.sb_u;
    const real rho = a.rho[b];
    real acc = 0, sq = 0;
    for (int e = lane; e < T * n; e += 64) {
        int t = e / n, j = e - t * n;
        real v = z[e];
        acc += (real(0.5) * Qd[e] * v + q[e]) * v;
        if (j >= nx) {
            int ju = j - nx;
            real vu = v - uhi[t * a.st_u + ju], vl = -v + ulo[t * a.st_u + ju];
            real cu = vu > 0 ? vu : real(0), cl = vl > 0 ? vl : real(0);
            int ru = neq + t * nit + ju;
            acc += lam[ru] * vu + lam[ru + nu] * vl;
            sq += cu * cu + cl * cl;
        }
    }
    for (int e = lane; e < T * a.nobs; e += 64) {
        const real ck = obs_row(a, z, b, e), cp = ck > 0 ? ck : real(0);
        acc += lam[neq + (e / a.nobs) * nit + 2 * nu + e % a.nobs] * ck;
        sq += cp * cp;
    }
    const int neq_rows = a.no_init ? neq - nx : neq;
    for (int e = lane; e < neq_rows; e += 64) {
        int t = e / nx, i = e - t * nx;
        real r = (t < T - 1) ? z[(t + 1) * n + i] - xn[t * nx + i] : z[i] - a.x0[(size_t)b * nx + i];
        acc += lam[e] * r;
        sq += r * r;
    }
    acc = wave_sum(acc);
    sq = wave_sum(sq);
    if (lane == 0) {
        a.phi[(size_t)kk * a.B + b] = acc + real(0.5) * rho * sq;
        if (a.rnorm2) a.rnorm2[(size_t)kk * a.B + b] = sq;
    }
}

// The whole line search of one Newton step in ONE launch (nonlinear-caller mode at scale): the merits of
// the n_ls candidates z + 2^-k d (al_utils.py:618-633; x_next of every candidate was evaluated by the
// caller's dynamics: xnext_all [n_ls][B][T-1][nx]), first-argmin with NaN winning like torch.min, strict
// accept, z <- z + alpha d in place, phi_prev <- phi_min regardless (:569), rnorm2 <- sum r+^2 of the
// chosen candidate when accepted. One wavefront per instance: it reads z, d once (not the 20-fold stack
// the reference materialises) and its 20 x_next slabs; everything else comes from registers.
template <typename real>
__global__ __launch_bounds__(64) void k_merit_pick(AuxArgs<real> a) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const int T = a.T, nx = a.nx, nu = a.nu, n = nx + nu, neq = T * nx, nit = 2 * nu + a.nobs;
    real *z = a.z + (size_t)b * T * n;
    const real *d = a.d + (size_t)b * T * n;
    const real *lam = a.lam + (size_t)b * (neq + T * nit);
    const real *Qd = a.Qd + (size_t)b * T * n, *q = a.q + (size_t)b * T * n;
    const real *ulo = a.ulo + (size_t)b * a.sb_u, *uhi = a.uhi + (size_t)b * a.sb_u;
    const real rho = a.rho[b];
    real acc[20], sq[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) { acc[k] = 0; sq[k] = 0; }
    // cost + bound rows: every lane walks its elements once, all candidates from registers
    for (int e = lane; e < T * n; e += 64) {
        const int t = e / n, j = e - t * n;
        const real zv = z[e], dv = d[e], Qv = Qd[e], qv = q[e];
        real bu = 0, bl = 0, lu = 0, ll = 0;
        const bool isu = j >= nx;
        if (isu) {
            bu = uhi[t * a.st_u + j - nx]; bl = ulo[t * a.st_u + j - nx];
            lu = lam[neq + t * nit + j - nx]; ll = lam[neq + t * nit + nu + j - nx];
        }
        real alpha = 1;
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            const real v = zv + alpha * dv;
            acc[k] += (real(0.5) * Qv * v + qv) * v;
            if (isu) {
                const real vu = v - bu, vl = -v + bl;
                const real cu = vu > 0 ? vu : real(0), cl = vl > 0 ? vl : real(0);
                acc[k] += lu * vu + ll * vl;
                sq[k] += cu * cu + cl * cl;
            }
            alpha *= real(0.5);
        }
    }
    // equality rows: r = x_{t+1}(candidate) - xnext_k, init rows x_0 - x0
    for (int e = lane; e < (a.no_init ? neq - nx : neq); e += 64) {
        const int t = e / nx, i = e - t * nx;
        const real le = lam[e];
        const int zi = (t < T - 1) ? (t + 1) * n + i : i;
        const real zv = z[zi], dv = d[zi];
        const real x0v = (t < T - 1) ? real(0) : a.x0[(size_t)b * nx + i];
        real alpha = 1;
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            if (k < a.n_ls) {
                const real ref = (t < T - 1) ? a.xnext[(((size_t)k * a.B + b) * (T - 1) + t) * nx + i] : x0v;
                const real r = zv + alpha * dv - ref;
                acc[k] += le * r;
                sq[k] += r * r;
            }
            alpha *= real(0.5);
        }
    }
    // obstacle rows (Obstacle_MPC): c_k at the candidate's position
    for (int e = lane; e < T * a.nobs; e += 64) {
        const int t = e / a.nobs;
        const real *o = a.obs + ((size_t)b * T * a.nobs + e) * 3;
        const real lk = lam[neq + t * nit + 2 * nu + e % a.nobs];
        real alpha = 1;
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            const real d0 = z[t * n] + alpha * d[t * n] - o[0], d1 = z[t * n + 1] + alpha * d[t * n + 1] - o[1],
                       d2 = z[t * n + 2] + alpha * d[t * n + 2] - o[2];
            const real ck = a.obs_r2 - (d0 * d0 + d1 * d1 + d2 * d2), cp = ck > 0 ? ck : real(0);
            acc[k] += lk * ck;
            sq[k] += cp * cp;
            alpha *= real(0.5);
        }
    }
    int kbest = 0;
    real best = 0, sqbest = 0;
#pragma unroll
    for (int k = 0; k < 20; ++k) {
        if (k < a.n_ls) {
            const real s2 = wave_sum(sq[k]);
            const real v = wave_sum(acc[k]) + real(0.5) * rho * s2;
            if (a.phi) a.phi[(size_t)k * a.B + b] = v;     // (all lanes hold the same value)
            if (k == 0) { best = v; sqbest = s2; }
            else if (!(best != best) && (v != v || v < best)) { best = v; kbest = k; sqbest = s2; }
        }
    }
    const real prev = a.phi_prev[b];
    const bool ok = best < prev;
    if (ok) {
        const real alpha = real(1) / real(1 << kbest);
        for (int e = lane; e < T * n; e += 64) z[e] += alpha * d[e];
    }
    if (lane == 0) {
        a.phi_prev[b] = best;
        if (a.k_out) a.k_out[b] = kbest;
        if (a.accept_out) a.accept_out[b] = ok ? 1 : 0;
        if (a.rnorm2 && ok) a.rnorm2[b] = sqbest;
    }
}

// line-search decision + update (al_utils.py:634-641), block = instance
template <typename real>
__global__ __launch_bounds__(64) void k_pick(AuxArgs<real> a) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const int Tn = a.T * (a.nx + a.nu);
    int kbest = 0;
    real best = a.phi_all[b];
    for (int k = 1; k < a.n_ls; ++k) {
        real v = a.phi_all[(size_t)k * a.B + b];
        if (!(best != best) && (v != v || v < best)) { best = v; kbest = k; }
    }
    const real prev = a.phi_prev[b];
    const bool acc = best < prev;
    const real alpha = acc ? real(1) / real(1 << kbest) : real(0);
    real *z = a.z + (size_t)b * Tn;
    const real *d = a.d + (size_t)b * Tn;
    if (acc)
        for (int e = lane; e < Tn; e += 64) z[e] += alpha * d[e];
    __syncthreads();
    if (lane == 0) {
        a.phi_prev[b] = best;
        if (a.k_out) a.k_out[b] = kbest;
        if (a.accept_out) a.accept_out[b] = acc ? 1 : 0;
    }
}

// dual update + projection + rho growth (AL_mpc.py:315-317,325), block = instance
template <typename real>
__global__ __launch_bounds__(64) void k_dual(AuxArgs<real> a) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const int T = a.T, nx = a.nx, nu = a.nu, n = nx + nu, neq = T * nx;
    const real *z = a.zc + (size_t)b * T * n;
    const real *xn = a.xnext + (size_t)b * (T - 1) * nx;
    const int nit = 2 * nu + a.nobs;
    real *lam = a.lam_io + (size_t)b * (neq + T * nit);
    const real *ulo = a.ulo + (size_t)b * a.sb_u, *uhi = a.uhi + (size_t)b * a.sb_u;
    const real rho = a.rho_io[b];
    for (int e = lane; e < (a.no_init ? neq - nx : neq); e += 64) {
        int t = e / nx, i = e - t * nx;
        real r = (t < T - 1) ? z[(t + 1) * n + i] - xn[t * nx + i] : z[i] - a.x0[(size_t)b * nx + i];
        lam[e] += rho * r;
    }
    for (int e = lane; e < T * nu; e += 64) {
        int t = e / nu, j = e - t * nu;
        real u = z[t * n + nx + j];
        int ru = neq + t * nit + j, rl = ru + nu;
        real v1 = lam[ru] + rho * (u - uhi[t * a.st_u + j]);
        real v2 = lam[rl] + rho * (-u + ulo[t * a.st_u + j]);
        lam[ru] = v1 < 0 ? real(0) : v1;
        lam[rl] = v2 < 0 ? real(0) : v2;
    }
    for (int e = lane; e < T * a.nobs; e += 64) {
        const int r = neq + (e / a.nobs) * nit + 2 * nu + e % a.nobs;
        const real v = lam[r] + rho * obs_row(a, z, b, e);
        lam[r] = v < 0 ? real(0) : v;
    }
    __syncthreads();
    if (lane == 0) a.rho_io[b] = rho * a.rho_scale;
}

#endif  // ALQP_BUILD_MAIN

// ---- dispatch -------------------------------------------------------------------------

// (nx, nu) instances compiled into the library: alqp_dims.hpp (shared with alqp_ipm.hip).

constexpr size_t kMaxLds = 160 * 1024;

// A launch whose first argument carries ALQP_EXIT_IN_KERNEL goes out as a COOPERATIVE launch (the kernel then uses
// grid-wide barriers; the runtime refuses grids that cannot be co-resident: ALQP_E_COOP, the caller falls back to the
// launch-per-step route); everything else as a plain launch.
template <typename T>
inline int flags_of(const T &) { return 0; }
template <typename real>
inline int flags_of(const SolveArgs<real> &a) { return a.flags; }
template <typename Fn, typename A0, typename... Rest>
int launch_maybe_coop(Fn fn, unsigned grid, size_t lds, hipStream_t stream, A0 a0, Rest... rest) {
    if (flags_of(a0) & ALQP_EXIT_IN_KERNEL) {
        void *argv[] = {(void *)&a0, (void *)&rest...};
        // Any refusal of the cooperative launch (grid too large, cooperative launches not supported by the device or the
        // queue, ...) is ALQP_E_COOP: the caller then takes the launch-per-step route, which needs no co-residency.
        static int coop_ok = -1;
        if (coop_ok < 0) {
            int dev = 0, v = 0;
            coop_ok = (hipGetDevice(&dev) == hipSuccess &&
                       hipDeviceGetAttribute(&v, hipDeviceAttributeCooperativeLaunch, dev) == hipSuccess && v) ? 1 : 0;
        }
        if (!coop_ok) return ALQP_E_COOP;
        hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(fn), dim3(grid), dim3(64), argv, (unsigned)lds, stream);
        if (e != hipSuccess) { (void)hipGetLastError(); return ALQP_E_COOP; }
        return 0;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, stream, a0, rest...);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}


#if ALQP_BUILD_MAIN
template <typename real, int NX, int NU>
size_t lds_bytes_for(int T) {
    using C = Cfg<real, NX, NU>;
    return (size_t)C::QPW * C::team_words(T) * sizeof(real);
}

// Launches `fn` with one wavefront per workgroup and the team LDS image as dynamic LDS.
template <typename real, int NX, int NU, typename Fn, typename... Args>
int launch_team_kernel(Fn fn, int B, int T, hipStream_t stream, Args... args) {
    using C = Cfg<real, NX, NU>;
    const size_t lds = lds_bytes_for<real, NX, NU>(T);
    if (lds > kMaxLds) return ALQP_E_UNSUPPORTED;
    const unsigned grid = (unsigned)((B + C::QPW - 1) / C::QPW);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ALQP_E_LAUNCH;
    }
    return launch_maybe_coop(fn, grid, lds, stream, args...);
}

inline long team_simds() {   // SIMDs of the device (4 per CU)
    static long n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        n = 4L * cus;
    }
    return n;
}

template <typename real>
int dispatch_solve(int nx, int nu, const SolveArgs<real> &a, const TraceArgs<real> *tr, hipStream_t stream) {
#define X(NX, NU)                                                                                     \
    if (nx == NX && nu == NU) {                                                                       \
        if (tr) return launch_team_kernel<real, NX, NU>(k_solve_lin<real, NX, NU, true>, a.B, a.T, stream, a, *tr); \
        const long waves = (a.B + Cfg<real, NX, NU>::QPW - 1) / Cfg<real, NX, NU>::QPW;                \
        if (sizeof(real) == 8 || waves <= team_simds())                                                 \
            return launch_team_kernel<real, NX, NU>(k_solve_lin<real, NX, NU, false, 1>, a.B, a.T, stream, a, TraceArgs<real>{}); \
        if constexpr (sizeof(real) == 4)                                                                \
            return launch_team_kernel<real, NX, NU>(k_solve_lin<real, NX, NU, false, 2>, a.B, a.T, stream, a, TraceArgs<real>{}); \
    }
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return ALQP_E_UNSUPPORTED;
}

#endif  // ALQP_BUILD_MAIN

template <typename real>
int dispatch_solve_quad(int nx, int nu, const SolveArgs<real> &a, const TraceArgs<real> *tr, real *ws,
                        hipStream_t stream);
template <typename real>
int dispatch_backward_quad(int nx, int nu, const BwdArgs<real> &a, real *ws, hipStream_t stream);
template <typename real>
int dispatch_solve_nonlin(int dyn_id, int nx, int nu, const SolveArgs<real> &a, real *ws, hipStream_t stream);
template <typename real>
int dispatch_step_quad(int nx, int nu, const StepArgs<real> &a, real *ws, hipStream_t stream);

#if ALQP_BUILD_QUAD
template <typename real, int NX, int NU, typename Fn, typename... Args>
int launch_quad_kernel(Fn fn, int B, hipStream_t stream, Args... args) {
    const unsigned grid = (unsigned)((B + 15) / 16);
    return launch_maybe_coop(fn, grid, 0, stream, args...);
}

template <typename real>
int dispatch_solve_quad(int nx, int nu, const SolveArgs<real> &a, const TraceArgs<real> *tr, real *ws,
                        hipStream_t stream) {
#define X(NX, NU)                                                                                       \
    if (nx == NX && nu == NU) {                                                                         \
        if (tr) return launch_quad_kernel<real, NX, NU>(k_solve_lin_quad<real, NX, NU, true>, a.B, stream, a, *tr, ws); \
        return launch_quad_kernel<real, NX, NU>(k_solve_lin_quad<real, NX, NU, false>, a.B, stream, a, TraceArgs<real>{}, ws); \
    }
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return ALQP_E_UNSUPPORTED;
}

template <typename real>
int dispatch_backward_quad(int nx, int nu, const BwdArgs<real> &a, real *ws, hipStream_t stream) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return launch_quad_kernel<real, NX, NU>(k_backward_quad<real, NX, NU>, a.B, stream, a, ws);
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return ALQP_E_UNSUPPORTED;
}

// nonlinear fused solve: the models of alqp_dyn.hpp, each with its own (nx, nu)
template <typename real>
int dispatch_solve_nonlin(int dyn_id, int nx, int nu, const SolveArgs<real> &a, real *ws, hipStream_t stream) {
    if (dyn_id == DynPendulum1l<real>::ID && nx == 2 && nu == 1)
        return launch_quad_kernel<real, 2, 1>(k_solve_lin_quad<real, 2, 1, false, DynPendulum1l<real>>, a.B, stream, a,
                                              TraceArgs<real>{}, ws);
    if (dyn_id == DynCartpole1l<real>::ID && nx == 4 && nu == 1)
        return launch_quad_kernel<real, 4, 1>(k_solve_lin_quad<real, 4, 1, false, DynCartpole1l<real>>, a.B, stream, a,
                                              TraceArgs<real>{}, ws);
    if (dyn_id == DynCartpole1l<real, 2>::ID && nx == 4 && nu == 1)
        return launch_quad_kernel<real, 4, 1>(k_solve_lin_quad<real, 4, 1, false, DynCartpole1l<real, 2>>, a.B, stream, a,
                                              TraceArgs<real>{}, ws);
    if (dyn_id == DynCartpole2l<real>::ID && nx == 6 && nu == 1)
        return launch_quad_kernel<real, 6, 1>(k_solve_lin_quad<real, 6, 1, false, DynCartpole2l<real>>, a.B, stream, a,
                                              TraceArgs<real>{}, ws);
    return ALQP_E_UNSUPPORTED;
}

template <typename real>
int dispatch_step_quad(int nx, int nu, const StepArgs<real> &a, real *ws, hipStream_t stream) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return launch_quad_kernel<real, NX, NU>(k_newton_step_quad<real, NX, NU>, a.B, stream, a, ws);
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return ALQP_E_UNSUPPORTED;
}

#if ALQP_QUAD_F32
template int dispatch_step_quad<float>(int, int, const StepArgs<float> &, float *, hipStream_t);
#endif
#if ALQP_QUAD_F64
template int dispatch_step_quad<double>(int, int, const StepArgs<double> &, double *, hipStream_t);
#endif

#if ALQP_QUAD_F32
template int dispatch_solve_nonlin<float>(int, int, int, const SolveArgs<float> &, float *, hipStream_t);
template int dispatch_solve_quad<float>(int, int, const SolveArgs<float> &, const TraceArgs<float> *, float *, hipStream_t);
template int dispatch_backward_quad<float>(int, int, const BwdArgs<float> &, float *, hipStream_t);
#endif
#if ALQP_QUAD_F64
template int dispatch_solve_nonlin<double>(int, int, int, const SolveArgs<double> &, double *, hipStream_t);
template int dispatch_solve_quad<double>(int, int, const SolveArgs<double> &, const TraceArgs<double> *, double *, hipStream_t);
template int dispatch_backward_quad<double>(int, int, const BwdArgs<double> &, double *, hipStream_t);
#endif

#endif  // ALQP_BUILD_QUAD

template <typename real>
size_t quad_ws_bytes(int nx, int nu, int B, int T) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return QCfg<real, NX, NU>::ws_covers(B, T) ? QCfg<real, NX, NU>::ws_words(B, T) * sizeof(real) : 0;
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return 0;
}

#if ALQP_BUILD_MAIN
template <typename real>
int dispatch_step(int nx, int nu, const StepArgs<real> &a, hipStream_t stream) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return launch_team_kernel<real, NX, NU>(k_newton_step<real, NX, NU>, a.B, a.T, stream, a);
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return ALQP_E_UNSUPPORTED;
}

template <typename real>
int dispatch_backward(int nx, int nu, const BwdArgs<real> &a, hipStream_t stream) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return launch_team_kernel<real, NX, NU>(k_backward<real, NX, NU>, a.B, a.T, stream, a);
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return ALQP_E_UNSUPPORTED;
}

template <typename real>
size_t lds_query(int nx, int nu, int T) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return lds_bytes_for<real, NX, NU>(T);
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return 0;
}

static int qpw_query(int nx, int nu) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return Cfg<float, NX, NU>::QPW;
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return 0;
}

static bool dims_ok(const AlqpDims *d) { return d && d->B > 0 && d->T >= 2 && d->nx > 0 && d->nu > 0; }

// ---- start offset between the wavefronts of a CU for the quad solve (SolveArgs::stagger) -------------------
// mode = AlqpParams.quad_stagger: 0 automatic, < 0 off, > 0 explicit units of ~1024 clocks (a per-call argument: the library
// keeps no process state)
static int quad_stagger(int mode, int B, int T, int nx, int nu, int newton_steps, bool f64) {
    if (mode < 0) return 0;
    if (mode > 0) return mode;
    static int n_simd = 0;
    if (n_simd == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        n_simd = 4 * cus;
    }
    const long waves = (B + 15) / 16;
    if (4 * waves < 3 * (long)n_simd || newton_steps < 1) return 0;   // SIMDs not filled: nothing to de-phase
    // measured per size at B = 16384 (profiles/r02/experiments): (13,4) T=20 +1.7 %, T=50 +1.2 %, (14,4) +1.7 %,
    // (6,2) +0.8 %, (8,2) -0.3 %, (2,1) at B = 65536 -2.5 %: only the sizes with long stages gain
    if (nx + nu < 12) return 0;
    // clocks per stage and sweep, fitted on the compiled sizes ((13,4): 26 k, (8,2): 12 k, (6,2): 7 k)
    const int n = nx + nu;
    double period = 2000.0 * n - 8000.0;
    if (period < 3000.0) period = 3000.0;
    if (f64) period *= 1.4;   // fp64 sweeps take 2.4x as long; measured: 140 units beat 100 at (13,4), T = 20
    // a fifth of a sweep between neighbouring SIMDs (measured optimum at (13,4), T = 20: 100-125 units), but the
    // last wave's delay (3 offsets) stays below ~6 % of the launch
    double frac = 0.04 * newton_steps;
    if (frac > 0.2) frac = 0.2;
    return (int)(frac * T * period / 1024.0 + 0.5);
}

template <typename real>
int solve_lin_impl(const AlqpDims *dims, const AlqpParams *prm, const void *Qd, const void *q,
                   const void *F, const void *c, const void *x0, const void *u_lo, const void *u_hi,
                   long sb_u, long st_u, void *z, void *lam, void *rho, void *phi, void *rnorm2,
                   int *info, unsigned char *status, void *factor_out, const AlqpTrace *trace,
                   void *workspace, size_t ws_bytes, void *stream) {
    if (!dims_ok(dims) || !prm || !Qd || !q || !F || !c || !x0 || !u_lo || !u_hi || !z || !lam || !rho || !phi)
        return ALQP_E_BADARG;
    if (prm->n_ls < 1 || prm->n_ls > 20 || prm->al_iter < 0 || prm->max_newton < 0) return ALQP_E_BADARG;
    if ((prm->flags & ALQP_SAVE_FACTOR) && !factor_out) return ALQP_E_BADARG;
    SolveArgs<real> a = {};
    a.B = dims->B; a.T = dims->T;
    a.al_iter = prm->al_iter; a.max_newton = prm->max_newton; a.n_ls = prm->n_ls; a.flags = prm->flags;
    a.rho_scale = (real)prm->rho_scale;
    a.Qd = (const real *)Qd; a.q = (const real *)q; a.F = (const real *)F; a.c = (const real *)c;
    a.x0 = (const real *)x0; a.ulo = (const real *)u_lo; a.uhi = (const real *)u_hi;
    a.sb_u = sb_u; a.st_u = st_u;
    a.z = (real *)z; a.lam = (real *)lam; a.rho = (real *)rho; a.phi = (real *)phi;
    a.rnorm2 = (real *)rnorm2; a.info = info; a.status = status; a.factor = (real *)factor_out;
    a.skip = prm->skip_flag;
    a.stagger = quad_stagger(prm->quad_stagger, dims->B, dims->T, dims->nx, dims->nu, prm->al_iter * prm->max_newton, sizeof(real) == 8);
    if (prm->flags & ALQP_EXIT_IN_KERNEL) {
        if (!prm->exit_scratch || trace || prm->skip_flag) return ALQP_E_BADARG;
        a.exit_tol = prm->exit_tol; a.newton_counts = prm->newton_counts; a.exit_scratch = prm->exit_scratch;
    }
    TraceArgs<real> tr = {};
    if (trace) {
        tr.g = (real *)trace->g; tr.d = (real *)trace->d; tr.phi = (real *)trace->phi;
        tr.phi_prev = (real *)trace->phi_prev; tr.k = trace->k; tr.accept = trace->accept;
    }
    // variant: 1 = team (factor in LDS), 2 = quad (4 lanes/instance, HBM workspace), 0 = auto
    const size_t need = quad_ws_bytes<real>(dims->nx, dims->nu, dims->B, dims->T);
    int variant = prm->variant;
    // auto: quad once the batch fills the chip (16 instances per wavefront, 1024 SIMDs), team below
    // (2-2.4x lower latency at small batches). The team kernels' time is a step function of the batch - 2048 (fp32) /
    // 1024 (fp64) teams fit the chip at once - and B = 4096 is exactly two / four full rounds: measured at (13,4) T=20
    // after round 3's team-kernel work, fp32 B = 4096 team 1.83 vs quad 2.00 ms, B = 5120 2.36 vs 2.08; fp64 B = 4096 4.35
    // vs 4.71, B = 5120 5.42 vs 5.02 (profiles/r03/experiments/README.md).
    if (variant == 0) {
        const size_t team_lds = lds_query<real>(dims->nx, dims->nu, dims->T);
        const bool team_fits = team_lds > 0 && team_lds <= kMaxLds;
        const bool quad_ok = need > 0 && workspace && ws_bytes >= need && !(prm->flags & ALQP_SAVE_FACTOR);
        // long horizons whose factor does not fit the team's LDS image run on the quad kernels at any batch
        // whole-wavefront teams (2n + nx + 1 > 32 rows, e.g. (13,4)): team through B = 4096 (full rounds), quad beyond;
        // smaller teams share a wavefront and were not re-measured: round 2's rule
        const bool wave_team = 2 * (dims->nx + dims->nu) + dims->nx + 1 > 32;
        const int qmin = wave_team ? 4097 : (sizeof(real) == 8 ? 4608 : 4096);
        variant = (quad_ok && (dims->B >= qmin || !team_fits)) ? 2 : 1;
    }
    if (variant == 2) {
        if (prm->flags & ALQP_SAVE_FACTOR) return ALQP_E_UNSUPPORTED;
        if (need == 0) return ALQP_E_UNSUPPORTED;
        if (!workspace || ws_bytes < need) return ALQP_E_BADARG;
        return dispatch_solve_quad<real>(dims->nx, dims->nu, a, trace ? &tr : nullptr, (real *)workspace,
                                         (hipStream_t)stream);
    }
    if (variant != 1) return ALQP_E_BADARG;
    return dispatch_solve<real>(dims->nx, dims->nu, a, trace ? &tr : nullptr, (hipStream_t)stream);
}

// nonlinear fused solve: workspace = [records | F linearisations [B][T-1][nx][n]]
template <typename real>
size_t nonlin_ws_bytes(int nx, int nu, int B, int T) {
    const size_t rec = quad_ws_bytes<real>(nx, nu, B, T);
    if (rec == 0) return 0;
    return rec + (size_t)B * (T - 1) * nx * (nx + nu) * sizeof(real);
}

template <typename real>
int solve_nonlin_impl(const AlqpDims *dims, const AlqpParams *prm, int dyn_id, double dyn_h, const void *Qd, const void *q,
                      const void *x0, const void *u_lo, const void *u_hi, long sb_u, long st_u, void *z, void *lam,
                      void *rho, void *phi, void *rnorm2, int *info, unsigned char *status, void *workspace,
                      size_t ws_bytes, void *stream) {
    if (!dims_ok(dims) || !prm || !Qd || !q || !x0 || !u_lo || !u_hi || !z || !lam || !rho || !phi || !workspace)
        return ALQP_E_BADARG;
    if (prm->n_ls != 20 || prm->al_iter < 0 || prm->max_newton < 0) return ALQP_E_BADARG;
    if (prm->flags & (ALQP_SAVE_FACTOR | ALQP_WS_PRIMED)) return ALQP_E_UNSUPPORTED;
    const size_t need = nonlin_ws_bytes<real>(dims->nx, dims->nu, dims->B, dims->T);
    if (need == 0) return ALQP_E_UNSUPPORTED;
    if (ws_bytes < need) return ALQP_E_BADARG;
    SolveArgs<real> a = {};
    a.B = dims->B; a.T = dims->T;
    a.al_iter = prm->al_iter; a.max_newton = prm->max_newton; a.n_ls = prm->n_ls; a.flags = prm->flags;
    a.rho_scale = (real)prm->rho_scale;
    a.Qd = (const real *)Qd; a.q = (const real *)q; a.c = nullptr; a.x0 = (const real *)x0;
    a.F = (const real *)((const char *)workspace + quad_ws_bytes<real>(dims->nx, dims->nu, dims->B, dims->T));
    a.ulo = (const real *)u_lo; a.uhi = (const real *)u_hi; a.sb_u = sb_u; a.st_u = st_u;
    a.z = (real *)z; a.lam = (real *)lam; a.rho = (real *)rho; a.phi = (real *)phi;
    a.rnorm2 = (real *)rnorm2; a.info = info; a.status = status; a.factor = nullptr;
    a.skip = prm->skip_flag;
    a.dyn_h = (real)dyn_h;
    if (prm->flags & ALQP_EXIT_IN_KERNEL) {
        if (!prm->exit_scratch || prm->skip_flag) return ALQP_E_BADARG;
        a.exit_tol = prm->exit_tol; a.newton_counts = prm->newton_counts; a.exit_scratch = prm->exit_scratch;
    }
    return dispatch_solve_nonlin<real>(dyn_id, dims->nx, dims->nu, a, (real *)workspace, (hipStream_t)stream);
}

template <typename real>
int newton_step_impl(const AlqpDims *dims, const void *z, const void *xnext, const void *F,
                     const void *x0, const void *lam, const void *rho, const void *Qd, const void *q,
                     const void *u_lo, const void *u_hi, long sb_u, long st_u, void *d_out,
                     void *g_out, void *factor_out, int *info, void *stream, const AlqpObstacles *obs = nullptr,
                     void *workspace = nullptr, size_t ws_bytes = 0) {
    if (!dims_ok(dims) || !z || !xnext || !F || !x0 || !lam || !rho || !Qd || !q || !u_lo || !u_hi || !d_out)
        return ALQP_E_BADARG;
    if (obs && (obs->nobs < 0 || (obs->nobs > 0 && (!obs->pos || dims->nx < 3)))) return ALQP_E_BADARG;
    StepArgs<real> a = {};
    if (obs && obs->nobs > 0) { a.obs = (const real *)obs->pos; a.nobs = obs->nobs; a.obs_r2 = (real)(obs->radius * obs->radius); }
    if (obs) a.no_init = obs->state_estimator;
    a.B = dims->B; a.T = dims->T;
    a.z = (const real *)z; a.xnext = (const real *)xnext; a.F = (const real *)F; a.x0 = (const real *)x0;
    a.lam = (const real *)lam; a.rho = (const real *)rho; a.Qd = (const real *)Qd; a.q = (const real *)q;
    a.ulo = (const real *)u_lo; a.uhi = (const real *)u_hi; a.sb_u = sb_u; a.st_u = st_u;
    a.d_out = (real *)d_out; a.g_out = (real *)g_out; a.factor = (real *)factor_out; a.info = info;
    if (workspace) {   // quad variant: the factor stays in the workspace records (alqp_backward_ws)
        if (factor_out) return ALQP_E_BADARG;
        const size_t need = quad_ws_bytes<real>(dims->nx, dims->nu, dims->B, dims->T);
        if (need == 0) return ALQP_E_UNSUPPORTED;
        if (ws_bytes < need) return ALQP_E_BADARG;
        return dispatch_step_quad<real>(dims->nx, dims->nu, a, (real *)workspace, (hipStream_t)stream);
    }
    return dispatch_step<real>(dims->nx, dims->nu, a, (hipStream_t)stream);
}

template <typename real>
int backward_impl(const AlqpDims *dims, const void *factor, const void *F, const void *rho,
                  const void *z_final, const void *gbar, void *q_grad, void *Qd_grad, void *stream) {
    if (!dims_ok(dims) || !factor || !F || !rho || !z_final || !gbar || !q_grad || !Qd_grad)
        return ALQP_E_BADARG;
    BwdArgs<real> a = {};
    a.B = dims->B; a.T = dims->T;
    a.factor = (const real *)factor; a.F = (const real *)F; a.rho = (const real *)rho;
    a.z_final = (const real *)z_final; a.gbar = (const real *)gbar;
    a.q_grad = (real *)q_grad; a.Qd_grad = (real *)Qd_grad;
    return dispatch_backward<real>(dims->nx, dims->nu, a, (hipStream_t)stream);
}

template <typename real>
int backward_ws_impl(const AlqpDims *dims, void *workspace, size_t ws_bytes, const void *F, const void *rho,
                     const void *z_final, const void *gbar, void *q_grad, void *Qd_grad, void *stream) {
    if (!dims_ok(dims) || !workspace || !F || !rho || !z_final || !gbar || !q_grad || !Qd_grad)
        return ALQP_E_BADARG;
    const size_t need = quad_ws_bytes<real>(dims->nx, dims->nu, dims->B, dims->T);
    if (need == 0) return ALQP_E_UNSUPPORTED;
    if (ws_bytes < need) return ALQP_E_BADARG;
    BwdArgs<real> a = {};
    a.B = dims->B; a.T = dims->T;
    a.F = (const real *)F; a.rho = (const real *)rho;
    a.z_final = (const real *)z_final; a.gbar = (const real *)gbar;
    a.q_grad = (real *)q_grad; a.Qd_grad = (real *)Qd_grad;
    return dispatch_backward_quad<real>(dims->nx, dims->nu, a, (real *)workspace, (hipStream_t)stream);
}

template <typename real>
int merit_impl(const AlqpDims *dims, int K, const void *zc, const void *xnext, const void *x0,
               const void *lam, const void *rho, const void *Qd, const void *q, const void *u_lo,
               const void *u_hi, long sb_u, long st_u, void *phi, void *rnorm2, void *stream,
               const AlqpObstacles *obs = nullptr) {
    if (!dims_ok(dims) || K < 1 || !zc || !xnext || !x0 || !lam || !rho || !Qd || !q || !u_lo || !u_hi || !phi)
        return ALQP_E_BADARG;
    if (obs && (obs->nobs < 0 || (obs->nobs > 0 && (!obs->pos || dims->nx < 3)))) return ALQP_E_BADARG;
    AuxArgs<real> a = {};
    if (obs && obs->nobs > 0) { a.obs = (const real *)obs->pos; a.nobs = obs->nobs; a.obs_r2 = (real)(obs->radius * obs->radius); }
    if (obs) a.no_init = obs->state_estimator;
    a.B = dims->B; a.T = dims->T; a.nx = dims->nx; a.nu = dims->nu; a.K = K;
    a.zc = (const real *)zc; a.xnext = (const real *)xnext; a.x0 = (const real *)x0;
    a.lam = (const real *)lam; a.rho = (const real *)rho; a.Qd = (const real *)Qd; a.q = (const real *)q;
    a.ulo = (const real *)u_lo; a.uhi = (const real *)u_hi; a.sb_u = sb_u; a.st_u = st_u;
    a.phi = (real *)phi; a.rnorm2 = (real *)rnorm2;
    hipLaunchKernelGGL(k_merit<real>, dim3((unsigned)((size_t)K * dims->B)), dim3(64), 0,
                       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

template <typename real>
int merit_pick_impl(const AlqpDims *dims, int n_ls, const void *d, const void *xnext_all, const void *x0,
                    const void *lam, const void *rho, const void *Qd, const void *q, const void *u_lo,
                    const void *u_hi, long sb_u, long st_u, const AlqpObstacles *obs, void *z, void *phi_prev,
                    void *rnorm2, void *phi_all, int *k_out, int *accept_out, void *stream) {
    if (!dims_ok(dims) || n_ls < 1 || n_ls > 20 || !d || !xnext_all || !x0 || !lam || !rho || !Qd || !q || !u_lo ||
        !u_hi || !z || !phi_prev)
        return ALQP_E_BADARG;
    if (obs && (obs->nobs < 0 || (obs->nobs > 0 && (!obs->pos || dims->nx < 3)))) return ALQP_E_BADARG;
    AuxArgs<real> a = {};
    if (obs && obs->nobs > 0) { a.obs = (const real *)obs->pos; a.nobs = obs->nobs; a.obs_r2 = (real)(obs->radius * obs->radius); }
    if (obs) a.no_init = obs->state_estimator;
    a.B = dims->B; a.T = dims->T; a.nx = dims->nx; a.nu = dims->nu; a.n_ls = n_ls;
    a.d = (const real *)d; a.xnext = (const real *)xnext_all; a.x0 = (const real *)x0;
    a.lam = (const real *)lam; a.rho = (const real *)rho; a.Qd = (const real *)Qd; a.q = (const real *)q;
    a.ulo = (const real *)u_lo; a.uhi = (const real *)u_hi; a.sb_u = sb_u; a.st_u = st_u;
    a.z = (real *)z; a.phi_prev = (real *)phi_prev; a.rnorm2 = (real *)rnorm2; a.phi = (real *)phi_all;
    a.k_out = k_out; a.accept_out = accept_out;
    hipLaunchKernelGGL(k_merit_pick<real>, dim3(dims->B), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

template <typename real>
int pick_impl(const AlqpDims *dims, int n_ls, const void *phi, void *phi_prev, const void *d, void *z,
              int *k_out, int *accept_out, void *stream) {
    if (!dims_ok(dims) || n_ls < 1 || !phi || !phi_prev || !d || !z) return ALQP_E_BADARG;
    AuxArgs<real> a = {};
    a.B = dims->B; a.T = dims->T; a.nx = dims->nx; a.nu = dims->nu; a.n_ls = n_ls;
    a.phi_all = (const real *)phi; a.phi_prev = (real *)phi_prev; a.d = (const real *)d; a.z = (real *)z;
    a.k_out = k_out; a.accept_out = accept_out;
    hipLaunchKernelGGL(k_pick<real>, dim3(dims->B), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

template <typename real>
int dual_impl(const AlqpDims *dims, const void *z, const void *xnext, const void *x0, const void *u_lo,
              const void *u_hi, long sb_u, long st_u, void *lam, void *rho, double rho_scale,
              void *stream, const AlqpObstacles *obs = nullptr) {
    if (!dims_ok(dims) || !z || !xnext || !x0 || !u_lo || !u_hi || !lam || !rho) return ALQP_E_BADARG;
    if (obs && (obs->nobs < 0 || (obs->nobs > 0 && (!obs->pos || dims->nx < 3)))) return ALQP_E_BADARG;
    AuxArgs<real> a = {};
    if (obs && obs->nobs > 0) { a.obs = (const real *)obs->pos; a.nobs = obs->nobs; a.obs_r2 = (real)(obs->radius * obs->radius); }
    if (obs) a.no_init = obs->state_estimator;
    a.B = dims->B; a.T = dims->T; a.nx = dims->nx; a.nu = dims->nu;
    a.zc = (const real *)z; a.xnext = (const real *)xnext; a.x0 = (const real *)x0;
    a.ulo = (const real *)u_lo; a.uhi = (const real *)u_hi; a.sb_u = sb_u; a.st_u = st_u;
    a.lam_io = (real *)lam; a.rho_io = (real *)rho; a.rho_scale = (real)rho_scale;
    hipLaunchKernelGGL(k_dual<real>, dim3(dims->B), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

// ---- dynamics provider: pendulum1l (deqmpc/my_envs/pendulum1l/src/generated_dynamics.c:55-140,
//      generated_derivatives.c:52-222). One RK4 step of theta'' = 4 tau - 19.62 sin(theta) with the
//      tangents w.r.t. (theta, omega, tau) carried along; one point per lane, outputs in the
//      solver's packed layout (x_next, F = [A | B] row-major 2x3). HBM-bound: 12 words per point.
template <typename real>
struct Dual3 {
    real v, d0, d1, d2;
};
template <typename real>
__device__ __forceinline__ Dual3<real> dadd(Dual3<real> a, real s, Dual3<real> b) {  // a + s b
    return {fma_(s, b.v, a.v), fma_(s, b.d0, a.d0), fma_(s, b.d1, a.d1), fma_(s, b.d2, a.d2)};
}
template <typename real>
__device__ __forceinline__ Dual3<real> pend_acc(Dual3<real> th, Dual3<real> ta) {
    const real sn = sin(th.v), cs = cos(th.v);
    const real kt = real(4), kg = real(19.62);
    return {kt * ta.v - kg * sn, kt * ta.d0 - kg * cs * th.d0, kt * ta.d1 - kg * cs * th.d1, kt * ta.d2 - kg * cs * th.d2};
}
template <typename real>
__global__ __launch_bounds__(256) void k_dyn_pendulum1l(long K, const real *x, const real *u, real h, const real *hpt, real *xn, real *F) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    if (hpt) h = hpt[i];
    using D = Dual3<real>;
    const D th = {x[2 * i], 1, 0, 0}, om = {x[2 * i + 1], 0, 1, 0}, ta = {u[i], 0, 0, 1};
    const real hh = real(0.5) * h;
    const D k1t = om, k1o = pend_acc(th, ta);
    const D om2 = dadd(om, hh, k1o), k2o = pend_acc(dadd(th, hh, k1t), ta);
    const D om3 = dadd(om, hh, k2o), k3o = pend_acc(dadd(th, hh, om2), ta);
    const D om4 = dadd(om, h, k3o), k4o = pend_acc(dadd(th, h, om3), ta);
    const real two = real(2), h6 = h / real(6);
    const D st = dadd(dadd(k1t, two, om2), real(1), dadd(om4, two, om3));
    const D so = dadd(dadd(k1o, two, k2o), real(1), dadd(k4o, two, k3o));
    const D tn = dadd(th, h6, st), on = dadd(om, h6, so);
    if (xn) {
        xn[2 * i] = tn.v;
        xn[2 * i + 1] = on.v;
    }
    if (F) {
        real *f = F + 6 * i;
        f[0] = tn.d0; f[1] = tn.d1; f[2] = tn.d2;
        f[3] = on.d0; f[4] = on.d1; f[5] = on.d2;
    }
}

template <typename real>
int dyn_pendulum1l_impl(long K, const void *x, const void *u, double h, const void *hpt, void *xn, void *F, void *stream) {
    if (K < 0 || !x || !u || (!xn && !F)) return ALQP_E_BADARG;
    if (K == 0) return 0;
    hipLaunchKernelGGL(k_dyn_pendulum1l<real>, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K,
                       (const real *)x, (const real *)u, (real)h, (const real *)hpt, (real *)xn, (real *)F);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

// ---- dynamics provider: cartpole1l (deqmpc/my_envs/cartpole1l/src/generated_dynamics.c,
//      generated_derivatives.c). RK4 of M(th) q'' = tau - (mb sin th th'^2, 0) + (0, 9.81 mb sin th),
//      M = [[ma, -mb cos th], [-mb cos th, md]], with the six tangents w.r.t. (q, qdot, tau).
//      cartpole1l: (ma, mb, md) = (11, 1, 2); cartpole1l_v2 (my_envs/cartpole1l_v2, same model, lighter cart
//      and pole): (0.7, 0.1, 0.05) - oracle/dyn_oracle.c, pinned by tests/golden/dyn_cartpole1l{,_v2}.npz.
template <typename real>
struct CartPar {
    real ma, mb, md;
};
template <typename real>
struct Dual6 {
    real v, d[6];
};
template <typename real>
__device__ __forceinline__ Dual6<real> d6c(real v) {
    Dual6<real> r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = 0;
    return r;
}
template <typename real>
__device__ __forceinline__ Dual6<real> d6axpy(Dual6<real> a, real s, Dual6<real> b) {  // a + s b
    Dual6<real> r;
    r.v = fma_(s, b.v, a.v);
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = fma_(s, b.d[i], a.d[i]);
    return r;
}
template <typename real>
__device__ __forceinline__ Dual6<real> d6mul(Dual6<real> a, Dual6<real> b) {
    Dual6<real> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = fma_(a.d[i], b.v, a.v * b.d[i]);
    return r;
}
template <typename real>
__device__ __forceinline__ void cart_acc(CartPar<real> p, Dual6<real> th, Dual6<real> thd, Dual6<real> t0, Dual6<real> t1,
                                         Dual6<real> &xdd, Dual6<real> &thdd) {
    using D = Dual6<real>;
    const real snv = sin(th.v), csv = cos(th.v);
    D sn, cs;
    sn.v = snv;
    cs.v = csv;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        sn.d[i] = csv * th.d[i];
        cs.d[i] = -snv * th.d[i];
    }
    const D r0 = d6axpy(t0, -p.mb, d6mul(sn, d6mul(thd, thd)));       // tau0 - mb sin(th) thd^2
    const D r1 = d6axpy(t1, real(9.81) * p.mb, sn);                   // tau1 + 9.81 mb sin(th)
    const D det = d6axpy(d6c<real>(p.ma * p.md), -p.mb * p.mb, d6mul(cs, cs));
    D idet;
    idet.v = real(1) / det.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) idet.d[i] = -det.d[i] * idet.v * idet.v;
    const D zero = d6c<real>(real(0));
    xdd = d6mul(idet, d6axpy(d6axpy(zero, p.mb, d6mul(cs, r1)), p.md, r0));    // M^-1 = [[md, mb c], [mb c, ma]] / det
    thdd = d6mul(idet, d6axpy(d6axpy(zero, p.mb, d6mul(cs, r0)), p.ma, r1));
}
template <typename real>
__global__ __launch_bounds__(256) void k_dyn_cartpole1l(long K, const real *x, const real *tau, real h, const real *hpt,
                                                        real *xn, real *J, CartPar<real> par) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    if (hpt) h = hpt[i];
    using D = Dual6<real>;
    real x0, x1, x2, x3;
    gld4(x + 4 * i, x0, x1, x2, x3);
    D px = d6c(x0), th = d6c(x1), xd = d6c(x2), thd = d6c(x3);
    D t0 = d6c(tau[2 * i]), t1 = d6c(tau[2 * i + 1]);
    px.d[0] = 1; th.d[1] = 1; xd.d[2] = 1; thd.d[3] = 1; t0.d[4] = 1; t1.d[5] = 1;
    const real hh = real(0.5) * h, two = real(2), h6 = h / real(6);
    D k1xd, k1td, k2xd, k2td, k3xd, k3td, k4xd, k4td;
    cart_acc(par, th, thd, t0, t1, k1xd, k1td);
    const D k2x = d6axpy(xd, hh, k1xd), k2t = d6axpy(thd, hh, k1td);
    cart_acc(par, d6axpy(th, hh, thd), k2t, t0, t1, k2xd, k2td);
    const D k3x = d6axpy(xd, hh, k2xd), k3t = d6axpy(thd, hh, k2td);
    cart_acc(par, d6axpy(th, hh, k2t), k3t, t0, t1, k3xd, k3td);
    const D k4x = d6axpy(xd, h, k3xd), k4t = d6axpy(thd, h, k3td);
    cart_acc(par, d6axpy(th, h, k3t), k4t, t0, t1, k4xd, k4td);
    D o[4];
    o[0] = d6axpy(px, h6, d6axpy(d6axpy(xd, two, k2x), real(1), d6axpy(k4x, two, k3x)));
    o[1] = d6axpy(th, h6, d6axpy(d6axpy(thd, two, k2t), real(1), d6axpy(k4t, two, k3t)));
    o[2] = d6axpy(xd, h6, d6axpy(d6axpy(k1xd, two, k2xd), real(1), d6axpy(k4xd, two, k3xd)));
    o[3] = d6axpy(thd, h6, d6axpy(d6axpy(k1td, two, k2td), real(1), d6axpy(k4td, two, k3td)));
    // 16-byte stores: a lane's 4 + 24 outputs are contiguous
    if (xn) gst4(xn + 4 * i, o[0].v, o[1].v, o[2].v, o[3].v);
    if (J) {
        real *jp = J + 24 * i;
        real f[24];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) f[6 * r + c] = o[r].d[c];
#pragma unroll
        for (int g = 0; g < 6; ++g) gst4(jp + 4 * g, f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]);
    }
}

template <typename real>
int dyn_cartpole1l_impl(long K, const void *x, const void *tau, double h, const void *hpt, void *xn, void *J, void *stream,
                        int version = 1) {
    if (K < 0 || !x || !tau || (!xn && !J)) return ALQP_E_BADARG;
    if (K == 0) return 0;
    const CartPar<real> par = version == 2 ? CartPar<real>{real(0.7), real(0.1), real(0.05)} : CartPar<real>{real(11), real(1), real(2)};
    hipLaunchKernelGGL(k_dyn_cartpole1l<real>, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K,
                       (const real *)x, (const real *)tau, (real)h, (const real *)hpt, (real *)xn, (real *)J, par);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

// ---- dynamics provider: cartpole2l (model and dual-number step: alqp_dyn.hpp DynCartpole2l) -----
template <typename real>
__global__ __launch_bounds__(128) void k_dyn_cartpole2l(long K, const real *x, const real *tau, real h, const real *hpt,
                                                        real *xn, real *J) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    if (hpt) h = hpt[i];
    if (J) {
        Dual<real, 9> xd[6], td[3], on[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            xd[k] = dconst<real, 9>(x[6 * i + k]);
            xd[k].d[k] = 1;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            td[k] = dconst<real, 9>(tau[3 * i + k]);
            td[k].d[6 + k] = 1;
        }
        DynCartpole2l<real>::template step_full<9>(xd, td, h, on);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (xn) xn[6 * i + r] = on[r].v;
#pragma unroll
            for (int c = 0; c < 9; ++c) J[54 * i + 9 * r + c] = on[r].d[c];
        }
    } else {
        Dual<real, 0> xd[6], td[3], on[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) xd[k].v = x[6 * i + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) td[k].v = tau[3 * i + k];
        DynCartpole2l<real>::template step_full<0>(xd, td, h, on);
#pragma unroll
        for (int r = 0; r < 6; ++r) xn[6 * i + r] = on[r].v;
    }
}

template <typename real>
int dyn_cartpole2l_impl(long K, const void *x, const void *tau, double h, const void *hpt, void *xn, void *J, void *stream) {
    if (K < 0 || !x || !tau || (!xn && !J)) return ALQP_E_BADARG;
    if (K == 0) return 0;
    hipLaunchKernelGGL(k_dyn_cartpole2l<real>, dim3((unsigned)((K + 127) / 128)), dim3(128), 0, (hipStream_t)stream, K,
                       (const real *)x, (const real *)tau, (real)h, (const real *)hpt, (real *)xn, (real *)J);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

// batch-global exit test of the Newton loop, taken on the device (al_utils.py:551-564)
__global__ void k_exit_test(const double *sumsq, double *ctl, int mode, double tol) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double nw = sqrt(sumsq[0]);
    if (mode == 0) {
        ctl[0] = 0.0;
        ctl[1] = 0.0;
        ctl[2] = nw;
    } else if (ctl[0] == 0.0) {
        ctl[1] += 1.0;
        const double old = ctl[2];
        if (nw < tol || fabs(old - nw) / nw < tol) ctl[0] = 1.0;   // nw = inf (a tripped instance): NaN, no exit
        else ctl[2] = nw;
    }
}
#endif  // ALQP_BUILD_MAIN

}  // namespace alqp

// ---- C ABI -------------------------------------------------------------------------------
#if ALQP_BUILD_MAIN
extern "C" {

int alqp_abi_version(void) { return 10; }

int alqp_dyn_pendulum1l_f32(long K, const void *x, const void *u, double h, const void *h_pt, void *xnext, void *F, void *stream) {
    return alqp::dyn_pendulum1l_impl<float>(K, x, u, h, h_pt, xnext, F, stream);
}
int alqp_dyn_pendulum1l_f64(long K, const void *x, const void *u, double h, const void *h_pt, void *xnext, void *F, void *stream) {
    return alqp::dyn_pendulum1l_impl<double>(K, x, u, h, h_pt, xnext, F, stream);
}

int alqp_dyn_cartpole1l_f32(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream) {
    return alqp::dyn_cartpole1l_impl<float>(K, x, tau, h, h_pt, xnext, J, stream);
}
int alqp_dyn_cartpole1l_f64(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream) {
    return alqp::dyn_cartpole1l_impl<double>(K, x, tau, h, h_pt, xnext, J, stream);
}
int alqp_dyn_cartpole1l_v2_f32(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream) {
    return alqp::dyn_cartpole1l_impl<float>(K, x, tau, h, h_pt, xnext, J, stream, 2);
}
int alqp_dyn_cartpole1l_v2_f64(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream) {
    return alqp::dyn_cartpole1l_impl<double>(K, x, tau, h, h_pt, xnext, J, stream, 2);
}

int alqp_dyn_cartpole2l_f32(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream) {
    return alqp::dyn_cartpole2l_impl<float>(K, x, tau, h, h_pt, xnext, J, stream);
}
int alqp_dyn_cartpole2l_f64(long K, const void *x, const void *tau, double h, const void *h_pt, void *xnext, void *J, void *stream) {
    return alqp::dyn_cartpole2l_impl<double>(K, x, tau, h, h_pt, xnext, J, stream);
}

size_t alqp_workspace_bytes_nonlin(const AlqpDims *dims, int is_f64) {
    if (!alqp::dims_ok(dims)) return 0;
    return is_f64 ? alqp::nonlin_ws_bytes<double>(dims->nx, dims->nu, dims->B, dims->T)
                  : alqp::nonlin_ws_bytes<float>(dims->nx, dims->nu, dims->B, dims->T);
}
int alqp_solve_nonlin_f32(const AlqpDims *dims, const AlqpParams *prm, int dyn_id, double dyn_h, const void *Qd,
                          const void *q, const void *x0, const void *u_lo, const void *u_hi, long sb_u, long st_u, void *z,
                          void *lam, void *rho, void *phi, void *rnorm2, int *info, unsigned char *status, void *workspace,
                          size_t ws_bytes, void *stream) {
    return alqp::solve_nonlin_impl<float>(dims, prm, dyn_id, dyn_h, Qd, q, x0, u_lo, u_hi, sb_u, st_u, z, lam, rho, phi,
                                          rnorm2, info, status, workspace, ws_bytes, stream);
}
int alqp_solve_nonlin_f64(const AlqpDims *dims, const AlqpParams *prm, int dyn_id, double dyn_h, const void *Qd,
                          const void *q, const void *x0, const void *u_lo, const void *u_hi, long sb_u, long st_u, void *z,
                          void *lam, void *rho, void *phi, void *rnorm2, int *info, unsigned char *status, void *workspace,
                          size_t ws_bytes, void *stream) {
    return alqp::solve_nonlin_impl<double>(dims, prm, dyn_id, dyn_h, Qd, q, x0, u_lo, u_hi, sb_u, st_u, z, lam, rho, phi,
                                           rnorm2, info, status, workspace, ws_bytes, stream);
}

int alqp_exit_test(const double *sumsq, double *ctl, int mode, double tol, void *stream) {
    if (!sumsq || !ctl || (mode != 0 && mode != 1)) return ALQP_E_BADARG;
    hipLaunchKernelGGL(alqp::k_exit_test, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq, ctl, mode, tol);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

size_t alqp_workspace_bytes(const AlqpDims *dims, int is_f64) {
    if (!alqp::dims_ok(dims)) return 0;
    return is_f64 ? alqp::quad_ws_bytes<double>(dims->nx, dims->nu, dims->B, dims->T)
                  : alqp::quad_ws_bytes<float>(dims->nx, dims->nu, dims->B, dims->T);
}

size_t alqp_lds_bytes(const AlqpDims *dims, int is_f64) {
    if (!alqp::dims_ok(dims)) return 0;
    size_t v = is_f64 ? alqp::lds_query<double>(dims->nx, dims->nu, dims->T)
                      : alqp::lds_query<float>(dims->nx, dims->nu, dims->T);
    return v;
}

int alqp_supported(const AlqpDims *dims, int is_f64) {
    // an (nx, nu) instance exists: the quad variant (HBM workspace) runs any horizon; the team variant
    // additionally needs its LDS image to fit (alqp_supported_variant)
    return alqp_workspace_bytes(dims, is_f64) > 0;
}

int alqp_supported_variant(const AlqpDims *dims, int is_f64, int variant) {
    if (variant == 2) return alqp_workspace_bytes(dims, is_f64) > 0;
    if (variant == 1) {
        size_t v = alqp_lds_bytes(dims, is_f64);
        return v > 0 && v <= alqp::kMaxLds;
    }
    return 0;
}

int alqp_qps_per_wave(const AlqpDims *dims, int is_f64) {
    (void)is_f64;
    if (!alqp::dims_ok(dims)) return 0;
    return alqp::qpw_query(dims->nx, dims->nu);
}

#define ALQP_DEFINE(SFX, REAL)                                                                        \
    int alqp_solve_lin_##SFX(const AlqpDims *dims, const AlqpParams *prm, const void *Qd,             \
                             const void *q, const void *F, const void *c, const void *x0,             \
                             const void *u_lo, const void *u_hi, long sb_u, long st_u, void *z,       \
                             void *lam, void *rho, void *phi, void *rnorm2, int *info,                \
                             unsigned char *status, void *factor_out, const AlqpTrace *trace,         \
                             void *workspace, size_t ws_bytes, void *stream) {                        \
        return alqp::solve_lin_impl<REAL>(dims, prm, Qd, q, F, c, x0, u_lo, u_hi, sb_u, st_u, z, lam, \
                                          rho, phi, rnorm2, info, status, factor_out, trace,          \
                                          workspace, ws_bytes, stream);                               \
    }                                                                                                 \
    int alqp_newton_step_##SFX(const AlqpDims *dims, const void *z, const void *xnext, const void *F, \
                               const void *x0, const void *lam, const void *rho, const void *Qd,      \
                               const void *q, const void *u_lo, const void *u_hi, long sb_u,          \
                               long st_u, void *d_out, void *g_out, void *factor_out, int *info,      \
                               void *stream) {                                                        \
        return alqp::newton_step_impl<REAL>(dims, z, xnext, F, x0, lam, rho, Qd, q, u_lo, u_hi, sb_u, \
                                            st_u, d_out, g_out, factor_out, info, stream);            \
    }                                                                                                 \
    int alqp_merit_##SFX(const AlqpDims *dims, int K, const void *zc, const void *xnext,              \
                         const void *x0, const void *lam, const void *rho, const void *Qd,            \
                         const void *q, const void *u_lo, const void *u_hi, long sb_u, long st_u,     \
                         void *phi, void *rnorm2, void *stream) {                                     \
        return alqp::merit_impl<REAL>(dims, K, zc, xnext, x0, lam, rho, Qd, q, u_lo, u_hi, sb_u,      \
                                      st_u, phi, rnorm2, stream);                                     \
    }                                                                                                 \
    int alqp_linesearch_pick_##SFX(const AlqpDims *dims, int n_ls, const void *phi, void *phi_prev,   \
                                   const void *d, void *z, int *k_out, int *accept_out,               \
                                   void *stream) {                                                    \
        return alqp::pick_impl<REAL>(dims, n_ls, phi, phi_prev, d, z, k_out, accept_out, stream);     \
    }                                                                                                 \
    int alqp_dual_update_##SFX(const AlqpDims *dims, const void *z, const void *xnext,                \
                               const void *x0, const void *u_lo, const void *u_hi, long sb_u,         \
                               long st_u, void *lam, void *rho, double rho_scale, void *stream) {     \
        return alqp::dual_impl<REAL>(dims, z, xnext, x0, u_lo, u_hi, sb_u, st_u, lam, rho, rho_scale, \
                                     stream);                                                         \
    }                                                                                                 \
    int alqp_backward_##SFX(const AlqpDims *dims, const void *factor, const void *F, const void *rho, \
                            const void *z_final, const void *gbar, void *q_grad, void *Qd_grad,       \
                            void *stream) {                                                           \
        return alqp::backward_impl<REAL>(dims, factor, F, rho, z_final, gbar, q_grad, Qd_grad,        \
                                         stream);                                                     \
    }

ALQP_DEFINE(f32, float)
ALQP_DEFINE(f64, double)

#define ALQP_DEFINE_OBS(SFX, REAL)                                                                    \
    int alqp_newton_step_obs_##SFX(const AlqpDims *dims, const void *z, const void *xnext, const void *F, \
                                   const void *x0, const void *lam, const void *rho, const void *Qd,  \
                                   const void *q, const void *u_lo, const void *u_hi, long sb_u,      \
                                   long st_u, const AlqpObstacles *obs, void *d_out, void *g_out,     \
                                   void *factor_out, int *info, void *stream) {                       \
        return alqp::newton_step_impl<REAL>(dims, z, xnext, F, x0, lam, rho, Qd, q, u_lo, u_hi, sb_u, \
                                            st_u, d_out, g_out, factor_out, info, stream, obs);       \
    }                                                                                                 \
    int alqp_merit_obs_##SFX(const AlqpDims *dims, int K, const void *zc, const void *xnext,          \
                             const void *x0, const void *lam, const void *rho, const void *Qd,        \
                             const void *q, const void *u_lo, const void *u_hi, long sb_u, long st_u, \
                             const AlqpObstacles *obs, void *phi, void *rnorm2, void *stream) {       \
        return alqp::merit_impl<REAL>(dims, K, zc, xnext, x0, lam, rho, Qd, q, u_lo, u_hi, sb_u,      \
                                      st_u, phi, rnorm2, stream, obs);                                \
    }                                                                                                 \
    int alqp_dual_update_obs_##SFX(const AlqpDims *dims, const void *z, const void *xnext,            \
                                   const void *x0, const void *u_lo, const void *u_hi, long sb_u,     \
                                   long st_u, const AlqpObstacles *obs, void *lam, void *rho,         \
                                   double rho_scale, void *stream) {                                  \
        return alqp::dual_impl<REAL>(dims, z, xnext, x0, u_lo, u_hi, sb_u, st_u, lam, rho, rho_scale, \
                                     stream, obs);                                                    \
    }

ALQP_DEFINE_OBS(f32, float)
ALQP_DEFINE_OBS(f64, double)

#define ALQP_DEFINE_STEP_WS(SFX, REAL)                                                                \
    int alqp_newton_step_ws_##SFX(const AlqpDims *dims, const void *z, const void *xnext, const void *F, \
                                  const void *x0, const void *lam, const void *rho, const void *Qd,   \
                                  const void *q, const void *u_lo, const void *u_hi, long sb_u,       \
                                  long st_u, void *workspace, size_t ws_bytes, void *d_out,           \
                                  void *g_out, int *info, void *stream) {                             \
        if (!workspace) return ALQP_E_BADARG;                                                         \
        return alqp::newton_step_impl<REAL>(dims, z, xnext, F, x0, lam, rho, Qd, q, u_lo, u_hi, sb_u, \
                                            st_u, d_out, g_out, nullptr, info, stream, nullptr,       \
                                            workspace, ws_bytes);                                     \
    }                                                                                                 \
    int alqp_newton_step_ws_obs_##SFX(const AlqpDims *dims, const void *z, const void *xnext, const void *F, \
                                      const void *x0, const void *lam, const void *rho, const void *Qd, \
                                      const void *q, const void *u_lo, const void *u_hi, long sb_u,   \
                                      long st_u, const AlqpObstacles *obs, void *workspace, size_t ws_bytes, \
                                      void *d_out, void *g_out, int *info, void *stream) {            \
        if (!workspace) return ALQP_E_BADARG;                                                         \
        return alqp::newton_step_impl<REAL>(dims, z, xnext, F, x0, lam, rho, Qd, q, u_lo, u_hi, sb_u, \
                                            st_u, d_out, g_out, nullptr, info, stream, obs,           \
                                            workspace, ws_bytes);                                     \
    }
#define ALQP_DEFINE_MERIT_PICK(SFX, REAL)                                                             \
    int alqp_merit_pick_##SFX(const AlqpDims *dims, int n_ls, const void *d, const void *xnext_all,   \
                              const void *x0, const void *lam, const void *rho, const void *Qd,       \
                              const void *q, const void *u_lo, const void *u_hi, long sb_u, long st_u, \
                              const AlqpObstacles *obs, void *z, void *phi_prev, void *rnorm2,        \
                              void *phi_all, int *k_out, int *accept_out, void *stream) {             \
        return alqp::merit_pick_impl<REAL>(dims, n_ls, d, xnext_all, x0, lam, rho, Qd, q, u_lo, u_hi, \
                                           sb_u, st_u, obs, z, phi_prev, rnorm2, phi_all, k_out,      \
                                           accept_out, stream);                                       \
    }
ALQP_DEFINE_MERIT_PICK(f32, float)
ALQP_DEFINE_MERIT_PICK(f64, double)
ALQP_DEFINE_STEP_WS(f32, float)
ALQP_DEFINE_STEP_WS(f64, double)

int alqp_backward_ws_f32(const AlqpDims *dims, void *workspace, size_t ws_bytes, const void *F,
                         const void *rho, const void *z_final, const void *gbar, void *q_grad,
                         void *Qd_grad, void *stream) {
    return alqp::backward_ws_impl<float>(dims, workspace, ws_bytes, F, rho, z_final, gbar, q_grad, Qd_grad, stream);
}
int alqp_backward_ws_f64(const AlqpDims *dims, void *workspace, size_t ws_bytes, const void *F,
                         const void *rho, const void *z_final, const void *gbar, void *q_grad,
                         void *Qd_grad, void *stream) {
    return alqp::backward_ws_impl<double>(dims, workspace, ws_bytes, F, rho, z_final, gbar, q_grad, Qd_grad, stream);
}

}  // extern "C"
#endif  // ALQP_BUILD_MAIN

#if defined(ALQP_PHASE_TIMING) && ALQP_BUILD_MAIN
// debug build only: read (and optionally reset) the per-phase cycle counters of k_solve_lin (team kernel)
extern "C" int alqp_debug_team_cycles(unsigned long long *out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(alqp::g_team_cycles), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(alqp::g_team_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#if defined(ALQP_PHASE_TIMING) && ALQP_QUAD_F32
// debug build only: read (and optionally reset) the per-phase cycle counters of k_solve_lin_quad
extern "C" int alqp_debug_phase_cycles(unsigned long long *out10, int reset) {
    if (out10 && hipMemcpyFromSymbol(out10, HIP_SYMBOL(alqp::g_phase_cycles), 10 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[10] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(alqp::g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
