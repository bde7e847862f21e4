// alqp_ipm_g4.hip - launchers of the register/LDS-resident interior-point kernel (alqp_ipm_g4.hpp on the gfx950
// policy alqp_ipm_g4_gpu.hpp). Compiled once per dtype (-DALQP_G4_F64 / -DALQP_G4_F32); alqp_ipm.hip dispatches
// here when the horizon fits the register slots (T <= 20) and falls back to its size-generic kernel otherwise.
#include <hip/hip_runtime.h>

#include "alqp_dims.hpp"
#include "alqp_ipm_g4_gpu.hpp"
#include "alqp_ipm_g4_launch.hpp"

namespace alqp_ipm_g4 {

#ifdef ALQP_G4_TIMING
// debug build only (tools/g4_timing.sh): per-phase cycle totals over all wavefronts of all launches since the reset
__device__ unsigned long long g_phase_cycles[16];
template <typename real>
__device__ __forceinline__ void GpuX<real>::publish_timing(const long long *t) {
    if (threadIdx.x == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_phase_cycles[i], (unsigned long long)t[i]);
}
extern "C" int alqp_g4_debug_phase_cycles(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), sizeof(g_phase_cycles)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

constexpr int kSlots = 5;   // stage slots per lane group: T <= 4 * kSlots

// Waves per SIMD the register allocation aims at: fp64 is LDS-limited to three workgroups per CU at (20,13,4) (53 KB
// each), one wavefront per SIMD with the whole register file; fp32 images are half the size (six per CU), so the
// kernel is held to 256 registers and two wavefronts share a SIMD and hide each other's LDS and DPP latencies.
#ifndef ALQP_G4_F32_WAVES
#define ALQP_G4_F32_WAVES 2
#endif
template <typename real> constexpr int kWavesPerSimd = sizeof(real) == 4 ? ALQP_G4_F32_WAVES : 1;

// The kernels use no static LDS, so the dynamic allocation of a launch starts at LDS address 0. Handing the solver
// an address inside it as a LITERAL lets every LDS access fold into (lane offset register) + (immediate): the address
// of an `extern __shared__` symbol is only resolved after instruction selection and leaves one `v_add_u32 v, 0, v` in
// front of each of the ~170 LDS instructions that combine a lane offset with a block offset (40 per sweep step).
// The literal is kLdsBase, not 0: a pointer made from the constant 0 is a null pointer to the compiler, whatever the
// address space. Solver::lds_words() counts the two words of the shift.
template <typename real>
__device__ __forceinline__ real *lds_base() {
    return (real *)reinterpret_cast<__attribute__((address_space(3))) real *>(2 * sizeof(real));
}

#ifdef ALQP_G4_WPE
#define G4_WPE __attribute__((amdgpu_waves_per_eu(1, 1)))
#else
#define G4_WPE
#endif
template <typename real, int NX, int NU, bool FULLT>
__global__ __launch_bounds__(64, kWavesPerSimd<real>) G4_WPE void k_ipm_g4(const IpmArgs<real> a) {
    const int b = blockIdx.x;
    if (b >= a.B) return;
    Solver<real, NX, NU, kSlots, GpuX<real>, FULLT> S(a, lds_base<real>(), b);
    S.run_forward();
}

template <typename real, int NX, int NU, bool FULLT>
__global__ __launch_bounds__(64, kWavesPerSimd<real>) void k_ipm_g4_backward(const IpmArgs<real> a, const real *lams, const real *slacks) {
    const int b = blockIdx.x;
    if (b >= a.B) return;
    Solver<real, NX, NU, kSlots, GpuX<real>, FULLT> S(a, lds_base<real>(), b);
    S.run_backward(lams, slacks);
}

constexpr size_t kLdsMax = 64 * 1024;

template <typename real, int NX, int NU, bool FULLT>
static int launch_kernel(const IpmArgs<real> &a, const real *lams, const real *slacks, bool backward, size_t lds, hipStream_t stream) {
    const void *kf = backward ? reinterpret_cast<const void *>(k_ipm_g4_backward<real, NX, NU, FULLT>)
                              : reinterpret_cast<const void *>(k_ipm_g4<real, NX, NU, FULLT>);
    if (lds > 48 * 1024 && hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ALQP_E_LAUNCH;
    if (backward) hipLaunchKernelGGL((k_ipm_g4_backward<real, NX, NU, FULLT>), dim3(a.B), dim3(64), lds, stream, a, lams, slacks);
    else hipLaunchKernelGGL((k_ipm_g4<real, NX, NU, FULLT>), dim3(a.B), dim3(64), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : ALQP_E_LAUNCH;
}

template <typename real, int NX, int NU>
static int launch(IpmArgs<real> a, const real *lams, const real *slacks, bool backward, hipStream_t stream) {
    using S = Solver<real, NX, NU, kSlots, GpuX<real>>;
    // what the kernel's 32-bit lane indices can address (element offsets inside one instance's strided inputs)
    const long span = (long)a.T * (a.sC_t > a.sF_t ? a.sC_t : a.sF_t);
    if (a.T > S::TMAX || a.T < 2 || span >= (1L << 31)) return ALQP_E_UNSUPPORTED;
    const size_t lds = (size_t)S::lds_words(a.T) * sizeof(real);
    if (lds > kLdsMax) return ALQP_E_UNSUPPORTED;
    a.ws_words = Lay<real, NX, NU>(a.T, true).total;
    // the full horizon (every register slot holds a stage) has its own instantiation
    return a.T == S::TMAX ? launch_kernel<real, NX, NU, true>(a, lams, slacks, backward, lds, stream)
                          : launch_kernel<real, NX, NU, false>(a, lams, slacks, backward, lds, stream);
}

template <typename real>
static int dispatch(int nx, int nu, const IpmArgs<real> &a, const real *lams, const real *slacks, bool backward,
                    hipStream_t stream) {
#define X(NX, NU) \
    if (nx == NX && nu == NU) return launch<real, NX, NU>(a, lams, slacks, backward, stream);
    ALQP_FOR_EACH_DIMS(X)
#undef X
    return ALQP_E_UNSUPPORTED;
}

#ifdef ALQP_G4_F64
int launch_f64(int nx, int nu, const IpmArgs<double> &a, const double *lams, const double *slacks, bool backward,
               void *stream) {
    return dispatch<double>(nx, nu, a, lams, slacks, backward, (hipStream_t)stream);
}
#endif
#ifdef ALQP_G4_F32
int launch_f32(int nx, int nu, const IpmArgs<float> &a, const float *lams, const float *slacks, bool backward,
               void *stream) {
    return dispatch<float>(nx, nu, a, lams, slacks, backward, (hipStream_t)stream);
}
#endif

}  // namespace alqp_ipm_g4
