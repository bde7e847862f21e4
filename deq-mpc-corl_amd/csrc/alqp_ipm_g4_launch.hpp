// Entry points of alqp_ipm_g4.hip (one translation unit per dtype) for the dispatcher in alqp_ipm.hip.
// Return 0, ALQP_E_UNSUPPORTED when the problem does not fit the register-resident kernel (T > 20, LDS image
// above 64 KB, strides beyond 32-bit lane indices - the caller then takes the generic kernel), ALQP_E_LAUNCH.
#pragma once
#include "alqp_ipm_args.hpp"

namespace alqp_ipm_g4 {
int launch_f64(int nx, int nu, const alqp_ipm::IpmArgs<double> &a, const double *lams, const double *slacks,
               bool backward, void *stream);
int launch_f32(int nx, int nu, const alqp_ipm::IpmArgs<float> &a, const float *lams, const float *slacks,
               bool backward, void *stream);
}  // namespace alqp_ipm_g4
