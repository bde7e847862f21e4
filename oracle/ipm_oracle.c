/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH (see ipm_oracle_impl.h).
 * fp64 and fp32 instances of the interior-point restatement: orc_ipm_*_f64 / _f32.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SFX f64
#define SQRT sqrt
#define FABS fabs
#include "ipm_oracle_impl.h"
#undef REAL
#undef SFX
#undef SQRT
#undef FABS

#define REAL float
#define SFX f32
#define SQRT sqrtf
#define FABS fabsf
#include "ipm_oracle_impl.h"
#undef REAL
#undef SFX
#undef SQRT
#undef FABS
