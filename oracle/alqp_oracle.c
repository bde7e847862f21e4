/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH (see alqp_oracle_impl.h).
 * Builds the fp64 and fp32 instances of the CPU restatement into one shared
 * library: symbols orc_*_f64 and orc_*_f32.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SFX f64
#define SQRT sqrt
#define FABS fabs
#include "alqp_oracle_impl.h"
#undef REAL
#undef SFX
#undef SQRT
#undef FABS

#define REAL float
#define SFX f32
#define SQRT sqrtf
#define FABS fabsf
#include "alqp_oracle_impl.h"
#undef REAL
#undef SFX
#undef SQRT
#undef FABS

int orc_max_threads(void);
#ifdef _OPENMP
#include <omp.h>
int orc_max_threads(void) { return omp_get_max_threads(); }
void orc_set_threads(int n) { omp_set_num_threads(n); }
#else
int orc_max_threads(void) { return 1; }
void orc_set_threads(int n) { (void)n; }
#endif
