// (nx, nu) instances compiled into the library. Anything else is ALQP_E_UNSUPPORTED:
// the product path fails loudly rather than falling back to a slow generic route.
#pragma once
#ifndef ALQP_FOR_EACH_DIMS   // debug builds (tools/phase_timing.sh) compile one size only
#define ALQP_FOR_EACH_DIMS(X) \
    X(2, 1) X(4, 1) X(4, 2) X(6, 1) X(6, 2) X(8, 2) X(10, 3) X(12, 4) X(13, 4) X(14, 4)
#endif
