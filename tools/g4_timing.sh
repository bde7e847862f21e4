#!/bin/bash
# Debug build of the register-resident interior-point kernel with per-phase cycle counters (-DALQP_G4_TIMING, fp64,
# (13,4) only) linked with the product objects into deq-mpc-corl_amd/csrc/build/libmi_alqp_g4timing.so;
# tools/g4_timing.py runs it on the GPU. Not part of the product build.
set -euo pipefail
cd "$(dirname "$0")/../deq-mpc-corl_amd/csrc"
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -I../../include -mllvm -pragma-unroll-threshold=1000000"
hipcc $FLAGS -DALQP_G4_F64 -DALQP_G4_TIMING '-DALQP_FOR_EACH_DIMS(X)=X(13,4)' -c alqp_ipm_g4.hip -o build/alqp_ipm_g4_f64_timing.o
hipcc --offload-arch=gfx950 -shared -fPIC build/alqp_part1.o build/alqp_part2.o build/alqp_part3.o build/alqp_ipm.o \
  build/alqp_ipm_g4_f64_timing.o build/alqp_ipm_g4_f32.o build/alqp_dyn_rigid.o -o build/libmi_alqp_g4timing.so
