#!/bin/bash
# Rebuilds only the two objects of the register-resident interior-point kernel and relinks libmi_alqp.so
# (the other objects must exist from a full csrc/build.sh run). Extra arguments go to both hipcc compiles.
set -euo pipefail
cd "$(dirname "$0")/../deq-mpc-corl_amd/csrc"
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -I../../include -mllvm -pragma-unroll-threshold=1000000"
hipcc $FLAGS -DALQP_G4_F64 -c alqp_ipm_g4.hip -o build/alqp_ipm_g4_f64.o "$@" &
p1=$!
hipcc $FLAGS -DALQP_G4_F32 -c alqp_ipm_g4.hip -o build/alqp_ipm_g4_f32.o "$@" &
p2=$!
wait $p1; wait $p2
hipcc --offload-arch=gfx950 -shared -fPIC build/alqp_part1.o build/alqp_part2.o build/alqp_part3.o build/alqp_ipm.o \
  build/alqp_ipm_g4_f64.o build/alqp_ipm_g4_f32.o build/alqp_dyn_rigid.o -o libmi_alqp.so
