#!/bin/bash
# Builds libmi_alqp.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
# alqp_kernels.hip is compiled as three objects in parallel (team+ABI, quad fp32, quad fp64),
# alqp_ipm.hip (interior-point path, generic kernel + ABI) as a fourth, alqp_ipm_g4.hip (its register-resident kernel)
# once per dtype.
set -euo pipefail
cd "$(dirname "$0")"
# -pragma-unroll-threshold: the panel loops of alqp_quad.hpp must be fully unrolled (register
# arrays need static indices); the default threshold silently leaves them rolled.
# -O2, not -O3: these kernels run one wavefront per SIMD and are bound by instruction count; -O3's extra transformations
# add instructions (A/B on one box: headline fp32 quad kernel 5.165 -> 5.269 M solves/s, resident interior-point fp64
# 372 k -> 382 k QP/s; nothing measured got slower)
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -I../../include -mllvm -pragma-unroll-threshold=1000000"
mkdir -p build
pids=()
for part in 1 2 3; do
  hipcc $FLAGS -DALQP_PART=$part -c alqp_kernels.hip -o build/alqp_part$part.o "$@" &
  pids+=($!)
done
hipcc $FLAGS -c alqp_ipm.hip -o build/alqp_ipm.o "$@" &
pids+=($!)
# register/LDS-resident interior-point kernel: one object per dtype
hipcc $FLAGS -DALQP_G4_F64 -c alqp_ipm_g4.hip -o build/alqp_ipm_g4_f64.o "$@" &
pids+=($!)
hipcc $FLAGS -DALQP_G4_F32 -c alqp_ipm_g4.hip -o build/alqp_ipm_g4_f32.o "$@" &
pids+=($!)
hipcc $FLAGS -c alqp_dyn_rigid.hip -o build/alqp_dyn_rigid.o "$@" &   # quadrotor / flying-cartpole dynamics providers
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC build/alqp_part1.o build/alqp_part2.o build/alqp_part3.o build/alqp_ipm.o \
  build/alqp_ipm_g4_f64.o build/alqp_ipm_g4_f32.o build/alqp_dyn_rigid.o -o libmi_alqp.so
