#!/bin/bash
# Debug build of the (13,4) kernels with per-phase cycle counters (-DALQP_PHASE_TIMING) into
# deq-mpc-corl_amd/csrc/build/libmi_alqp_timing.so; tools/phase_timing.py runs it on the GPU.
# Not part of the product build: the counters serialise loads and compute.
set -euo pipefail
cd "$(dirname "$0")/../deq-mpc-corl_amd/csrc"
mkdir -p build/timing
DIMS="${1:-X(13, 4)}"
cp alqp_kernels.hip build/timing/alqp_kernels_timing.hip
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -I../../include -I. -mllvm -pragma-unroll-threshold=1000000 \
  -DALQP_PHASE_TIMING "-DALQP_FOR_EACH_DIMS(X)=$DIMS" ${EXTRA_FLAGS:-} -c build/timing/alqp_kernels_timing.hip -o build/timing/alqp_kernels_timing.o
# the interior-point and rigid-body objects come from a product build (csrc/build.sh)
hipcc --offload-arch=gfx950 -shared -fPIC build/timing/alqp_kernels_timing.o build/alqp_ipm.o build/alqp_ipm_g4_f64.o \
  build/alqp_ipm_g4_f32.o build/alqp_dyn_rigid.o -o build/libmi_alqp_timing.so
