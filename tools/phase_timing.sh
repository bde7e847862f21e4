#!/bin/bash
# Debug build of the (13,4) kernels with per-phase cycle counters (-DALQP_PHASE_TIMING) into
# deq-mpc-corl_amd/csrc/build/libmi_alqp_timing.so; tools/phase_timing.py runs it on the GPU.
# Not part of the product build: the counters serialise loads and compute.
set -euo pipefail
cd "$(dirname "$0")/../deq-mpc-corl_amd/csrc"
mkdir -p build/timing
DIMS="${1:-X(13, 4)}"
cp alqp_kernels.hip build/timing/alqp_kernels_timing.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -mllvm -pragma-unroll-threshold=1000000 \
  -DALQP_PHASE_TIMING "-DALQP_FOR_EACH_DIMS(X)=$DIMS" ${EXTRA_FLAGS:-} -shared build/timing/alqp_kernels_timing.hip alqp_ipm.hip -o build/libmi_alqp_timing.so
