#!/bin/bash
# Builds libmi_alqp.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
# -pragma-unroll-threshold: the panel loops of alqp_quad.hpp must be fully unrolled (register
# arrays need static indices); the default threshold silently leaves them rolled.
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I../../include \
      -mllvm -pragma-unroll-threshold=1000000 \
      alqp_kernels.hip -o libmi_alqp.so "$@"
