#!/bin/bash
# Builds libmi_alqp.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I../../include \
      alqp_kernels.hip -o libmi_alqp.so "$@"
