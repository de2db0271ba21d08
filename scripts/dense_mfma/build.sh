#!/bin/bash
# Builds the dense-factorisation experiment (gfx950). Not part of the product build.
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o libdense_step.so dense_step.hip
