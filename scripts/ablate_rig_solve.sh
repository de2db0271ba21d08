#!/bin/bash
# Timing-only ablation builds of k_rig_solve (results are wrong by construction; read the kernel's average
# duration from a rocprofv3 trace of scripts/bench_rig.py run with CC_LIB_PATH pointing at a build).
# 1 = return after the loads and the gradient test, 2 = after the Cholesky, 3 = after the substitutions.
set -e
cd "$(dirname "$0")/../camera_calibrator_amd/csrc"
mkdir -p ../../scripts/ablate_build
for v in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCC_ABLATE_RS=$v -c cc_rig.hip -o /tmp/cc_rig_ab$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/ablate_build/libcc_abrs$v.so cc_intrinsics.o /tmp/cc_rig_ab$v.o cc_zhang.o cc_points.o cc_common.o cc_comm.o data_generator.o geometry.o -ldl
done
