#!/bin/bash
# Timing-only ablation builds of the sweep kernel (results are wrong by construction):
# 1 = no MFMA, 2 = no LDS staging writes, 3 = empty main loop (prologue + epilogue only).
set -e
cd "$(dirname "$0")/../camera_calibrator_amd/csrc"
for v in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCC_ABLATE=$v -c cc_intrinsics.hip -o /tmp/cc_intr_ab$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/ablate_build/libcc_ab$v.so /tmp/cc_intr_ab$v.o cc_rig.o cc_points.o cc_common.o cc_comm.o -ldl
done
