#!/bin/bash
# Timing-only ablation builds of k_intr_decide_elim (results are wrong by construction). The kernel
# returns early after: 1 = the control-block read, 2 = the statistics reduction + decision,
# 3 = the publishing barrier, 4 = the 6x6 Cholesky, 5 = the triangular solves + Y write, 6 = the Schur sums (no
# partial-row store), 7 = the arrival of the last block, 8 = the row reads and sums (no 9x9 solve).
# Usage (GPU box): bash scripts/ablate_decide.sh && for v in 1 2 3 4 5 6 7 8; do CC_LIB_PATH=scripts/ablate_build/libcc_abd$v.so python scripts/time_kernels.py; done
set -e
cd "$(dirname "$0")/../camera_calibrator_amd/csrc"
mkdir -p ../../scripts/ablate_build
for v in ${ABLATE_LEVELS:-1 2 3 4 5 6 7 8}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCC_ABLATE_D=$v -c cc_intrinsics.hip -o /tmp/cc_intr_abd$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/ablate_build/libcc_abd$v.so /tmp/cc_intr_abd$v.o cc_rig.o cc_zhang.o cc_points.o cc_common.o cc_comm.o data_generator.o rig_scenario.o geometry.o -ldl -Wl,-rpath,/opt/rocm/lib
done
