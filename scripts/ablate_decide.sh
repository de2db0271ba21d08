#!/bin/bash
# Timing-only ablation builds of k_intr_decide_elim (results are wrong by construction). The kernel
# returns early after: 1 = the control-block read, 2 = the statistics reduction + decision,
# 3 = the publishing barrier, 4 = the 6x6 Cholesky, 5 = the triangular solves + Y write, 6 = the Schur sums (no
# partial-row store), 7 = the arrival of the last block, 8 = the row reads and sums (no 9x9 solve). The early returns live in
# scripts/variants/timing.patch, not in the product sources.
# Usage (GPU box): bash scripts/ablate_decide.sh && for v in 1 2 3 4 5 6 7 8; do CC_LIB_PATH=scripts/ablate_build/libcc_abd$v.so python scripts/time_kernels.py; done
set -e
for v in ${ABLATE_LEVELS:-1 2 3 4 5 6 7 8}; do bash "$(dirname "$0")/build_variant.sh" abd$v cc_intrinsics.hip --patch timing -DCC_ABLATE_D=$v; done
