#!/bin/bash
# Timing-only ablation builds of the sweep kernel k_intr_sweep (results are wrong by construction):
# 1 = no MFMA, 2 = no LDS staging writes, 3 = empty main loop (prologue + epilogue only). The early returns live in
# scripts/variants/timing.patch, not in the product sources.  ->  scripts/ablate_build/libcc_ab{1,2,3}.so (CC_LIB_PATH)
set -e
for v in 1 2 3; do bash "$(dirname "$0")/build_variant.sh" ab$v cc_intrinsics.hip --patch timing -DCC_ABLATE=$v; done
