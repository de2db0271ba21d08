#!/bin/bash
# Round 5 same-box A/B: sweep with intrinsics as k_rig_sweep_k2 (compact records, default) against k_rig_sweep_adjk (tiles,
# CC_RIG_K_COMPACT=0) at BASELINE configs[3] / configs[4] size; per-kernel times from a profiled solve (scripts/time_forms.py style).
export REPS=${REPS:-5}
for cfg in "4 400 300 shared" "8 2000 500 shared" "8 2000 500 per_camera"; do
  set -- $cfg
  for compact in 1 0; do
    echo -n "compact=$compact "
    CC_RIG_K_COMPACT=$compact C=$1 F=$2 M=$3 K=$4 PROFILE=1 python scripts/bench_rig.py 2>/dev/null
  done
done
