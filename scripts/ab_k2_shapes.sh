#!/bin/bash
# Round 5: sweep time of the intrinsics extension against observations per group at a fixed 8 M observations (fixed cost per
# workgroup against the main loop), builds in LIBS (scripts/ablate_build/libcc_NAME.so; "cur" = the in-tree library, "tiles" = CC_RIG_K_COMPACT=0)
export REPS=${REPS:-3}
for cfg in "8 4000 250" "8 2000 500" "8 1000 1000" "8 500 2000" "8 250 4000"; do
  set -- $cfg
  for lib in ${LIBS:-cur tiles}; do
    unset CC_LIB_PATH CC_RIG_K_COMPACT
    if [ $lib = tiles ]; then export CC_RIG_K_COMPACT=0; elif [ $lib != cur ]; then export CC_LIB_PATH=scripts/ablate_build/libcc_$lib.so; fi
    echo -n "$lib "
    C=$1 F=$2 M=$3 K=shared PROFILE=1 python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cams'],d['frames'],d['pts'],d['kernel_us_per_full_launch'],round(d['gpu_us_per_iteration'],1))"
  done
done
