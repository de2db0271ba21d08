#!/bin/bash
# Round 5: the k2 sweep under several builds (scripts/ablate_build/libcc_NAME.so from build_variant.sh), same box.
# usage: LIBS="cur k2one k2w1" bash scripts/ab_k2_libs.sh
export REPS=${REPS:-5}
for cfg in "4 400 300 shared" "8 2000 500 shared"; do
  set -- $cfg
  for lib in ${LIBS:-cur}; do
    if [ $lib = cur ]; then unset CC_LIB_PATH; else export CC_LIB_PATH=scripts/ablate_build/libcc_$lib.so; fi
    echo -n "$lib "
    C=$1 F=$2 M=$3 K=$4 PROFILE=1 python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cams'],d['frames'],d['intrinsics'],d['kernel_us_per_full_launch'],round(d['gpu_us_per_iteration'],1), d['final_cost'])"
  done
done
