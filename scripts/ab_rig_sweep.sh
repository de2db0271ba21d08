#!/bin/bash
# A/B of the poses-only rig sweep on the GPU box: workgroup sizes, matrix-pipe sweep, occupancy builds (scripts/build_variant.sh adjN)
cd "$(dirname "$0")/.."
OUT=gpurun_out/ab_rig_sweep.txt; : > $OUT
run() { echo "== $*" >> $OUT; env "$@" python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['cams'],d['frames'],d['pts'],'us/it',round(d['gpu_us_per_iteration'],2),'it',d['iterations'])" >> $OUT; }
for cfg in "C=4 F=400 M=300" "C=8 F=2000 M=500" "C=2 F=1000 M=4"; do
  for w in 1 2 4; do run $cfg CC_RIG_SWEEP_WG_WAVES=$w; done
  run $cfg CC_RIG_SWEEP_MFMA=1
  for v in adj5 adj6; do [ -f scripts/ablate_build/libcc_$v.so ] && run $cfg CC_LIB_PATH=scripts/ablate_build/libcc_$v.so; done
done
cat $OUT
