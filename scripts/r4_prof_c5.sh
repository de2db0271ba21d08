# per-kernel times at BASELINE configs[4] size (8 x 2000 x 500, poses), current library; LIBS="r3 cur" adds the round-3 one
R=$PWD
OUT=$R/gpurun_out/${OUTDIR:-r4g}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C=${C:-8} F=${F:-2000} M=${M:-500} REPS=5
for lib in ${LIBS:-cur}; do
  if [ $lib = cur ]; then unset CC_LIB_PATH; else export CC_LIB_PATH=$R/scripts/ablate_build/libcc_$lib.so; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$lib -- python3 $R/scripts/bench_rig.py > $OUT/trace_$lib.log 2>&1
  echo "trace $lib done rc=$?"
  cp $(find $OUT/trace_$lib -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$lib.csv
  rm -rf $OUT/trace_$lib
  head -6 $OUT/kernel_stats_$lib.csv | cut -c1-150
done
