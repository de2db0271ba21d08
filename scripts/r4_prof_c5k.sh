# kernel trace summaries of the rig + intrinsics extension at configs[4] size (shared and per camera)
R=$PWD
OUT=$R/gpurun_out/r4k5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C=8 F=2000 M=500 REPS=5
for k in shared per_camera; do
  K=$k timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$k -- python3 $R/scripts/bench_rig.py > $OUT/trace_$k.log 2>&1
  echo "trace $k done rc=$?"
  cp $(find $OUT/trace_$k -name "*kernel_stats.csv" | head -1) $OUT/rig_c5_${k}_kernel_stats.csv
  rm -rf $OUT/trace_$k
  head -7 $OUT/rig_c5_${k}_kernel_stats.csv | cut -c1-140
done
