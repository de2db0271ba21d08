mkdir -p gpurun_out/r4e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
one() { python $R/scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cams'],d['frames'],d['pts'],d['iterations'],round(d['gpu_us_per_iteration'],2))"; }
cd $R
echo -n "gate on  : " >> gpurun_out/r4e/lean.txt; C=2 F=1000 M=4 REPS=20 one >> gpurun_out/r4e/lean.txt
echo -n "gate off : " >> gpurun_out/r4e/lean.txt; CC_RIG_CTL_GATE=0 C=2 F=1000 M=4 REPS=20 one >> gpurun_out/r4e/lean.txt
echo -n "r3       : " >> gpurun_out/r4e/lean.txt; CC_LIB_PATH=scripts/ablate_build/libcc_r3.so C=2 F=1000 M=4 REPS=20 one >> gpurun_out/r4e/lean.txt
echo -n "gate on 2x800x4 : " >> gpurun_out/r4e/lean.txt; C=2 F=800 M=4 REPS=20 one >> gpurun_out/r4e/lean.txt
echo -n "r3      2x800x4 : " >> gpurun_out/r4e/lean.txt; CC_LIB_PATH=scripts/ablate_build/libcc_r3.so C=2 F=800 M=4 REPS=20 one >> gpurun_out/r4e/lean.txt
cat gpurun_out/r4e/lean.txt
for lib in r3 cur; do
  if [ $lib = cur ]; then unset CC_LIB_PATH; else export CC_LIB_PATH=$R/scripts/ablate_build/libcc_$lib.so; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4e/prof_$lib -o p -- python3 $R/scripts/bench_rig.py > $R/gpurun_out/r4e/prof_$lib.log 2>&1
  f=$(find $R/gpurun_out/r4e/prof_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; head -8 $f | cut -c1-160
  cp $f $R/gpurun_out/r4e/kernel_stats_$lib.csv
  rm -rf $R/gpurun_out/r4e/prof_$lib
done
