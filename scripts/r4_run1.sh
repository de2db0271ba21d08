set -e
mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_rig.py tests/test_gpu_rigk.py tests/test_gpu_rig_sweeps.py -q -x -p no:cacheprovider > gpurun_out/r4b/tests.log 2>&1 || true
tail -3 gpurun_out/r4b/tests.log
for lib in rigtime rigtime8; do
  echo "== $lib" >> gpurun_out/r4b/reduce_marks.txt
  CC_RIG_PERSIST=0 CC_LIB_PATH=scripts/ablate_build/libcc_$lib.so C=8 F=2000 M=500 python scripts/time_rig_reduce.py >> gpurun_out/r4b/reduce_marks.txt 2>&1
  CC_RIG_PERSIST=0 CC_LIB_PATH=scripts/ablate_build/libcc_$lib.so C=8 F=2000 M=500 K=shared python scripts/time_rig_reduce.py >> gpurun_out/r4b/reduce_marks.txt 2>&1
done
for lib in default panel8; do
  echo "== $lib" >> gpurun_out/r4b/bench_rig.txt
  if [ $lib = default ]; then unset CC_LIB_PATH; else export CC_LIB_PATH=scripts/ablate_build/libcc_$lib.so; fi
  C=8 F=2000 M=500 python scripts/bench_rig.py >> gpurun_out/r4b/bench_rig.txt 2>&1
  C=4 F=400 M=300 python scripts/bench_rig.py >> gpurun_out/r4b/bench_rig.txt 2>&1
  CC_RIG_PERSIST=0 C=4 F=400 M=300 python scripts/bench_rig.py >> gpurun_out/r4b/bench_rig.txt 2>&1
  C=2 F=1000 M=4 python scripts/bench_rig.py >> gpurun_out/r4b/bench_rig.txt 2>&1
  C=8 F=2000 M=500 K=shared python scripts/bench_rig.py >> gpurun_out/r4b/bench_rig.txt 2>&1
done
cat gpurun_out/r4b/bench_rig.txt
