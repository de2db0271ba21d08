# A/B on one box: round-3 library against the current one, rig configurations (scripts/bench_rig.py), REPS solves each
mkdir -p gpurun_out/r4d
for lib in r3 default r3 default; do
  if [ $lib = default ]; then unset CC_LIB_PATH; else export CC_LIB_PATH=scripts/ablate_build/libcc_$lib.so; fi
  for cfg in "8 2000 500" "4 400 300" "2 1000 4"; do
    set -- $cfg
    echo -n "$lib " >> gpurun_out/r4d/ab.txt
    C=$1 F=$2 M=$3 REPS=${REPS:-20} python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cams'],d['frames'],d['pts'],d['iterations'],round(d['gpu_us_per_iteration'],2))" >> gpurun_out/r4d/ab.txt
  done
  echo -n "$lib 3k " >> gpurun_out/r4d/ab.txt
  CC_RIG_PERSIST=0 C=2 F=1000 M=4 REPS=${REPS:-20} python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cams'],d['frames'],d['pts'],d['iterations'],round(d['gpu_us_per_iteration'],2))" >> gpurun_out/r4d/ab.txt
done
cat gpurun_out/r4d/ab.txt
