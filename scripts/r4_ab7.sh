# A/B of CC_RIG_FAST_HUBER in k_rig_sweep_frame (scripts/ablate_build/libcc_fasthuber.so)
R=$PWD
mkdir -p gpurun_out/r4m
one() { python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cams'],d['frames'],d['pts'],d['iterations'],round(d['gpu_us_per_iteration'],2),d['final_cost'])"; }
export REPS=10
for rep in 1 2; do
for lib in cur fasthuber; do
  if [ $lib = cur ]; then unset CC_LIB_PATH; else export CC_LIB_PATH=$R/scripts/ablate_build/libcc_$lib.so; fi
  for cfg in "8 2000 500" "4 400 300"; do
    set -- $cfg
    echo -n "$lib : " >> gpurun_out/r4m/ab.txt; CC_RIG_PERSIST=0 C=$1 F=$2 M=$3 one >> gpurun_out/r4m/ab.txt
  done
done
done
export CC_LIB_PATH=$R/scripts/ablate_build/libcc_fasthuber.so
python -m pytest tests/test_gpu_rig_sweeps.py tests/test_gpu_rig.py -q -x -p no:cacheprovider 2>&1 | tail -2
unset CC_LIB_PATH
cat gpurun_out/r4m/ab.txt
OUTDIR=r4m LIBS="cur fasthuber" bash scripts/r4_prof_c5.sh
