# A/B of the slot-record prologue of k_rig_sweep_frame<.., true> (scripts/ablate_build/libcc_nofw.so: -DCC_RIG_NO_FWAVE)
R=$PWD
mkdir -p gpurun_out/r4k
one() { python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cams'],d['frames'],d['pts'],d['iterations'],round(d['gpu_us_per_iteration'],2))"; }
export REPS=10
python -m pytest tests/test_gpu_rig_sweeps.py -q -x -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
for lib in cur nofw; do
  if [ $lib = cur ]; then unset CC_LIB_PATH; else export CC_LIB_PATH=$R/scripts/ablate_build/libcc_$lib.so; fi
  for cfg in "8 2000 500" "8 2000 64" "4 400 300"; do
    set -- $cfg
    echo -n "$lib : " >> gpurun_out/r4k/ab.txt; CC_RIG_PERSIST=0 C=$1 F=$2 M=$3 one >> gpurun_out/r4k/ab.txt
  done
done
done
unset CC_LIB_PATH
cat gpurun_out/r4k/ab.txt
OUTDIR=r4k LIBS="cur nofw" bash scripts/r4_prof_c5.sh
