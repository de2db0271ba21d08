mkdir -p gpurun_out/r4j
one() { python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cams'],d['frames'],d['pts'],d['iterations'],round(d['gpu_us_per_iteration'],2))"; }
export REPS=10
python -m pytest tests/test_gpu_rig_sweeps.py -q -x -p no:cacheprovider 2>&1 | tail -2
for nw in default 4 8; do
  if [ $nw = default ]; then unset CC_RIG_FRAME_WAVES; else export CC_RIG_FRAME_WAVES=$nw; fi
  echo -n "frame waves $nw : " >> gpurun_out/r4j/ab.txt; C=8 F=2000 M=500 one >> gpurun_out/r4j/ab.txt
done
unset CC_RIG_FRAME_WAVES
echo -n "group form : " >> gpurun_out/r4j/ab.txt; CC_RIG_SWEEP_FRAME=0 C=8 F=2000 M=500 one >> gpurun_out/r4j/ab.txt
for cfg in "4 400 300" "2 1000 4" "16 500 100" "8 4000 60" "6 1500 200"; do
  set -- $cfg
  echo -n "3k frame : " >> gpurun_out/r4j/ab.txt; CC_RIG_PERSIST=0 C=$1 F=$2 M=$3 one >> gpurun_out/r4j/ab.txt
  echo -n "3k group : " >> gpurun_out/r4j/ab.txt; CC_RIG_SWEEP_FRAME=0 CC_RIG_PERSIST=0 C=$1 F=$2 M=$3 one >> gpurun_out/r4j/ab.txt
done
cat gpurun_out/r4j/ab.txt
OUTDIR=r4j bash scripts/r4_prof_c5.sh
