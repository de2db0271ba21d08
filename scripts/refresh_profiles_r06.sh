#!/bin/bash
# Round 6: regenerates the measurements kept under profiles/r06/ (run on the GPU box from the repo root; results land in
# gpurun_out/refresh6/, copy what should be judged into profiles/r06/).
# Sections: bench trace pmc pmcbusy rig rigtrace surface (default: all).
set -o pipefail
R=$PWD
OUT=$R/gpurun_out/refresh6
mkdir -p $OUT
SECTIONS=${@:-bench trace pmc pmcbusy rig rigtrace surface}
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
if has bench; then
  python bench.py > $OUT/bench_sample.json 2> $OUT/bench_sample.err
  echo "bench done rc=$?"
fi
if has trace; then
  # the bench LOOP's process only: no class-surface children, no rig configurations, no CPU leg
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-configs --no-class-surface > $OUT/trace_bench_line.json 2> /dev/null
  echo "trace done rc=$?"
  cd $R
  cp "$(python scripts/pick_kernel_stats.py $OUT/trace k_intr_persist)" $OUT/kernel_stats.csv && rm -rf $OUT/trace
  head -4 $OUT/kernel_stats.csv
fi
if has pmc; then
  # HBM traffic of the headline kernel (one launch = one solve), separate passes per counter (MI355X guide)
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/scripts/prof_pmc.py > /dev/null 2>&1 || echo "pmc pass $c failed"
  done
  cd $R
  echo "counter,kernel,dispatches,mean_KB,min_KB,max_KB" > $OUT/pmc_summary.csv
  for c in FETCH_SIZE WRITE_SIZE; do python scripts/pmc_summarise.py $c $OUT/pmc_$c >> $OUT/pmc_summary.csv; rm -rf $OUT/pmc_$c; done
  cat $OUT/pmc_summary.csv
fi
if has pmcbusy; then
  # issue / busy counters of the headline kernel k_intr_persist<4> (one launch = one solve of configs[2]); one
  # --pmc pass per pair of counters, the program directly after "--"
  cd /tmp && export TMPDIR=/tmp
  echo "counter,kernel,dispatches,mean,min,max" > $OUT/pmc_intr_persist.csv
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES" "SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
    tag=busy_$(echo $set | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/scripts/prof_pmc.py > /dev/null 2>&1 || echo "pass $tag failed"
    for c in $set; do python $R/scripts/pmc_summarise.py $c $OUT/$tag | grep k_intr_persist >> $OUT/pmc_intr_persist.csv; done
    rm -rf $OUT/$tag
  done
  cd $R
  cat $OUT/pmc_intr_persist.csv
fi
if has rig; then
  export REPS=10
  python scripts/bench_rig.py > $OUT/rig_bench.jsonl 2>/dev/null
  C=8 F=2000 M=500 python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
  C=2 F=1000 M=4 python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
  K=shared python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
  C=8 F=2000 M=500 K=shared python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
  C=8 F=2000 M=500 K=per_camera python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
  unset REPS
  echo "rig bench done"
fi
if has rigtrace; then
  cd /tmp && export TMPDIR=/tmp
  for cfg in "c4 4 400 300 none" "c5 8 2000 500 none" "c4k 4 400 300 shared" "c5k 8 2000 500 shared" "c5kpc 8 2000 500 per_camera"; do
    set -- $cfg
    export C=$2 F=$3 M=$4
    if [ $5 = none ]; then unset K; else export K=$5; fi
    CC_RIG_PERSIST=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace_$1 -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
    cp "$(python $R/scripts/pick_kernel_stats.py $OUT/rigtrace_$1 k_rig_)" $OUT/rig_$1_kernel_stats.csv && rm -rf $OUT/rigtrace_$1
    echo "rig trace $1 done"
  done
  unset C F M K
  cd $R
fi
if has surface; then
  tests/cpp/test_dropin --class-surface 1000 500 20 > $OUT/class_surface.jsonl
  tests/cpp/test_dropin --class-surface 1000 500 20 >> $OUT/class_surface.jsonl
  tests/cpp/test_dropin --class-surface-rig 4 400 300 8 > $OUT/class_surface_rig.jsonl
  tests/cpp/test_dropin --class-surface-rig 8 2000 500 4 >> $OUT/class_surface_rig.jsonl
  echo "class surface done"
fi
ls -la $OUT
