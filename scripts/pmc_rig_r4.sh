#!/bin/bash
# Round 4: instruction / traffic counters of the rig sweeps at BASELINE configs[4] size (8 x 2000 x 500, poses only), frame form
# (k_rig_sweep_frame, default) against group form (CC_RIG_SWEEP_FRAME=0, k_rig_sweep_adj); separate --pmc passes
# (MI355X guide); summary -> gpurun_out/pmc_r4/summary_{frame,group}.csv
R=$PWD
OUT=$R/gpurun_out/pmc_r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C=8 F=2000 M=500 REPS=2
for form in frame group; do
  if [ $form = group ]; then export CC_RIG_SWEEP_FRAME=0; else unset CC_RIG_SWEEP_FRAME; fi
  for set in "SQ_INSTS_VALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $set | tr ' ' '_')
    timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$form/$tag -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1 || echo "pass $form $tag failed"
    echo "pass $form $tag done"
  done
  echo "counter,kernel,dispatches,mean,min,max" > $OUT/summary_$form.csv
  for c in SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD FETCH_SIZE WRITE_SIZE; do
    (cd $R && python scripts/pmc_summarise.py $c $OUT/$form) >> $OUT/summary_$form.csv
  done
  rm -rf $OUT/$form
  cat $OUT/summary_$form.csv
done
