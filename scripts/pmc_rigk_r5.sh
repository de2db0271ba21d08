#!/bin/bash
# Round 5: instruction / traffic counters of the sweep with intrinsics at BASELINE configs[4] size (8 x 2000 x 500, shared
# intrinsics): k_rig_sweep_k2 (default) against k_rig_sweep_adjk (CC_RIG_K_COMPACT=0); separate --pmc passes (MI355X guide);
# summary -> gpurun_out/pmc_r5/summary_{k2,tiles}.csv
R=$PWD
OUT=$R/gpurun_out/pmc_r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C=8 F=2000 M=500 K=${K:-shared} REPS=2
for form in k2 tiles; do
  if [ $form = tiles ]; then export CC_RIG_K_COMPACT=0; else unset CC_RIG_K_COMPACT; fi
  for set in "SQ_INSTS_VALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $set | tr ' ' '_')
    timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$form/$tag -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1 || echo "pass $form $tag failed"
    echo "pass $form $tag done"
  done
  echo "counter,kernel,dispatches,mean,min,max" > $OUT/summary_$form.csv
  for c in SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD FETCH_SIZE WRITE_SIZE; do
    (cd $R && python scripts/pmc_summarise.py $c $OUT/$form) >> $OUT/summary_$form.csv
  done
  rm -rf $OUT/$form
  grep "k_rig_sweep\|k_rig_elim" $OUT/summary_$form.csv
done
