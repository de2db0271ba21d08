#!/bin/bash
# Instruction / occupancy counters of the rig kernels at BASELINE configs[4] size (poses only), separate --pmc passes;
# summary in gpurun_out/pmc_rig/compute_summary.csv. Evidence for "the sweep is bound by instruction issue".
set -e -o pipefail
R=$PWD
OUT=$R/gpurun_out/pmc_rig
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C=8 F=2000 M=500
for set in "SQ_INSTS_VALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1 || echo "pass $tag failed"
done
cd $R
echo "counter,kernel,dispatches,mean,min,max" > $OUT/compute_summary.csv
for c in SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD; do
  python scripts/pmc_summarise.py $c $OUT >> $OUT/compute_summary.csv
done
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
grep "k_rig_sweep" $OUT/compute_summary.csv
