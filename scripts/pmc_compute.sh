#!/bin/bash
# MFMA / LDS counters of the kernels of the bench workload (separate --pmc passes; summaries in gpurun_out/pmc2/).
set -e -o pipefail
R=$PWD
OUT=$R/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/scripts/prof_pmc.py > /dev/null 2>&1 || echo "pass $tag failed"
done
cd $R
echo "counter,kernel,dispatches,mean,min,max" > $OUT/compute_summary.csv
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES; do
  python scripts/pmc_summarise.py $c $OUT >> $OUT/compute_summary.csv
done
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $OUT/compute_summary.csv
