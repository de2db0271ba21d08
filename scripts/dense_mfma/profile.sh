#!/bin/bash
# Kernel trace + matrix-core counters of the dense-factorisation experiment at C3 (GPU box, repo root).
# Results: gpurun_out/dense/{dense_c3.json,kernel_stats.csv,pmc.csv}; copy what should be kept into profiles/rNN/.
set -e -o pipefail
R=$PWD
OUT=$R/gpurun_out/dense
mkdir -p $OUT
bash scripts/dense_mfma/build.sh
python scripts/dense_mfma/run_dense.py 1000 500 > $OUT/dense_c3.json
cd /tmp && export TMPDIR=/tmp REPS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/dense_mfma/run_dense.py 1000 500 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_a -- python3 $R/scripts/dense_mfma/run_dense.py 1000 500 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_b -- python3 $R/scripts/dense_mfma/run_dense.py 1000 500 > /dev/null 2>&1
cd $R
grep -E "Name|anonymous" $(find $OUT/trace -name "*kernel_stats.csv" | head -1) > $OUT/kernel_stats.csv
echo "counter,kernel,dispatches,mean,min,max" > $OUT/pmc.csv
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do python scripts/pmc_summarise.py $c $OUT/pmc_a | grep anonymous >> $OUT/pmc.csv; done
for c in SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES; do python scripts/pmc_summarise.py $c $OUT/pmc_b | grep anonymous >> $OUT/pmc.csv; done
rm -rf $OUT/trace $OUT/pmc_a $OUT/pmc_b
cat $OUT/dense_c3.json $OUT/kernel_stats.csv $OUT/pmc.csv
