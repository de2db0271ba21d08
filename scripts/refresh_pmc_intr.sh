#!/bin/bash
# HBM traffic of the intrinsics kernels only (the two --pmc passes of refresh_profiles.sh) -> gpurun_out/refresh/pmc_summary.csv
set -e -o pipefail
R=$PWD
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/prof_pmc.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/prof_pmc.py > /dev/null 2>&1
cd $R
echo "counter,kernel,dispatches,mean_KB,min_KB,max_KB" > $OUT/pmc_summary.csv
python scripts/pmc_summarise.py FETCH_SIZE $OUT/pmc_fetch >> $OUT/pmc_summary.csv
python scripts/pmc_summarise.py WRITE_SIZE $OUT/pmc_write >> $OUT/pmc_summary.csv
rm -rf $OUT/pmc_fetch $OUT/pmc_write
grep k_intr_sweep $OUT/pmc_summary.csv
