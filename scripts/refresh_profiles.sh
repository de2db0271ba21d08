#!/bin/bash
# Regenerates the measurements kept under profiles/r01/ (run on the GPU box from the repo root;
# results land in gpurun_out/refresh/, copy what should be judged into profiles/).
set -e -o pipefail
R=$PWD
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
python bench.py > $OUT/bench_sample.json
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/prof_pmc.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/prof_pmc.py > /dev/null 2>&1
echo "pmc done"
cd $R
echo "counter,kernel,dispatches,mean_KB,min_KB,max_KB" > $OUT/pmc_summary.csv
python scripts/pmc_summarise.py FETCH_SIZE $OUT/pmc_fetch >> $OUT/pmc_summary.csv
python scripts/pmc_summarise.py WRITE_SIZE $OUT/pmc_write >> $OUT/pmc_summary.csv
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python scripts/bench_configs.py > $OUT/configs.jsonl
echo "configs done"
python scripts/bench_rig.py > $OUT/rig_bench.jsonl
C=8 F=500 M=500 python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
cd $R
cp $(find $OUT/rigtrace -name "*kernel_stats.csv" | head -1) $OUT/rig_kernel_stats.csv
python scripts/time_comm1.py 2>&1 | grep -E "graph|mailbox|rccl 1|sweep" > $OUT/exchange_1rank.txt
python scripts/bench_rigk.py > $OUT/rigk_bench.jsonl
python scripts/bench_estimate.py > $OUT/estimate_bench.json
bash scripts/pmc_compute.sh > /dev/null 2>&1 && (head -1 gpurun_out/pmc2/compute_summary.csv; grep "k_intr_" gpurun_out/pmc2/compute_summary.csv) > $OUT/pmc_compute.csv
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/rigtrace
ls -la $OUT
