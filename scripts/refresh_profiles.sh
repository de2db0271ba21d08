#!/bin/bash
# Regenerates the measurements kept under profiles/rNN/ (run on the GPU box from the repo root;
# results land in gpurun_out/refresh/, copy what should be judged into profiles/).
set -e -o pipefail
R=$PWD
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
python bench.py > $OUT/bench_sample.json
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-configs > /dev/null 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/prof_pmc.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/prof_pmc.py > /dev/null 2>&1
echo "pmc done"
cd $R
echo "counter,kernel,dispatches,mean_KB,min_KB,max_KB" > $OUT/pmc_summary.csv
python scripts/pmc_summarise.py FETCH_SIZE $OUT/pmc_fetch >> $OUT/pmc_summary.csv
python scripts/pmc_summarise.py WRITE_SIZE $OUT/pmc_write >> $OUT/pmc_summary.csv
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python scripts/bench_rig.py > $OUT/rig_bench.jsonl
C=8 F=2000 M=500 python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
K=shared python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
C=8 F=2000 M=500 K=shared python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
C=8 F=2000 M=500 K=per_camera python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
export C=8 F=2000 M=500
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace5 -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
unset C F M
cd $R
cp $(find $OUT/rigtrace -name "*kernel_stats.csv" | head -1) $OUT/rig_c4_kernel_stats.csv
cp $(find $OUT/rigtrace5 -name "*kernel_stats.csv" | head -1) $OUT/rig_c5_kernel_stats.csv
# rig path with shared intrinsics at configs[4] size, and the HBM traffic of the rig kernels at configs[4] size (poses only)
cd /tmp
export C=8 F=2000 M=500
K=shared rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace5k -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
C=4 F=400 M=300 K=shared rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace4k -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_rig_fetch -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_rig_write -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
unset C F M
cd $R
cp $(find $OUT/rigtrace5k -name "*kernel_stats.csv" | head -1) $OUT/rig_c5_shared_kernel_stats.csv
cp $(find $OUT/rigtrace4k -name "*kernel_stats.csv" | head -1) $OUT/rig_c4_shared_kernel_stats.csv
echo "counter,kernel,dispatches,mean_KB,min_KB,max_KB" > $OUT/pmc_rig_c5_summary.csv
python scripts/pmc_summarise.py FETCH_SIZE $OUT/pmc_rig_fetch >> $OUT/pmc_rig_c5_summary.csv
python scripts/pmc_summarise.py WRITE_SIZE $OUT/pmc_rig_write >> $OUT/pmc_rig_c5_summary.csv
rm -rf $OUT/rigtrace5k $OUT/rigtrace4k $OUT/pmc_rig_fetch $OUT/pmc_rig_write
# both formulations of the rig sweep and its workgroup sizes, same box
bash scripts/ab_rig_sweep.sh > /dev/null 2>&1 && cp gpurun_out/ab_rig_sweep.txt $OUT/rig_sweep_ab.txt
echo "rig done"
bash scripts/pmc_compute.sh > /dev/null 2>&1 && (head -1 gpurun_out/pmc2/compute_summary.csv; grep "k_intr_" gpurun_out/pmc2/compute_summary.csv) > $OUT/pmc_compute.csv
# stage breakdowns from the timing-only builds (wall-clock marks inside the kernels)
bash scripts/build_variant.sh intrtime cc_intrinsics.hip --patch timing -DCC_INTR_TIMING > /dev/null 2>&1
bash scripts/build_variant.sh rigtime cc_rig.hip --patch timing -DCC_RIG_TIMING > /dev/null 2>&1
(CC_LIB_PATH=scripts/ablate_build/libcc_intrtime.so python scripts/time_intr_decide.py; CC_LIB_PATH=scripts/ablate_build/libcc_intrtime.so F=125 python scripts/time_intr_decide.py) > $OUT/intr_stage_marks.jsonl 2>/dev/null
(CC_LIB_PATH=scripts/ablate_build/libcc_rigtime.so python scripts/time_rig_reduce.py; CC_LIB_PATH=scripts/ablate_build/libcc_rigtime.so C=8 F=2000 M=500 python scripts/time_rig_reduce.py; CC_LIB_PATH=scripts/ablate_build/libcc_rigtime.so C=2 F=1000 M=4 python scripts/time_rig_reduce.py) > $OUT/rig_stage_marks.jsonl 2>/dev/null
C=2 F=1000 M=4 python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/rigtrace $OUT/rigtrace5
ls -la $OUT
