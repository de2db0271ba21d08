#!/bin/bash
# Round 3: regenerates the measurements kept under profiles/r03/ (run on the GPU box from the repo root; results land in
# gpurun_out/refresh3/, copy what should be judged into profiles/r03/). Prints a progress line per step.
set -e -o pipefail
R=$PWD
OUT=$R/gpurun_out/refresh3
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
# rig path: timings and kernel traces at configs[3] / configs[4] size
python scripts/bench_rig.py > $OUT/rig_bench.jsonl
C=8 F=2000 M=500 python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
C=2 F=1000 M=4 python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
K=shared python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
C=8 F=2000 M=500 K=shared python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
C=8 F=2000 M=500 K=per_camera python scripts/bench_rig.py >> $OUT/rig_bench.jsonl
echo "rig bench done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
export C=8 F=2000 M=500
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace5 -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
unset C F M
cd $R
cp $(find $OUT/rigtrace -name "*kernel_stats.csv" | head -1) $OUT/rig_c4_kernel_stats.csv
cp $(find $OUT/rigtrace5 -name "*kernel_stats.csv" | head -1) $OUT/rig_c5_kernel_stats.csv
echo "rig traces done"
# stage breakdowns from the timing-only builds (wall-clock marks inside the kernels)
bash scripts/build_variant.sh ptime cc_intrinsics_persist.hip --patch timing -DCC_PERSIST_TIMING > /dev/null 2>&1
bash scripts/build_variant.sh rigtime cc_rig.hip --patch timing -DCC_RIG_TIMING > /dev/null 2>&1
(CC_LIB_PATH=scripts/ablate_build/libcc_ptime.so python scripts/time_intr_persist.py; CC_LIB_PATH=scripts/ablate_build/libcc_ptime.so F=500 python scripts/time_intr_persist.py; CC_LIB_PATH=scripts/ablate_build/libcc_ptime.so F=250 python scripts/time_intr_persist.py; CC_LIB_PATH=scripts/ablate_build/libcc_ptime.so F=125 python scripts/time_intr_persist.py) > $OUT/intr_persist_marks.jsonl 2>/dev/null
# (the stage marks of the THREE-KERNEL form: small rigs run the lean persistent kernel by default)
export CC_RIG_PERSIST=0
(CC_LIB_PATH=scripts/ablate_build/libcc_rigtime.so python scripts/time_rig_reduce.py; CC_LIB_PATH=scripts/ablate_build/libcc_rigtime.so C=8 F=2000 M=500 python scripts/time_rig_reduce.py; CC_LIB_PATH=scripts/ablate_build/libcc_rigtime.so C=2 F=1000 M=4 python scripts/time_rig_reduce.py) > $OUT/rig_stage_marks.jsonl 2>/dev/null
python scripts/bench_rig.py > $OUT/rig_bench_three_kernel.jsonl
C=2 F=1000 M=4 python scripts/bench_rig.py >> $OUT/rig_bench_three_kernel.jsonl
unset CC_RIG_PERSIST
# round timelines of the persistent rig kernels (lean: default; glued: opt-in)
bash scripts/build_variant.sh rptime cc_rig.hip --patch timing -DCC_RIG_PTIMING > /dev/null 2>&1
(CC_LIB_PATH=scripts/ablate_build/libcc_rptime.so python scripts/time_rig_persist.py; CC_LIB_PATH=scripts/ablate_build/libcc_rptime.so C=2 F=1000 M=4 python scripts/time_rig_persist.py; CC_RIG_PERSIST=1 CC_RIG_PERSIST_LEAN=0 CC_LIB_PATH=scripts/ablate_build/libcc_rptime.so python scripts/time_rig_persist.py) > $OUT/rig_persist_marks_raw.jsonl 2>/dev/null
echo "marks done"
# the two forms of the intrinsics solver on one box, per counted LM iteration (complete solves)
(python scripts/time_forms.py) > $OUT/intr_forms.jsonl 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/rigtrace $OUT/rigtrace5
ls -la $OUT
