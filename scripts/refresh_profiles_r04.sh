#!/bin/bash
# Round 4: regenerates the measurements kept under profiles/r04/ (run on the GPU box from the repo root; results land in
# gpurun_out/refresh4/, copy what should be judged into profiles/r04/). Prints a progress line per step.
set -o pipefail
R=$PWD
OUT=$R/gpurun_out/refresh4
mkdir -p $OUT
python bench.py > $OUT/bench_sample.json 2> $OUT/bench_sample.err
echo "bench done rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-configs > /dev/null 2>&1
echo "trace done rc=$?"
cd $R
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/trace
# rig path: timings and kernel traces at configs[3] / configs[4] size
export REPS=10
python scripts/bench_rig.py > $OUT/rig_bench.jsonl 2>/dev/null
C=8 F=2000 M=500 python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
C=2 F=1000 M=4 python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
K=shared python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
C=8 F=2000 M=500 K=shared python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
C=8 F=2000 M=500 K=per_camera python scripts/bench_rig.py >> $OUT/rig_bench.jsonl 2>/dev/null
CC_RIG_PERSIST=0 python scripts/bench_rig.py > $OUT/rig_bench_three_kernel.jsonl 2>/dev/null
CC_RIG_PERSIST=0 C=2 F=1000 M=4 python scripts/bench_rig.py >> $OUT/rig_bench_three_kernel.jsonl 2>/dev/null
CC_RIG_SWEEP_FRAME=0 C=8 F=2000 M=500 python scripts/bench_rig.py > $OUT/rig_bench_group_form.jsonl 2>/dev/null
echo "rig bench done"
unset REPS
cd /tmp
CC_RIG_PERSIST=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
export C=8 F=2000 M=500
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rigtrace5 -- python3 $R/scripts/bench_rig.py > /dev/null 2>&1
unset C F M
cd $R
cp $(find $OUT/rigtrace -name "*kernel_stats.csv" | head -1) $OUT/rig_c4_kernel_stats.csv
cp $(find $OUT/rigtrace5 -name "*kernel_stats.csv" | head -1) $OUT/rig_c5_kernel_stats.csv
rm -rf $OUT/rigtrace $OUT/rigtrace5
echo "rig traces done"
python scripts/time_big_rig.py 2>/dev/null | grep "^{" > $OUT/rig_big.jsonl
python scripts/time_rig_sweep_scaling.py 2>/dev/null | grep "^{" > $OUT/rig_sweep_scaling.jsonl
CC_RIG_SWEEP_FRAME=0 python scripts/time_rig_sweep_scaling.py 2>/dev/null | grep "^{" >> $OUT/rig_sweep_scaling.jsonl
tests/cpp/test_dropin --class-surface 1000 500 20 > $OUT/class_surface.jsonl
tests/cpp/test_dropin --class-surface 1000 500 20 >> $OUT/class_surface.jsonl
tests/cpp/test_dropin --class-surface 200 500 20 >> $OUT/class_surface.jsonl
tests/cpp/test_dropin --class-surface-rig 4 400 300 8 > $OUT/class_surface_rig.jsonl
tests/cpp/test_dropin --class-surface-rig 8 2000 500 4 >> $OUT/class_surface_rig.jsonl
export CC_RIG_HOST_TIMING=1
(REPS=6 python scripts/time_rig_oneshot.py; REPS=6 C=8 F=2000 M=500 python scripts/time_rig_oneshot.py) 2> $OUT/rig_oneshot_phases.txt | grep "^{" > $OUT/rig_oneshot.jsonl
unset CC_RIG_HOST_TIMING
echo "big / scaling / class surface done"
# stage breakdowns from the timing-only build (wall-clock marks inside the kernels)
bash scripts/build_variant.sh rigtime cc_rig.hip --patch timing -DCC_RIG_TIMING > /dev/null 2>&1
export CC_LIB_PATH=scripts/ablate_build/libcc_rigtime.so
(CC_RIG_PERSIST=0 python scripts/time_rig_reduce.py; CC_RIG_PERSIST=0 C=8 F=2000 M=500 python scripts/time_rig_reduce.py; CC_RIG_PERSIST=0 C=8 F=2000 M=500 K=shared python scripts/time_rig_reduce.py) 2>/dev/null | grep "^{" > $OUT/rig_stage_marks.jsonl
sed -i 's/for S in (30, 42, 63):/for S in (18, 30, 42, 51, 63):/' scripts/time_chol.py
python scripts/time_chol.py 2>/dev/null | grep "^{" > $OUT/chol_routines.jsonl
unset CC_LIB_PATH
echo "stage marks done"
ls -la $OUT
