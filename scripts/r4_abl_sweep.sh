# timing-only ablations of k_rig_sweep_frame (results are wrong, the LM flow degenerates: only launches that did the whole
# sweep count -- the longest ones of a kernel trace): without the cross-lane reduction, the passes, the assembly
R=$PWD
OUT=$R/gpurun_out/r4l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C=${C:-8} F=${F:-2000} M=${M:-500} REPS=8
for lib in cur abl_NO_RS abl_NO_PASS abl_NO_ASM; do
  if [ $lib = cur ]; then unset CC_LIB_PATH; else export CC_LIB_PATH=$R/scripts/ablate_build/libcc_$lib.so; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$lib -- python3 $R/scripts/bench_rig.py > $OUT/trace_$lib.log 2>&1
  echo "trace $lib done rc=$?"
  python3 - $lib $(find $OUT/trace_$lib -name "*kernel_trace.csv" | head -1) <<'PY' >> $OUT/abl_trace.txt
import csv, sys
lib, path = sys.argv[1], sys.argv[2]
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(path)) if "k_rig_sweep_frame" in r["Kernel_Name"])
full = [x for x in d if x > 0.5 * d[-1]]
print(lib, "launches", len(d), "full", len(full), "median_full_us", full[len(full)//2] / 1e3, "min_full_us", full[0] / 1e3, "max_us", d[-1] / 1e3)
PY
  rm -rf $OUT/trace_$lib
done
cat $OUT/abl_trace.txt
