# idle time between consecutive kernels of the rig iteration (kernel trace timestamps), configs[4] size by default
R=$PWD
OUT=$R/gpurun_out/gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C=${C:-8} F=${F:-2000} M=${M:-500} REPS=5
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/scripts/bench_rig.py > $OUT/trace.log 2>&1
python3 - $(find $OUT/trace -name "*kernel_trace.csv" | head -1) <<'PY' > $OUT/gaps.txt
import csv, sys, statistics
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
short = lambda n: n.split("(")[0].replace("void cc::", "").replace("cc::", "")
gaps = {}
for a, b in zip(rows, rows[1:]):
    if "k_rig" in a[2] and "k_rig" in b[2] and a[1] - a[0] > 8000 and b[1] - b[0] > 8000:   # full launches only
        gaps.setdefault(short(a[2]) + " -> " + short(b[2]), []).append(b[0] - a[1])
for k, v in gaps.items():
    if len(v) > 20: print(k, "n", len(v), "median_gap_us", statistics.median(v) / 1e3, "p90", sorted(v)[int(0.9 * len(v))] / 1e3)
PY
rm -rf $OUT/trace
cat $OUT/gaps.txt
