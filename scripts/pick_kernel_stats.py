"""rocprofv3 writes one kernel_stats.csv per traced PROCESS (bench.py also starts child processes). Prints the path of the one
with the most launches of the named kernel -- the process the trace was taken for. Usage: pick_kernel_stats.py DIR KERNEL_SUBSTRING
(Round 4 copied `find | head -1` and committed the class-surface child's trace as the bench loop's.)"""
import csv
import glob
import os
import sys

root, needle = sys.argv[1], sys.argv[2]
best, best_calls = None, -1
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        calls = sum(int(r["Calls"]) for r in csv.DictReader(fh) if needle in r["Name"])
    if calls > best_calls:
        best, best_calls = f, calls
if best is None or best_calls <= 0:
    sys.exit(f"no kernel_stats.csv under {root} has a launch of {needle}")
print(best)
