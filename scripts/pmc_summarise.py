"""Per-kernel mean of one rocprofv3 --pmc counter from its counter_collection.csv (KB units of
FETCH_SIZE / WRITE_SIZE). Usage: pmc_summarise.py COUNTER dir-with-csv >> summary.csv"""
import csv
import glob
import os
import sys
from collections import defaultdict

counter, root = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(list)
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f'{counter},"{k}",{len(v)},{sum(v) / len(v):.4f},{min(v):.4f},{max(v):.4f}')
