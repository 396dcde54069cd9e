#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name: python tools/pmc_summary.py <dir> [substr]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if sub and sub not in k:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k].add(row["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(cnt[k]))
    for c, v in sorted(acc[k].items()):
        print("   %-28s %.6g" % (c, v))
