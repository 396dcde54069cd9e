#!/usr/bin/env python3
"""Per-frame kernel timeline from a rocprofv3 --kernel-trace CSV: python tools/timeline.py <kt_kernel_trace.csv> [frame index from the end]
Prints every kernel of the chosen frame (start relative to the frame's first kernel, duration, gap to the previous kernel's end)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("mcpt::", "").replace("void ", "")) for r in rows))
# frames start with k_primary_dirs or the primary-hit trace (k_trace_persistent<PrimaryRaySource>)
starts = [i for i, e in enumerate(ev) if "PrimaryRaySource" in e[2] and "slow" not in e[2]]
f0 = starts[-which]
f1 = starts[-which + 1] if which > 1 else len(ev)
t0 = ev[f0][0]
prev_end = t0
tot = {}
for s, e, n in ev[f0:f1]:
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:60]))
    prev_end = max(prev_end, e)
    tot[n] = tot.get(n, 0) + (e - s) / 1e3
print("frame span %.1f us" % ((prev_end - t0) / 1e3))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("   %9.1f us  %s" % (v, n[:70]))
