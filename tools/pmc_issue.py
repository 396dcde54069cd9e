#!/usr/bin/env python3
"""Issue-side utilisation of one kernel from the rocprofv3 --pmc passes of tools/pmc_trace.sh:
   python tools/pmc_issue.py <pmc dir> <kernel substring> <out.json> "<command profiled>"
VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / (SIMDs x GRBM_GUI_ACTIVE per XCD); instructions per ray need the ray count of the run."""
import csv
import glob
import json
import sys
from collections import defaultdict


def build_id():
    """hash of the sources the loaded libmcpt.so was compiled from (bench.py quotes a profile only for the build that made it)"""
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    try:
        import montecarlopathtracing_amd as M
        return M.build_id()
    except Exception as e:      # noqa: BLE001
        return "unknown (%s)" % e

d, sub, out, cmd = sys.argv[1:5]
acc = defaultdict(float)
disp = set()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"] and "slow" not in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            disp.add(row["Dispatch_Id"])
xcds, simds = 8, 1024
cycles = acc["GRBM_GUI_ACTIVE"] / xcds
rec = {"kernel": sub, "build_id": build_id(), "command": cmd, "dispatches_counted": len(disp),
       "valu_wave_instructions": acc["SQ_INSTS_VALU"], "salu_wave_instructions": acc["SQ_INSTS_SALU"],
       "valu_busy": acc["SQ_ACTIVE_INST_VALU"] * 4 / (simds * cycles) if cycles else None,
       "tcp_busy": acc["TCP_GATE_EN2_sum"] / acc["TCP_GATE_EN1_sum"] if acc["TCP_GATE_EN1_sum"] else None,
       "l1_accesses": acc["TCP_TOTAL_CACHE_ACCESSES_sum"], "l1_to_l2_reads": acc["TCP_TCC_READ_REQ_sum"],
       "l2_hits": acc["TCC_HIT_sum"], "l2_misses": acc["TCC_MISS_sum"],
       "note": "VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); one rocprofv3 --pmc pass per counter set"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
