#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes) -> profiles/<name>.json read by bench.py for roofline.traffic.
   python tools/pmc_to_traffic.py gpurun_out/pmc_r1_final k_wf_trace "<command profiled>" profiles/r01_final_hbm_traffic.json
Units and corrections as MI355X_MICROARCH.md prescribes: counter values are KiB; on gfx950 FETCH_SIZE reads half the bytes
of wide coalesced streams, so it is doubled (an upper bound for this kernel's mix of 16-B gathers and 8-B streams)."""
import csv
import glob
import json
import sys


def build_id():
    """hash of the sources the loaded libmcpt.so was compiled from (bench.py quotes a profile only for the build that made it)"""
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    try:
        import montecarlopathtracing_amd as M
        return M.build_id()
    except Exception as e:      # noqa: BLE001
        return "unknown (%s)" % e

d, kernel, command, out = sys.argv[1:5]
acc, disp = {}, {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].split("<")[0]        # template arguments off: k_wf_trace<27, 4> is k_wf_trace
        if not name.endswith(kernel):
            continue
        acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        disp.setdefault(row["Counter_Name"], set()).add(row["Dispatch_Id"])
n = len(disp["FETCH_SIZE"])
fetch, write = acc["FETCH_SIZE"] * 1024.0, acc["WRITE_SIZE"] * 1024.0
json.dump({"kernel": kernel, "build_id": build_id(), "command": command, "dispatches": n, "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write,
           "bytes_per_launch": (2.0 * fetch + write) / n, "bytes_per_launch_uncorrected": (fetch + write) / n,
           "note": "separate --pmc passes; KiB -> bytes; FETCH_SIZE doubled (gfx950 wide-stream correction, upper bound here)"},
          open(out, "w"), indent=1)
print(open(out).read())
