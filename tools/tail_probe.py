#!/usr/bin/env python3
"""Wavefront iterations all the way down (no finishing kernel) on a small frame: run under rocprofv3 --kernel-trace to see how
the duration of a logic / trace launch pair depends on the number of live paths."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MCPT_FINISH_PATHS"] = "0"
import montecarlopathtracing_amd as M  # noqa: E402

sc = M.Scene(os.path.join(ROOT, "scenes") + os.sep, "cornell-box", width=640, height=360)
dev = M.Device(sc, 0)
for _ in range(2):
    st = M.Stats()
    dev.generateImg(16, seed=1, stats=st)
print("launches", st.launches, "ms", st.ms_total)
