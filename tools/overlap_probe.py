#!/usr/bin/env python3
"""Does the GPU have slack that a second, independent stream of the same work could fill?  Two devices (two workspaces) on one
GPU render the same frame, first one after the other, then at the same time from two host threads on two streams.
   python tools/overlap_probe.py"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlopathtracing_amd as M  # noqa: E402
import bench  # noqa: E402

os.environ.setdefault("MCPT_WORKSPACE_GB", "50")
d = bench.write_scene_dir("cornell-box", 1280, 720)
tdev = torch.device("cuda", 0)
sc = M.Scene(d, "cornell-box")
devs = [M.Device(sc, 0), M.Device(sc, 0)]
frames = [torch.zeros((720 * 1280, 3), dtype=torch.float64, device=tdev) for _ in devs]
streams = [torch.cuda.Stream(tdev) for _ in devs]
SPP = 128      # half a frame each: together one frame's worth of samples


def render(i, seed):
    devs[i].render_device(frames[i].data_ptr(), SPP, seed, 0, 1, 0, 0, 0, None, streams[i].cuda_stream)
    streams[i].synchronize()


for i in (0, 1):
    render(i, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(3):
    render(0, 2); render(1, 3)
t1 = time.perf_counter()
for rep in range(3):
    th = [threading.Thread(target=render, args=(i, 2 + i)) for i in (0, 1)]
    for t in th: t.start()
    for t in th: t.join()
t2 = time.perf_counter()
print("two half-frames one after the other: %.1f ms; at the same time: %.1f ms" % ((t1 - t0) / 3 * 1e3, (t2 - t1) / 3 * 1e3))
