#!/usr/bin/env python3
"""How the persistent closest-hit kernel's time depends on the number of rays (latency floor of a small launch).
Rays leave random points on the scene's surfaces in random directions, like bounce rays.
   python tools/launch_floor_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlopathtracing_amd as M  # noqa: E402

W, H = 1280, 720
sc = M.Scene(os.path.join(ROOT, "scenes") + os.sep, "cornell-box", width=W, height=H)
dev = M.Device(sc, 0)
rng = np.random.default_rng(1)
i = sc.info
n = 2_000_000
eye = np.array(i.eye); look = np.array(i.look_at); up = np.array(i.up) / np.linalg.norm(i.up)
fwd = look - eye
dy = np.tan(i.fovy / 2 / 180 * 3.1415926) * np.linalg.norm(fwd); dx = dy / H * W
xd = np.cross(fwd, up); xd /= np.linalg.norm(xd)
pos = look + np.outer((rng.random(n) * 2 - 1) * dx, xd) + np.outer((rng.random(n) * 2 - 1) * dy, up)
d = pos - eye
d /= np.linalg.norm(d, axis=1, keepdims=True)
f, t, p, pn = dev.ray_intersect(np.hstack([np.broadcast_to(eye, (n, 3)), d]))
o = p[f >= 0]
dd = rng.normal(size=o.shape)
dd /= np.linalg.norm(dd, axis=1, keepdims=True)
rays = np.ascontiguousarray(np.hstack([o + 0.01 * dd, dd]))
print("bounce-like rays:", rays.shape[0])
for m in (64, 256, 1000, 10000, 50000, 100000, 200000, 400000, 1000000):
    best = 1e9
    for _ in range(3):
        st = M.Stats()
        dev.ray_intersect(rays[:m], stats=st)
        best = min(best, st.ms_trace)
    print("%8d rays: %8.1f us  %8.1f Mrays/s  nodes/ray %.1f tris/ray %.1f" % (m, best * 1e3, m / best / 1e3, st.node_visits / m, st.tri_tests / m))
