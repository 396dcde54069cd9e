#!/usr/bin/env python3
"""How much does ordering the 256 bounce rays of a pixel by direction buy the closest-hit kernel?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlopathtracing_amd as M  # noqa: E402

sc = M.Scene(os.path.join(ROOT, "scenes") + os.sep, "cornell-box", width=1280, height=720)
dev = M.Device(sc, 0)
rng = np.random.default_rng(3)
i = sc.info
eye, look, up = np.array(i.eye), np.array(i.look_at), np.array(i.up) / np.linalg.norm(i.up)
fwd = look - eye
dy = np.tan(i.fovy / 2 / 180 * 3.1415926) * np.linalg.norm(fwd)
dx = dy / 720 * 1280
xd = np.cross(fwd, up)
xd /= np.linalg.norm(xd)
npix, spp = 24000, 256
u = rng.random(npix) * 0.7 - 0.35
v = rng.random(npix) * 2 - 1
d = look + np.outer(u * dx, xd) + np.outer(v * dy, up) - eye
d /= np.linalg.norm(d, axis=1, keepdims=True)
f, t, p, pn = dev.ray_intersect(np.hstack([np.broadcast_to(eye, (npix, 3)), d]))
ok = f >= 0
p, pn = p[ok], pn[ok]
pn /= np.linalg.norm(pn, axis=1, keepdims=True)
n = p.shape[0]
# cosine-ish directions around pn for every sample
dirs = rng.normal(size=(n, spp, 3))
dirs /= np.linalg.norm(dirs, axis=2, keepdims=True)
dirs = dirs + pn[:, None, :] * 1.0
dirs /= np.linalg.norm(dirs, axis=2, keepdims=True)
org = np.broadcast_to(p[:, None, :], dirs.shape) + 0.01 * dirs


def run(order_fn, label):
    dd, oo = dirs.copy(), org.copy()
    if order_fn is not None:
        key = order_fn(dd)
        idx = np.argsort(key, axis=1, kind="stable")
        dd = np.take_along_axis(dd, idx[..., None], axis=1)
        oo = np.take_along_axis(oo, idx[..., None], axis=1)
    rays = np.hstack([oo.reshape(-1, 3), dd.reshape(-1, 3)])
    best = 1e9
    for _ in range(3):
        st = M.Stats()
        dev.ray_intersect(rays, stats=st)
        best = min(best, st.ms_trace)
    print("%-28s %7.2f ms  %7.1f Mrays/s  nodes/ray %.2f tris/ray %.2f" % (label, best, rays.shape[0] / best / 1e3, st.node_visits / rays.shape[0], st.tri_tests / rays.shape[0]))


def octant(dd):
    return (dd[..., 0] > 0) * 4 + (dd[..., 1] > 0) * 2 + (dd[..., 2] > 0) * 1


def fine(dd):
    a = np.abs(dd)
    q = np.clip((a * 3.999).astype(np.int64), 0, 3)
    return octant(dd) * 64 + q[..., 0] * 16 + q[..., 1] * 4 + q[..., 2]


print("rays", n * spp)
run(None, "unsorted (RNG order)")
run(octant, "sorted by octant (8 bins)")
run(fine, "sorted by octant+magnitudes")
