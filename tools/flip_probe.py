#!/usr/bin/env python3
"""How many camera samples differ between the HIP path and the CPU oracle beyond 1e-9 relative ("flips"), per test scene, and how
many of those belong to paths with a ray that starts ON the surface it leaves (refraction / total reflection: the reference gives
those no 0.01 offset, MTPC/pathTracing.cpp:102,109, so whether such a ray re-hits its own triangle is decided by rounding noise).
Run on the GPU box from the repo root:  python tools/flip_probe.py [n_samples]  -> one line per scene + a JSON line."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import montecarlopathtracing_amd as M  # noqa: E402
import oracle_lib as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
SCENES = os.path.join(ROOT, "scenes") + os.sep
from conftest import extra_scene_dir  # noqa: E402
EXTRA = extra_scene_dir()
out = {}
for name in ("cornell-box", "veach-mis", "glassroom", "interior"):
    w, h = 160, 90
    base = EXTRA if name == "glassroom" else SCENES
    if name == "interior":
        from montecarlopathtracing_amd import synthetic
        base = tempfile.mkdtemp(prefix="flip_") + os.sep
        synthetic.write_interior(base, "interior", width=w, height=h, detail=0.1)
    osc = O.OracleScene(base + name, texture_dir=base, width=w, height=h)
    sc = M.Scene(base, name, width=w, height=h)
    dev = M.Device(sc, 0)
    rng = np.random.default_rng(5)
    pix = rng.integers(0, w * h, size=n).astype(np.int32)
    k = rng.integers(0, 64, size=n).astype(np.int32)
    g = dev.sample_radiance(77, pix, k)
    o = np.zeros((n, 3))
    on_surface = np.zeros(n, dtype=bool)
    for i in range(n):
        st = O.Stats()
        o[i] = osc.sample_radiance(77, int(pix[i] // w), int(pix[i] % w), int(k[i]), stats=st)
        on_surface[i] = st.rays_on_surface > 0
    err = np.abs(g - o).max(axis=1) / np.maximum(np.abs(o).max(axis=1), 1e-12)
    flip = err > 1e-9
    rec = {"samples": n, "flips": int(flip.sum()), "paths_with_on_surface_ray": int(on_surface.sum()),
           "flips_on_those": int((flip & on_surface).sum()), "flips_elsewhere": int((flip & ~on_surface).sum()),
           "max_rel_err_of_the_rest": float(err[~flip].max())}
    out[name] = rec
    print(name, rec)
    dev.close(); sc.close(); osc.close()
print(json.dumps(out))
