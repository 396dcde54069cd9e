#!/usr/bin/env python3
"""Soak of the pool engine's end game and of the logic kernel's ring (tools, not a test): many frames of random small sizes, so that
launches of every size from a few rays to millions start, drain and retire their slots, with logic grids of one to seven blocks (the
ring of the later passes then runs many rounds per block) or the default; every frame is rendered twice and must repeat bit for bit,
and a sample of them is compared with the voting engine and with the megakernel; every third frame runs with the finishing pass on
(its pool form).  usage: python tools/soak_pool.py [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["MCPT_FINISH_PATHS"] = "0"          # every bounce through the wavefront kernels
import montecarlopathtracing_amd as M

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(1)
t0 = time.time()
frames = 0
for name in ("cornell-box", "veach-mis"):
    for it in range(1000):
        if time.time() - t0 > budget * (0.5 if name == "cornell-box" else 1.0):
            break
        w, h = int(rng.integers(8, 400)), int(rng.integers(8, 240))
        spp = int(rng.choice([1, 2, 3, 5, 8, 16]))
        sc = M.Scene("scenes/", name, width=w, height=h)
        os.environ["MCPT_TRACE_ENGINE"] = "pool"
        os.environ["MCPT_FINISH_PATHS"] = "0" if it % 3 else str(int(rng.choice([200, 5000, 1500000])))
        g = int(rng.choice([0, 0, 1, 2, 3, 7]))
        if g: os.environ["MCPT_LOGIC_GRID"] = str(g)
        else: os.environ.pop("MCPT_LOGIC_GRID", None)
        dp = M.Device(sc, 0)
        a = dp.generateImg(spp, seed=it)
        b = dp.generateImg(spp, seed=it)
        assert np.array_equal(a.view(np.int64), b.view(np.int64)), (name, w, h, spp, "not repeatable")
        if it % 4 == 0:
            os.environ["MCPT_TRACE_ENGINE"] = "vote"
            dv = M.Device(sc, 0)
            c = dv.generateImg(spp, seed=it)
            assert np.array_equal(a.view(np.int64), c.view(np.int64)), (name, w, h, spp, "differs from the voting engine")
            dv.close()
            m = dp.generateImg(spp, seed=it, flags=M.RENDER_MEGAKERNEL)
            assert np.array_equal(a.view(np.int64), m.view(np.int64)), (name, w, h, spp, g, "differs from the megakernel")
        dp.close(); sc.close()
        frames += 2
        if frames % 50 == 0:
            print("%d frames, %.0f s" % (frames, time.time() - t0), flush=True)
print("soak ok: %d frames in %.0f s" % (frames, time.time() - t0))
