#!/usr/bin/env python3
"""Repeatability at full size (tools, not a test): the headline frame (cornell-box 1280x720 SPP 256) and veach-mis SPP 100 rendered N times
each; every frame must equal the first bit for bit.  usage: python tools/repeat_full.py [frames]"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import montecarlopathtracing_amd as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for name, spp in (("cornell-box", 256), ("veach-mis", 100)):
    sc = M.Scene("scenes/", name, width=1280, height=720)
    dev = M.Device(sc, 0)
    t0 = time.time()
    first = None
    for i in range(n):
        img = dev.generateImg(spp, seed=0)
        h = hashlib.sha256(np.ascontiguousarray(img).view(np.uint8)).hexdigest()[:16]
        if first is None: first = h
        assert h == first, (name, i, h, first)
    print("%s: %d frames, all %s, %.1f s" % (name, n, first, time.time() - t0), flush=True)
    dev.close(); sc.close()
print("repeat ok")
