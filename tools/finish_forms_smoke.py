# The two forms of the finishing pass and the megakernel on the four test scenes, bit for bit (quick check before a GPU suite run): python tools/finish_forms_smoke.py
# pool form of the finishing pass against the one-lane form and the megakernel, small frames, bit for bit
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import montecarlopathtracing_amd as M
from conftest import SCENES, extra_scene_dir
from montecarlopathtracing_amd import synthetic
import tempfile
tmp = tempfile.mkdtemp() + os.sep
synthetic.write_interior(tmp, "interior", width=160, height=90, detail=0.1)
bits = lambda a: np.ascontiguousarray(a).view(np.uint64)
ok = True
for base, name in ((SCENES, "cornell-box"), (SCENES, "veach-mis"), (extra_scene_dir(), "glassroom"), (tmp, "interior")):
    sc = M.Scene(base, name, width=160, height=90)
    os.environ["MCPT_FINISH_ENGINE"] = "lane"
    dl = M.Device(sc, 0)
    del os.environ["MCPT_FINISH_ENGINE"]
    dp = M.Device(sc, 0)
    for spp in (1, 8, 32):
        sl, sp = M.Stats(), M.Stats()
        t0 = time.time(); a = dl.generateImg(spp, seed=3, stats=sl); t1 = time.time()
        b = dp.generateImg(spp, seed=3, stats=sp); t2 = time.time()
        c = dp.generateImg(spp, seed=3, flags=M.RENDER_MEGAKERNEL)
        same = np.array_equal(bits(a), bits(b)) and np.array_equal(bits(b), bits(c))
        st = (sl.rays_shadow, sl.rays_bounce, sl.shade_calls, sl.shadow_skipped, sl.max_depth) == (sp.rays_shadow, sp.rays_bounce, sp.shade_calls, sp.shadow_skipped, sp.max_depth)
        print(name, spp, "same" if same else "DIFF %d" % int((bits(a) != bits(b)).sum()), "stats ok" if st else "STATS %s %s" % ((sl.rays_shadow, sl.rays_bounce, sl.shade_calls, sl.shadow_skipped, sl.max_depth), (sp.rays_shadow, sp.rays_bounce, sp.shade_calls, sp.shadow_skipped, sp.max_depth)),
              "lane %.1f ms pool %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
        ok = ok and same and st
    dl.close(); dp.close(); sc.close()
print("SMOKE", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
