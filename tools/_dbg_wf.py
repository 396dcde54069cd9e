import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["MCPT_FINISH_PATHS"] = "0"
import montecarlopathtracing_amd as M
W, H, spp = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (160, 90, 8)))
sc = M.Scene("scenes/", "cornell-box", width=W, height=H)
os.environ.pop("MCPT_TRACE_ENGINE", None)
dv = M.Device(sc, 0)
os.environ["MCPT_TRACE_ENGINE"] = "pool"
dp = M.Device(sc, 0)
sv, sp = M.Stats(), M.Stats()
a = dv.generateImg(spp, seed=3, stats=sv)
for rep in range(3):
    b = dp.generateImg(spp, seed=3, stats=sp)
    diff = (a.view(np.int64) != b.view(np.int64)).any(axis=2)
    print("rep", rep, "pixels differing", int(diff.sum()), "of", W * H)
    print(" vote", {k: v for k, v in sv.as_dict().items() if k in ("rays_shadow", "rays_bounce", "dom_rays", "dom_node_visits", "dom_tri_tests", "launches", "max_depth")})
    print(" pool", {k: v for k, v in sp.as_dict().items() if k in ("rays_shadow", "rays_bounce", "dom_rays", "dom_node_visits", "dom_tri_tests", "launches", "max_depth")})
    if diff.sum():
        ys, xs = np.nonzero(diff)
        print(" rows", np.bincount(ys // 8)[:20], "cols", np.bincount(xs // 16)[:20])
        i = 0
        print(" first", ys[i], xs[i], a[ys[i], xs[i]], b[ys[i], xs[i]])
