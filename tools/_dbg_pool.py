import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import montecarlopathtracing_amd as M
sc = M.Scene("scenes/", "cornell-box", width=320, height=180)
dev = M.Device(sc, 0)
rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
o = np.array([0.0, 1.0, 0.5]) + (rng.random(size=(n, 3)) - 0.5) * np.array([1.9, 1.9, 0.9])
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.hstack([o, d])
dev.set_trace_mode(M.TRACE_REFERENCE)
rf, rt, rp, _ = dev.ray_intersect(rays)
dev.set_trace_mode(M.TRACE_FAST)
valid = np.ones(n, bool)
if os.environ.get('DBG_INVALID'):
    valid = (((np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xffffffff)) >> np.uint64(7)) & np.uint64(3) != 0
for rep in range(3):
    ff, ft, fp, _ = dev.ray_intersect(rays)
    bad = np.nonzero((rf != ff) & valid)[0]
    tb = np.nonzero(valid & (rf == ff) & (rf >= 0) & (rt.view(np.int64) != ft.view(np.int64)))[0]
    print("rep", rep, "hits", int((rf >= 0).sum()), "face mismatches", bad.size, "t mismatches", tb.size)
    if bad.size:
        print(" first", bad[:40])
        print(" q%256//64 hist", np.bincount((bad % 256) // 64, minlength=4), "q//256 hist", np.bincount(bad // 256))
        print(" lane%64 hist", np.bincount(bad % 64, minlength=64))
        print(" ref face", rf[bad[:10]], "got", ff[bad[:10]])
        print(" ref t", rt[bad[:6]], "got", ft[bad[:6]])
        print(" got==0 (unwritten?)", int((ff[bad] == 0).sum()), "got==-1", int((ff[bad] == -1).sum()), "ref==-1", int((rf[bad] == -1).sum()))
