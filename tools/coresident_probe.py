# Do a logic block and the pool engine workgroup share a CU?  (profiles/r04_coresidency_probe.txt)  usage on the GPU box: [MCPT_LOGIC_GRID=256] [MCPT_LIB=...] python tools/coresident_probe.py <frames> <spp>
# Do a logic block and the pool engine's workgroup share a CU?  Two devices on GPU 0, one host thread each, frames rendered concurrently.
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import bench
import montecarlopathtracing_amd as M
sd = bench.write_scene_dir("cornell-box", 1280, 720)
sc = M.Scene(sd, "cornell-box")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
devs = [M.Device(sc, 0), M.Device(sc, 0)]
for d in devs:
    d.generateImg(spp, seed=1)          # workspaces
def run(d, n):
    for i in range(n):
        d.generateImg(spp, seed=2 + i)
t0 = time.perf_counter(); run(devs[0], frames); run(devs[1], frames); t_seq = time.perf_counter() - t0
ths = [threading.Thread(target=run, args=(d, frames)) for d in devs]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
t_par = time.perf_counter() - t0
print("spp %d: %d + %d frames one after the other %.1f ms per frame; two threads %.1f ms per frame (%.2fx)" % (spp, frames, frames, t_seq / (2 * frames) * 1e3, t_par / (2 * frames) * 1e3, t_seq / t_par))
