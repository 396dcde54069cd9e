import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import montecarlopathtracing_amd as M
import pins_common as P
SC = os.path.join(os.getcwd(), 'scenes') + os.sep
sc = M.Scene(SC, 'veach-mis'); dev = M.Device(sc, 0)
out = {}
for name, spp in (('veach_spp10', 10), ('veach_spp100', 100)):
    qs = [M.imshow_rgb8(dev.generateImg(spp, seed=s)) for s in (201, 202, 203, 204)]
    out[name + '_q'] = np.array(qs)
np.savez_compressed('gpurun_out/r2a/veach_q.npz', **out)
