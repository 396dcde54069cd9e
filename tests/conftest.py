import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENES = os.path.join(ROOT, "scenes") + os.sep

_EXTRA_DIR = None


def extra_scene_dir():
    """Directory holding the glassroom test scene (tests/scenes_extra: .obj/.mtl/.camera and its own checker texture) together with the
    reference's cherry-wood texture, which the scene also uses and which is kept once, under scenes/ -- assembled in a temporary
    directory on first use (textures are looked up next to the scene)."""
    global _EXTRA_DIR
    if _EXTRA_DIR is None:
        import atexit
        import shutil
        import tempfile
        d = tempfile.mkdtemp(prefix="mcpt_extra_")
        atexit.register(shutil.rmtree, d, ignore_errors=True)
        src = os.path.join(ROOT, "tests", "scenes_extra")
        for f in os.listdir(src):
            shutil.copy(os.path.join(src, f), d)
        for f in ("cherry-wood-texture.jpg", "cherry-wood-texture.jpg.ppm"):
            shutil.copy(os.path.join(SCENES, f), d)
        _EXTRA_DIR = d + os.sep
    return _EXTRA_DIR


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def mcpt():
    import montecarlopathtracing_amd as M
    M.lib()
    return M


def make_rays(oscene, n, seed):
    """Ray batch for closest-hit parity: camera rays, interior rays, and adversarial ones
    (axis-parallel directions -> 0/0 and x/0 in the slab test, tiny d.x -> unstable t_x, origins on box planes)."""
    rng = np.random.default_rng(seed)
    box, lvl, leaf = oscene.bvh_nodes()
    root = box[0]
    hi, lo = root[:3], root[3:]
    rays = []
    # camera rays
    H, W = oscene.height, oscene.width
    for _ in range(n // 4):
        rays.append(oscene.primary_ray(int(rng.integers(H)), int(rng.integers(W))))
    # interior random rays
    m = n // 2
    o = lo + (hi - lo) * rng.random((m, 3))
    d = rng.normal(size=(m, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays += list(np.hstack([o, d]))
    # adversarial
    leaf_boxes = box[lvl == lvl.max()]
    k = n - len(rays)
    for i in range(k):
        b = leaf_boxes[rng.integers(len(leaf_boxes))]
        kind = i % 5
        o = lo + (hi - lo) * rng.random(3)
        d = rng.normal(size=3)
        if kind == 0:      # axis-parallel
            ax = rng.integers(3)
            d = np.zeros(3)
            d[ax] = rng.choice([-1.0, 1.0])
        elif kind == 1:    # one zero component, origin on that slab plane of a leaf box -> 0/0
            ax = rng.integers(3)
            d[ax] = 0.0
            o[ax] = b[3 + ax]
        elif kind == 2:    # tiny d.x
            d[0] = rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-18, -6)
        elif kind == 3:    # origin exactly at a leaf box corner
            o = np.array([b[3], b[4], b[5]])
        else:              # aimed at a leaf box centre from outside
            c = 0.5 * (b[:3] + b[3:])
            o = c + (hi - lo) * rng.normal(size=3)
            d = c - o
        nrm = np.linalg.norm(d)
        rays.append(np.hstack([o, d / nrm if nrm > 0 else d]))
    return np.ascontiguousarray(np.array(rays, dtype=np.float64))
