"""Generator of the synthetic stress scene of BASELINE.json configs[4]: n small triangles jittered around
a lattice that fills the Morton domain [-1,4]^3 (so the reference's 30-bit keys are not clamped), a floor and a back wall,
four quad lights, 64 diffuse materials with Kd ~ U[0.2,0.8]^3 (Ks = 0, Ns = 1, Ni = 1), seed 42.  Arrays go straight to
mcpt_scene_create (10 M triangles would be ~1 GB of .obj text); small instances can also be written as
.obj/.mtl/.camera for cross-checks against the CPU oracle."""
import os

import numpy as np


def generate(n_tris=10_000_000, seed=42, width=3840, height=2160, edge=None):
    rng = np.random.default_rng(seed)
    side = int(np.ceil(n_tris ** (1.0 / 3.0)))
    cell = 4.8 / side
    edge = edge if edge is not None else cell * 0.85
    idx = rng.permutation(side ** 3)[:n_tris]
    ix, iy, iz = idx // (side * side), (idx // side) % side, idx % side
    centre = np.stack([ix, iy, iz], axis=1) * cell + (-0.9 + 0.5 * cell) + rng.uniform(-0.2 * cell, 0.2 * cell, size=(n_tris, 3))
    # a random small triangle around each centre
    a = rng.normal(size=(n_tris, 3))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(a, rng.normal(size=(n_tris, 3)))
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    r = edge * 0.5
    v1 = centre + r * a
    v2 = centre + r * (-0.5 * a + 0.866 * b)
    v3 = centre + r * (-0.5 * a - 0.866 * b)
    nrm = np.cross(v2 - v1, v3 - v1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    v = np.concatenate([v1, v2, v3], axis=1)
    vn = np.concatenate([nrm, nrm, nrm], axis=1)
    mat = rng.integers(0, 64, size=n_tris).astype(np.int32)

    def quad(p0, p1, p2, p3, n):
        p0, p1, p2, p3, n = map(lambda q: np.asarray(q, dtype=np.float64), (p0, p1, p2, p3, n))
        return (np.array([np.concatenate([p0, p1, p2]), np.concatenate([p0, p2, p3])]),
                np.array([np.concatenate([n, n, n]), np.concatenate([n, n, n])]))

    extra_v, extra_n, extra_m = [], [], []
    fv, fn = quad([-1, -0.95, -1], [4, -0.95, -1], [4, -0.95, 4], [-1, -0.95, 4], [0, 1, 0])     # floor
    extra_v.append(fv); extra_n.append(fn); extra_m += [64, 64]
    bv, bn = quad([-1, -1, -0.95], [4, -1, -0.95], [4, 4, -0.95], [-1, 4, -0.95], [0, 0, 1])     # back wall
    extra_v.append(bv); extra_n.append(bn); extra_m += [65, 65]
    lights = [([0.0, 3.98, 3.2], [1.0, 3.98, 3.2], [1.0, 3.98, 3.9], [0.0, 3.98, 3.9]),
              ([2.0, 3.98, 3.2], [3.0, 3.98, 3.2], [3.0, 3.98, 3.9], [2.0, 3.98, 3.9]),
              ([-0.98, 1.0, 3.2], [-0.98, 2.0, 3.2], [-0.98, 2.0, 3.9], [-0.98, 1.0, 3.9]),
              ([3.98, 1.0, 3.2], [3.98, 2.0, 3.2], [3.98, 2.0, 3.9], [3.98, 1.0, 3.9])]
    lnorm = [[0, -1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0]]
    for li, (q, n) in enumerate(zip(lights, lnorm)):
        lv, ln = quad(*q, n)
        extra_v.append(lv); extra_n.append(ln); extra_m += [66 + li, 66 + li]
    v = np.vstack([v] + extra_v)
    vn = np.vstack([vn] + extra_n)
    mat = np.concatenate([mat, np.array(extra_m, dtype=np.int32)])
    kd = rng.uniform(0.2, 0.8, size=(66, 3))
    rec = np.zeros((70, 8))
    rec[:66, 0:3] = kd
    rec[:, 6] = 1.0
    rec[:, 7] = 1.0
    names = ["Kd%02d" % i for i in range(64)] + ["Floor", "BackWall"] + ["Light%d" % (i + 1) for i in range(4)]
    return dict(v=v, vn=vn, material=mat, material_rec=rec, material_names=names,
                light_material=np.array([66, 67, 68, 69], dtype=np.int32), light_radiance=np.full((4, 3), 40.0),
                eye=[1.5, 1.5, 11.0], look_at=[1.5, 1.5, 10.0], up=[0.0, 1.0, 0.0], fovy=32.0, width=width, height=height)


def make_scene(M, n_tris, defer_build=True, **kw):
    g = generate(n_tris, **kw)
    return M.Scene.from_arrays(g["v"], g["vn"], g["material"], g["material_rec"], g["light_material"], g["light_radiance"],
                               g["eye"], g["look_at"], g["up"], g["fovy"], g["width"], g["height"],
                               material_names=g["material_names"], defer_build=defer_build)


def write_obj(g, directory, name):
    """The same scene as .obj/.mtl/.camera (small instances only): each face gets its own v/vn/vt triple."""
    os.makedirs(directory, exist_ok=True)
    n = g["v"].shape[0]
    with open(os.path.join(directory, name + ".mtl"), "w") as f:
        for i, nm in enumerate(g["material_names"]):
            r = [float(x) for x in g["material_rec"][i]]
            f.write("newmtl %s\nKd %r %r %r\nKs %r %r %r\nNs %r\nNi %r\n" % (nm, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
    with open(os.path.join(directory, name + ".obj"), "w") as f:
        cur = -1
        for i in range(n):
            for c in range(3):
                f.write("v %r %r %r\nvn %r %r %r\nvt 0 0\n" % (*[float(x) for x in g["v"][i, c * 3:c * 3 + 3]],
                                                                *[float(x) for x in g["vn"][i, c * 3:c * 3 + 3]]))
            if g["material"][i] != cur:
                cur = g["material"][i]
                f.write("usemtl %s\n" % g["material_names"][cur])
            b = 3 * i + 1
            f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (b, b, b, b + 1, b + 1, b + 1, b + 2, b + 2, b + 2))
    with open(os.path.join(directory, name + ".camera"), "w") as f:
        f.write("eye %r %r %r\nlookat %r %r %r\nup %r %r %r\nfovy %r\nwidth %d\nheight %d\n" %
                (*[float(x) for x in g["eye"]], *[float(x) for x in g["look_at"]], *[float(x) for x in g["up"]], float(g["fovy"]),
                 g["width"], g["height"]))
        for lm, rad in zip(g["light_material"], g["light_radiance"]):
            f.write("mtlname %s %r %r %r\n" % (g["material_names"][lm], float(rad[0]), float(rad[1]), float(rad[2])))
