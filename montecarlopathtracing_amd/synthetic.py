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


# ---------------------------------------------------------------------------------------------------------------------
# Substitute for BASELINE.json configs[3] ("bedroom.obj ... textured materials"): the reference ships only renders of that
# scene, not its inputs.  A generated textured interior of the same character: > 200 k triangles, five map_Kd materials with
# procedurally generated rasters, one glass object (Ni 1.5, so the refraction branch runs), one glossy object, two area
# lights.  Written as .obj/.mtl/.camera (+ "<texture>.png.ppm" rasters) so that it goes through the reference's file surface.
def _grid(origin, du, dv, nu, nv, height=None, uv_scale=(1.0, 1.0)):
    """(nu+1) x (nv+1) vertices on origin + s*du + t*dv (s,t in [0,1]), optional displacement along the normal."""
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    origin, du, dv = (np.asarray(q, dtype=np.float64) for q in (origin, du, dv))
    n = np.cross(du, dv)
    n /= np.linalg.norm(n)
    p = origin + s[..., None] * du + t[..., None] * dv
    if height is not None:
        hgt = height(s, t)
        p = p + hgt[..., None] * n
        # smooth normals from finite differences of the displaced surface
        ps = np.gradient(p, axis=0)
        pt = np.gradient(p, axis=1)
        nn = np.cross(ps, pt)
        nn /= np.linalg.norm(nn, axis=2, keepdims=True)
    else:
        nn = np.broadcast_to(n, p.shape).copy()
    uv = np.stack([s * uv_scale[0], t * uv_scale[1]], axis=-1)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tris = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return p.reshape(-1, 3), nn.reshape(-1, 3), uv.reshape(-1, 2), tris


def _icosphere(centre, radius, level):
    t = (1.0 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1],
                  [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4],
                  [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    for _ in range(level):
        a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        n0 = len(v)
        nf = len(f)
        v = np.vstack([v, ab, bc, ca])
        iab, ibc, ica = n0 + np.arange(nf), n0 + nf + np.arange(nf), n0 + 2 * nf + np.arange(nf)
        f = np.vstack([np.stack([f[:, 0], iab, ica], 1), np.stack([f[:, 1], ibc, iab], 1), np.stack([f[:, 2], ica, ibc], 1), np.stack([iab, ibc, ica], 1)])
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    uv = np.stack([np.arctan2(v[:, 2], v[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(v[:, 1], -1, 1)) / np.pi], 1)
    return np.asarray(centre) + radius * v, v.copy(), uv, f


def _texture(kind, size, rng):
    y, x = np.mgrid[0:size, 0:size] / float(size)
    noise = rng.random((size, size))
    if kind == "wood":
        g = 0.5 + 0.5 * np.sin(40 * x + 6 * np.sin(6 * y) + 2 * noise)
        img = np.stack([0.45 + 0.35 * g, 0.25 + 0.22 * g, 0.10 + 0.10 * g], -1)
    elif kind == "stripes":
        g = (np.floor(x * 24) % 2)
        img = np.stack([0.75 - 0.2 * g, 0.72 - 0.1 * g, 0.62 + 0.1 * g], -1)
    elif kind == "fabric":
        g = ((np.floor(x * 32) + np.floor(y * 32)) % 2)
        img = np.stack([0.25 + 0.5 * g, 0.3 + 0.1 * noise, 0.55 - 0.3 * g], -1)
    elif kind == "rug":
        r = np.hypot(x - 0.5, y - 0.5)
        g = 0.5 + 0.5 * np.cos(40 * r)
        img = np.stack([0.6 * g + 0.2, 0.15 + 0.1 * noise, 0.2 + 0.3 * (1 - g)], -1)
    else:
        img = np.stack([0.6 + 0.3 * noise, 0.6 + 0.3 * noise, 0.55 + 0.3 * noise], -1)
    return (np.clip(img, 0, 1) * 255).astype(np.uint8)


def write_interior(directory, name="bedroom", width=1280, height=720, detail=1.0, seed=7):
    """Writes the textured interior; returns the number of triangles.  detail scales the tessellation (1.0 -> ~215 k triangles)."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.default_rng(seed)
    parts = []          # (material, P, N, UV, F)

    def g(n):
        return max(2, int(round(n * detail)))

    x0, x1, y0, y1, z0, z1 = -0.8, 3.8, -0.8, 2.0, -0.8, 3.8
    parts.append(("Floor", *_grid([x0, y0, z1], [x1 - x0, 0, 0], [0, 0, z0 - z1], g(70), g(70), uv_scale=(3, 3))))
    parts.append(("Ceiling", *_grid([x0, y1, z0], [x1 - x0, 0, 0], [0, 0, z1 - z0], g(40), g(40))))
    parts.append(("WallBack", *_grid([x0, y0, z0], [x1 - x0, 0, 0], [0, y1 - y0, 0], g(70), g(45), uv_scale=(2, 1))))
    parts.append(("WallLeft", *_grid([x0, y0, z1], [0, 0, z0 - z1], [0, y1 - y0, 0], g(70), g(45), uv_scale=(2, 1))))
    parts.append(("WallRight", *_grid([x1, y0, z0], [0, 0, z1 - z0], [0, y1 - y0, 0], g(70), g(45), uv_scale=(2, 1))))
    # bed: displaced blanket + two pillows
    parts.append(("Blanket", *_grid([0.2, -0.25, 2.6], [2.0, 0, 0], [0, 0, -2.6], g(200), g(200),
                                   height=lambda s, t: 0.05 * np.sin(9 * s) * np.cos(7 * t) + 0.03 * np.sin(23 * s * t), uv_scale=(2, 2))))
    for px in (0.45, 1.35):
        parts.append(("Pillow", *_grid([px, -0.18, 0.55], [0.7, 0, 0], [0, 0, -0.45], g(45), g(45),
                                      height=lambda s, t: 0.12 * np.sin(np.pi * s) * np.sin(np.pi * t))))
    # rug and curtains
    parts.append(("Rug", *_grid([2.3, y0 + 0.005, 3.2], [1.3, 0, 0], [0, 0, -1.8], g(50), g(50))))
    parts.append(("Curtain", *_grid([x0 + 0.05, y0 + 0.1, 3.4], [0, 0, -2.4], [0, 2.5, 0], g(210), g(100),
                                   height=lambda s, t: 0.06 * np.sin(38 * s) * (0.4 + 0.6 * t), uv_scale=(3, 1))))
    # wardrobe and dresser: boxes made of grids
    def box(mat, lo, hi, n):
        (bx0, by0, bz0), (bx1, by1, bz1) = lo, hi
        parts.append((mat, *_grid([bx0, by0, bz1], [bx1 - bx0, 0, 0], [0, by1 - by0, 0], n, n)))
        parts.append((mat, *_grid([bx1, by0, bz0], [bx0 - bx1, 0, 0], [0, by1 - by0, 0], n, n)))
        parts.append((mat, *_grid([bx0, by0, bz0], [0, 0, bz1 - bz0], [0, by1 - by0, 0], n, n)))
        parts.append((mat, *_grid([bx1, by0, bz1], [0, 0, bz0 - bz1], [0, by1 - by0, 0], n, n)))
        parts.append((mat, *_grid([bx0, by1, bz1], [bx1 - bx0, 0, 0], [0, 0, bz0 - bz1], n, n)))
    box("Wardrobe", (2.6, y0, -0.7), (3.7, 1.4, -0.1), g(26))
    box("Wardrobe", (2.9, y0, 0.4), (3.7, 0.1, 1.6), g(22))
    # glass ball on the dresser, glossy ball on the floor
    parts.append(("Glass", *_icosphere([3.3, 0.32, 1.0], 0.22, 5 if detail >= 0.75 else 3)))
    parts.append(("Gloss", *_icosphere([2.6, y0 + 0.25, 2.4], 0.25, 4 if detail >= 0.75 else 2)))
    # two ceiling lights, the first one the smaller: the reference freezes its area-sampling range to the FIRST light's area
    # (static u1, pathTracing.cpp:185); a later light with a smaller area would get "no triangle chosen" -> NaN cosines
    parts.append(("LampA", *_grid([0.6, y1 - 0.01, 0.8], [0.7, 0, 0], [0, 0, 0.6], 1, 1)))
    parts.append(("LampB", *_grid([2.4, y1 - 0.01, 2.2], [0.9, 0, 0], [0, 0, 0.7], 1, 1)))

    tex = {"Floor": "wood", "WallBack": "stripes", "WallLeft": "stripes", "WallRight": "stripes", "Blanket": "fabric", "Rug": "rug",
           "Curtain": "fabric", "Wardrobe": "wood"}
    kinds = sorted(set(tex.values()))
    for k in kinds:
        img = _texture(k, 256, rng)
        with open(os.path.join(directory, "tex_%s.png.ppm" % k), "wb") as f:
            f.write(b"P6\n256 256\n255\n" + img.tobytes())
    mats = {"Floor": ((1, 1, 1), (0, 0, 0), 1, 1), "Ceiling": ((0.85, 0.85, 0.85), (0, 0, 0), 1, 1), "WallBack": ((1, 1, 1), (0, 0, 0), 1, 1),
            "WallLeft": ((1, 1, 1), (0, 0, 0), 1, 1), "WallRight": ((1, 1, 1), (0, 0, 0), 1, 1), "Blanket": ((1, 1, 1), (0, 0, 0), 1, 1),
            "Pillow": ((0.85, 0.85, 0.8), (0, 0, 0), 1, 1), "Rug": ((1, 1, 1), (0, 0, 0), 1, 1), "Curtain": ((1, 1, 1), (0, 0, 0), 1, 1),
            "Wardrobe": ((1, 1, 1), (0, 0, 0), 1, 1), "Glass": ((0.02, 0.02, 0.02), (0.9, 0.9, 0.9), 500, 1.5),
            "Gloss": ((0.15, 0.12, 0.1), (0.8, 0.7, 0.5), 80, 1), "LampA": ((0, 0, 0), (0, 0, 0), 1, 1), "LampB": ((0, 0, 0), (0, 0, 0), 1, 1)}
    with open(os.path.join(directory, name + ".mtl"), "w") as f:
        for m, (kd, ks, ns, ni) in mats.items():
            f.write("newmtl %s\nKd %r %r %r\nKs %r %r %r\nNs %r\nNi %r\n" % (m, *map(float, kd), *map(float, ks), float(ns), float(ni)))
            if m in tex:
                f.write("map_Kd tex_%s.png\n" % tex[m])
    ntri = 0
    with open(os.path.join(directory, name + ".obj"), "w") as f:
        base = 1
        for m, P, N, UV, F in parts:
            lines = ["v %r %r %r" % tuple(map(float, p)) for p in P]
            lines += ["vn %r %r %r" % tuple(map(float, n)) for n in N]
            lines += ["vt %r %r" % tuple(map(float, t)) for t in UV]
            lines.append("usemtl %s" % m)
            # the reference's face syntax a/b/c: 2nd index -> vn, 3rd -> vt; all three equal here
            # (its v/vn/vt arrays are separate lists, so equal indices need equally long lists: one vn and vt per v)
            lines += ["f %d/%d/%d %d/%d/%d %d/%d/%d" % (a, a, a, b, b, b, c, c, c) for a, b, c in (F + base)]
            f.write("\n".join(lines) + "\n")
            base += len(P)
            ntri += len(F)
    with open(os.path.join(directory, name + ".camera"), "w") as f:
        f.write("eye 3.5 1.1 3.6\nlookat 2.85 0.92 2.86\nup 0 1 0\nfovy 58\nwidth %d\nheight %d\nmtlname LampA 30 28 24\nmtlname LampB 22 22 26\n"
                % (width, height))
    return ntri
