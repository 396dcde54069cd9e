"""ctypes binding of oracle/libmcpt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (montecarlopathtracing_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SCENES = os.path.join(ROOT, "scenes")

TRACE_REAL_ONLY, TRACE_ALIAS, TRACE_FLAT = 0, 1, 2
ORDER_STABLE, ORDER_LIBSTDCXX = 0, 1


class BvhInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("t", "Lc", "Lv", "Nc", "Nv", "Nr", "Level")]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_primary", "rays_shadow", "rays_bounce", "box_tests",
                                           "tri_tests", "shade_calls", "samples")] + [("max_depth", C.c_int), ("rays_on_surface", C.c_uint64)]

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_bounce


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libmcpt_oracle.so")
    src = os.path.join(ORACLE_DIR, "mcpt_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ORACLE_DIR, "std_sort_order.cpp"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libmcpt_oracle.so"], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        L.orc_scene_load.restype = C.c_void_p
        L.orc_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_scene_set_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_scene_set_walk_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_scene_set_leaf_order.argtypes = [C.c_void_p, C.c_int]
        for f in ("orc_num_faces", "orc_num_materials", "orc_num_lights"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_get_camera.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_get_faces.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        L.orc_get_leaf_order.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.orc_get_bvh_info.argtypes = [C.c_void_p, C.POINTER(BvhInfo)]
        L.orc_get_bvh_nodes.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_find_index.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_get_material.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.orc_get_light.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_double)]
        L.orc_trace_closest.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(Stats)]
        L.orc_sample_radiance.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(Stats)]
        L.orc_primary_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_primary_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_double), C.POINTER(Stats)]
        L.orc_render_reference_style.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(Stats)]
        L.orc_render_strided.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_double), C.POINTER(Stats)]
        L.orc_quantize.argtypes = [C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_uint8)]
        L.orc_png_encode.restype = C.c_int64
        L.orc_png_encode.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int64]
        L.orc_uniform.restype = C.c_double
        L.orc_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_morton_code.restype = C.c_uint32
        L.orc_morton_code.argtypes = [C.c_float, C.c_float, C.c_float]
        _lib = L
    return _lib


class OracleScene:
    """read_scene + stable Morton sort + BVH build (MTPC/MTPC.cpp:38-45) on the CPU oracle."""

    def __init__(self, prefix, texture_dir=None, width=None, height=None):
        L = lib()
        err = C.create_string_buffer(512)
        texture_dir = texture_dir or os.path.dirname(prefix) or "."
        self.h = L.orc_scene_load(prefix.encode(), texture_dir.encode(), err, 512)
        if not self.h:
            raise RuntimeError("oracle: " + err.value.decode())
        if width is not None:
            L.orc_scene_set_resolution(self.h, width, height)
        cam = np.zeros(10)
        wh = np.zeros(2, dtype=np.int32)
        L.orc_get_camera(self.h, _ptr(cam, C.c_double), _ptr(wh, C.c_int))
        self.camera = cam
        self.width, self.height = int(wh[0]), int(wh[1])
        self.num_faces = L.orc_num_faces(self.h)
        self.num_materials = L.orc_num_materials(self.h)
        self.num_lights = L.orc_num_lights(self.h)

    def close(self):
        if self.h:
            lib().orc_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_walk_mode(self, mode):
        lib().orc_scene_set_walk_mode(self.h, mode)

    def set_leaf_order(self, which):
        if lib().orc_scene_set_leaf_order(self.h, which) != 0:
            raise RuntimeError("oracle: leaf order rebuild failed")

    def faces(self):
        n = self.num_faces
        g = np.zeros((n, 27))
        m = np.zeros(n, dtype=np.int32)
        k = np.zeros(n, dtype=np.uint32)
        lib().orc_get_faces(self.h, _ptr(g, C.c_double), _ptr(m, C.c_int32), _ptr(k, C.c_uint32))
        return g, m, k

    def leaf_order(self):
        o = np.zeros(self.num_faces, dtype=np.int32)
        lib().orc_get_leaf_order(self.h, _ptr(o, C.c_int32))
        return o

    def bvh_info(self):
        b = BvhInfo()
        lib().orc_get_bvh_info(self.h, C.byref(b))
        return b

    def bvh_nodes(self):
        nr = self.bvh_info().Nr
        box = np.zeros((nr, 6))
        lvl = np.zeros(nr, dtype=np.int32)
        leaf = np.zeros(nr, dtype=np.int32)
        lib().orc_get_bvh_nodes(self.h, _ptr(box, C.c_double), _ptr(lvl, C.c_int32), _ptr(leaf, C.c_int32))
        return box, lvl, leaf

    def material(self, i):
        name = C.create_string_buffer(64)
        rec = np.zeros(8)
        fl = np.zeros(4, dtype=np.int32)
        lib().orc_get_material(self.h, i, name, _ptr(rec, C.c_double), _ptr(fl, C.c_int32))
        return name.value.decode(), rec, fl

    def light(self, i):
        name = C.create_string_buffer(64)
        rad = np.zeros(3)
        m = C.c_int32()
        a = C.c_double()
        lib().orc_get_light(self.h, i, name, _ptr(rad, C.c_double), C.byref(m), C.byref(a))
        return name.value.decode(), rad, m.value, a.value

    def trace_closest(self, rays, mode=TRACE_REAL_ONLY, stats=None):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        face = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        p = np.zeros((n, 3))
        pn = np.zeros((n, 3))
        lib().orc_trace_closest(self.h, _ptr(rays, C.c_double), n, mode, _ptr(face, C.c_int32), _ptr(t, C.c_double),
                                _ptr(p, C.c_double), _ptr(pn, C.c_double), C.byref(stats) if stats is not None else None)
        return face, t, p, pn

    def primary_ray(self, row, col):
        r = np.zeros(6)
        lib().orc_primary_ray(self.h, row, col, _ptr(r, C.c_double))
        return r

    def primary_rays(self, row0=0, row1=None):
        """primary rays of rows [row0,row1), all columns: [(row1-row0)*W, 6]"""
        row1 = self.height if row1 is None else row1
        r = np.zeros(((row1 - row0) * self.width, 6))
        lib().orc_primary_rays(self.h, row0, row1, _ptr(r, C.c_double))
        return r

    def sample_radiance(self, seed, row, col, k, stats=None):
        rgb = np.zeros(3)
        lib().orc_sample_radiance(self.h, seed, row, col, k, _ptr(rgb, C.c_double),
                                  C.byref(stats) if stats is not None else None)
        return rgb

    def render(self, spp, seed=0, rows=None, cols=None, faithful_cost=False, nthreads=0, stats=None, img=None):
        r0, r1 = rows if rows else (0, self.height)
        c0, c1 = cols if cols else (0, self.width)
        if img is None:
            img = np.zeros((self.height, self.width, 3))
        lib().orc_render(self.h, spp, seed, r0, r1, c0, c1, int(faithful_cost), nthreads, _ptr(img, C.c_double),
                         C.byref(stats) if stats is not None else None)
        return img


def _render_strided(self, spp, seed, row_stride, faithful_cost=True, nthreads=0, stats=None, img=None):
    if img is None:
        img = np.zeros((self.height, self.width, 3))
    lib().orc_render_strided(self.h, spp, seed, row_stride, int(faithful_cost), nthreads, 0, _ptr(img, C.c_double),
                             C.byref(stats) if stats is not None else None)
    return img


OracleScene.render_strided = _render_strided


def _render_reference_style(self, spp, seed, row_stride, col_stride, stats=None, img=None):
    """the reference's own parallel structure (fork/join of min(spp, 8) threads per pixel), faithful cost, on a pixel lattice"""
    if img is None:
        img = np.zeros((self.height, self.width, 3))
    lib().orc_render_reference_style(self.h, spp, seed, row_stride, col_stride, _ptr(img, C.c_double),
                                     C.byref(stats) if stats is not None else None)
    return img


OracleScene.render_reference_style = _render_reference_style


def quantize(img):
    img = np.ascontiguousarray(img, dtype=np.float64)
    out = np.zeros(img.shape, dtype=np.uint8)
    lib().orc_quantize(_ptr(img, C.c_double), img.size, _ptr(out, C.c_uint8))
    return out


def png_encode(rgb8):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    cap = 64 + h * (w * 3 + 6) + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_png_encode(_ptr(rgb8, C.c_uint8), w, h, _ptr(out, C.c_uint8), cap)
    assert n > 0
    return out[:n].tobytes()


def uniform(seed, pixel, sample, depth, slot):
    return lib().orc_uniform(seed, pixel, sample, depth, slot)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def morton(x, y, z):
    return lib().orc_morton_code(x, y, z)


def read_stored_png(path):
    """Decode an svpng-style PNG (stored deflate blocks, one per row); tolerates a truncated tail.
    Returns (width, height, rows ndarray [rows_present, width, 3])."""
    import struct
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, w, h, idat = 8, None, None, b""
    while pos + 8 <= len(b):
        ln, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + ln]
        if typ == b"IHDR":
            w, h = struct.unpack(">II", data[:8])
        if typ == b"IDAT":
            idat += data
        pos += 12 + ln
    p, rows = 2, []
    while p + 5 <= len(idat) and len(rows) < h:
        ln = struct.unpack("<H", idat[p + 1:p + 3])[0]
        row = idat[p + 5:p + 5 + ln]
        if len(row) < ln:
            break
        rows.append(np.frombuffer(row[1:], dtype=np.uint8).reshape(w, 3))
        p += 5 + ln
    return w, h, np.stack(rows)
