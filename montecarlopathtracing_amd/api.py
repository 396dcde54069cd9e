"""Python face of libmcpt's C ABI, named after the reference's own functions and types
(render_scene / scene_data / BVH / ray_intersect / generateImg / imshow)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import McptError, RenderParams, RenderSceneOptions, SceneDesc, SceneInfo, Stats, check, lib


TRACE_FAST, TRACE_REFERENCE = 0, 1
RENDER_DEFAULT, RENDER_MEGAKERNEL = 0, 2
RENDER_KEEP_STATS, RENDER_PIPELINE = 4, 8
LOAD_STANDARD_OBJ, LOAD_MTLLIB, LOAD_MORTON_BOUNDS = 1, 2, 4
OUT_PNG_DEFLATE, OUT_PFM = 1, 2
BUILD_HOST, BUILD_DEVICE, BUILD_DEVICE_FAST, BUILD_DEVICE_SAH = 0, 1, 2, 3
SCENE_DEFER_BUILD = 1
GATHER_PEER, GATHER_RCCL = 0, 1


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def build_id():
    """hash of the sources libmcpt.so was compiled from (profiles/ files are stamped with it)"""
    return lib().mcpt_build_id().decode()


def hip_runtime_path():
    """which libamdhip64 this process has mapped (the torch wheel bundles a copy with the same soname: whichever is loaded first is
    the one libmcpt.so runs on)"""
    lib()
    found = []
    try:
        for ln in open("/proc/self/maps"):
            if "libamdhip64" in ln:
                f = ln.split()[-1]
                if f not in found:
                    found.append(f)
    except OSError:
        pass
    return ", ".join(found) if found else "unknown"


def hip_runtime_info():
    """(HIP_VERSION libmcpt.so was compiled against, the loaded runtime's version, the file that runtime came from)"""
    comp, run = C.c_int32(0), C.c_int32(0)
    path = C.create_string_buffer(512)
    check(lib().mcpt_hip_runtime_info(C.byref(comp), C.byref(run), path, 512))
    return comp.value, run.value, path.value.decode(errors="replace")


def hip_runtime_check(compiled, runtime, runtime_path=""):
    """the comparison mcpt_device_create makes (pure): (0 or MCPT_ERR_HIP, message)"""
    msg = C.create_string_buffer(1024)
    rc = lib().mcpt_hip_runtime_check(compiled, runtime, runtime_path.encode(), msg, 1024)
    return rc, msg.value.decode(errors="replace")


def allow_runtime_mismatch(allow=True):
    """explicit consent to run libmcpt.so's kernels on a HIP runtime of another release (a process that imported torch first)"""
    lib().mcpt_allow_runtime_mismatch(1 if allow else 0)


def device_count():
    return lib().mcpt_device_count()


class Scene:
    """scene_data::read_scene + Morton sort + BVH::BVH (MTPC/MTPC.cpp:38-45)."""

    def __init__(self, path, filename, width=None, height=None, load_flags=0):
        """load_flags: LOAD_STANDARD_OBJ | LOAD_MTLLIB | LOAD_MORTON_BOUNDS (opt-in; 0 = the reference's reader)."""
        self._h = C.c_void_p()
        check(lib().mcpt_scene_load_ex(path.encode(), filename.encode(), load_flags, C.byref(self._h)))
        if width is not None:
            self.set_resolution(width, height)

    @classmethod
    def from_arrays(cls, v, vn, material, material_rec, light_material, light_radiance, eye, look_at, up, fovy, width, height,
                    vt=None, material_names=None, defer_build=False):
        """scene_data from arrays (mcpt_scene_create): faces in .obj order, v/vn [n,9], vt [n,6] or None."""
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1, 9)
        vn = np.ascontiguousarray(vn, dtype=np.float64).reshape(-1, 9)
        n = v.shape[0]
        material = np.ascontiguousarray(material, dtype=np.int32)
        material_rec = np.ascontiguousarray(material_rec, dtype=np.float64).reshape(-1, 8)
        light_material = np.ascontiguousarray(light_material, dtype=np.int32)
        light_radiance = np.ascontiguousarray(light_radiance, dtype=np.float64).reshape(-1, 3)
        if vt is not None:
            vt = np.ascontiguousarray(vt, dtype=np.float64).reshape(-1, 6)
        d = SceneDesc()
        d.num_faces = n
        d.v, d.vn = _p(v, C.c_double), _p(vn, C.c_double)
        d.vt = _p(vt, C.c_double) if vt is not None else None
        d.material = _p(material, C.c_int32)
        d.num_materials = material_rec.shape[0]
        d.material_rec = _p(material_rec, C.c_double)
        if material_names is not None:
            arr = (C.c_char_p * len(material_names))(*[m.encode() for m in material_names])
            d.material_names = arr
        d.num_lights = light_material.shape[0]
        d.light_material = _p(light_material, C.c_int32)
        d.light_radiance = _p(light_radiance, C.c_double)
        d.eye = (C.c_double * 3)(*eye)
        d.look_at = (C.c_double * 3)(*look_at)
        d.up = (C.c_double * 3)(*up)
        d.fovy, d.width, d.height = fovy, width, height
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        check(lib().mcpt_scene_create(C.byref(d), SCENE_DEFER_BUILD if defer_build else 0, C.byref(self._h)))
        return self

    def close(self):
        if getattr(self, "_h", None):
            lib().mcpt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass

    def set_resolution(self, width, height):
        check(lib().mcpt_scene_set_resolution(self._h, width, height))

    @property
    def info(self):
        i = SceneInfo()
        check(lib().mcpt_scene_get_info(self._h, C.byref(i)))
        return i

    @property
    def width(self):
        return self.info.width

    @property
    def height(self):
        return self.info.height

    def faces(self):
        n = self.info.num_faces
        g = np.zeros((n, 27))
        m = np.zeros(n, dtype=np.int32)
        k = np.zeros(n, dtype=np.uint32)
        check(lib().mcpt_scene_get_faces(self._h, _p(g, C.c_double), _p(m, C.c_int32), _p(k, C.c_uint32)))
        return g, m, k

    def leaf_order(self):
        o = np.zeros(self.info.num_faces, dtype=np.int32)
        check(lib().mcpt_scene_get_leaf_order(self._h, _p(o, C.c_int32)))
        return o

    def bvh_nodes(self):
        nr = self.info.bvh.Nr
        box = np.zeros((nr, 6))
        lvl = np.zeros(nr, dtype=np.int32)
        leaf = np.zeros(nr, dtype=np.int32)
        check(lib().mcpt_scene_get_bvh_nodes(self._h, _p(box, C.c_double), _p(lvl, C.c_int32), _p(leaf, C.c_int32)))
        return box, lvl, leaf

    def find_index(self, i, l):
        return lib().mcpt_scene_find_index(self._h, i, l)

    def material(self, m):
        name = C.create_string_buffer(64)
        rec = np.zeros(8)
        fl = np.zeros(4, dtype=np.int32)
        check(lib().mcpt_scene_get_material(self._h, m, name, _p(rec, C.c_double), _p(fl, C.c_int32)))
        return name.value.decode(), rec, fl

    def light(self, i):
        name = C.create_string_buffer(64)
        rad = np.zeros(3)
        m = np.zeros(1, dtype=np.int32)
        a = np.zeros(1)
        check(lib().mcpt_scene_get_light(self._h, i, name, _p(rad, C.c_double), _p(m, C.c_int32), _p(a, C.c_double)))
        return name.value.decode(), rad, int(m[0]), float(a[0])

    def trace_engine(self):
        """'pool' or 'vote': the closest-hit engine a device created for this scene now would run (mcpt_scene_trace_engine)"""
        return "pool" if lib().mcpt_scene_trace_engine(self._h) == 1 else "vote"

    def fast_bvh_stats(self):
        n = np.zeros(1, dtype=np.int32)
        d = np.zeros(1, dtype=np.int32)
        ok = np.zeros(1, dtype=np.int32)
        order = np.zeros(self.info.num_faces, dtype=np.int32)
        check(lib().mcpt_scene_fast_bvh_stats(self._h, _p(n, C.c_int32), _p(d, C.c_int32), _p(order, C.c_int32), _p(ok, C.c_int32)))
        return int(n[0]), int(d[0]), order, bool(ok[0])

    def owned_pixels(self, rank=0, world=1, tile_w=0, tile_h=0):
        rp = RenderParams(1, 0, rank, world, tile_w, tile_h, 0)
        n = lib().mcpt_owned_pixels(self._h, C.byref(rp), None)
        if n < 0:
            check(int(n))
        out = np.zeros(n, dtype=np.int32)
        lib().mcpt_owned_pixels(self._h, C.byref(rp), _p(out, C.c_int32))
        return out


class Device:
    """One MI355X holding a resident copy of a Scene."""

    def __init__(self, scene, ordinal=0, build=None):
        self.scene = scene
        self._h = C.c_void_p()
        if build is None:
            check(lib().mcpt_device_create(scene._h, ordinal, C.byref(self._h)))
        else:
            check(lib().mcpt_device_create_ex(scene._h, ordinal, build, C.byref(self._h)))
        i = scene.info
        self.width, self.height = i.width, i.height

    def close(self):
        if getattr(self, "_h", None):
            lib().mcpt_device_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass

    def bvh_nodes(self):
        """(box6 [Nr,6], leaf_face [Nr]) read back from HBM."""
        nr = self.scene.info.bvh.Nr
        box = np.zeros((nr, 6))
        leaf = np.zeros(nr, dtype=np.int32)
        check(lib().mcpt_device_get_bvh_nodes(self._h, _p(box, C.c_double), _p(leaf, C.c_int32)))
        return box, leaf

    def leaf_order(self):
        o = np.zeros(self.scene.info.num_faces, dtype=np.int32)
        check(lib().mcpt_device_get_leaf_order(self._h, _p(o, C.c_int32)))
        return o

    def set_trace_mode(self, mode):
        """TRACE_FAST (default) or TRACE_REFERENCE: which walk answers closest-hit queries (same results)."""
        check(lib().mcpt_device_set_trace_mode(self._h, mode))

    def ray_intersect(self, rays, stats=None):
        """Batch of ray_intersect (MTPC/pathTracing.cpp:382): rays [n,6] -> face (.obj index or -1), t, p, pn."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        face = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        p = np.zeros((n, 3))
        pn = np.zeros((n, 3))
        check(lib().mcpt_trace_closest(self._h, _p(rays, C.c_double), n, _p(face, C.c_int32), _p(t, C.c_double),
                                       _p(p, C.c_double), _p(pn, C.c_double), C.byref(stats) if stats is not None else None))
        return face, t, p, pn

    def generateImg(self, spp, seed=0, rank=0, world=1, tile_w=0, tile_h=0, flags=0, stats=None, img=None):
        """generateImg (MTPC/pathTracing.cpp:274): returns image::img as [H,W,3] float64."""
        if img is None:
            img = np.zeros((self.height, self.width, 3))
        rp = RenderParams(spp, seed, rank, world, tile_w, tile_h, flags)
        check(lib().mcpt_render(self._h, C.byref(rp), _p(img, C.c_double), C.byref(stats) if stats is not None else None))
        return img

    def render_device(self, d_img_ptr, spp, seed=0, rank=0, world=1, tile_w=0, tile_h=0, flags=0, stats=None, stream=None):
        """Same, into a caller-owned device buffer (e.g. a torch tensor's data_ptr()) on `stream`."""
        rp = RenderParams(spp, seed, rank, world, tile_w, tile_h, flags)
        check(lib().mcpt_render_device(self._h, C.byref(rp), C.c_void_p(d_img_ptr), C.byref(stats) if stats is not None else None,
                                       C.c_void_p(stream) if stream else None))

    def collect_stats(self, stats=None):
        """statistics of every RENDER_KEEP_STATS frame since the last call (waits for those frames)"""
        stats = stats if stats is not None else Stats()
        check(lib().mcpt_device_collect_stats(self._h, C.byref(stats)))
        return stats

    def sample_radiance(self, seed, pix, k):
        pix = np.ascontiguousarray(pix, dtype=np.int32)
        k = np.ascontiguousarray(k, dtype=np.int32)
        rgb = np.zeros((pix.shape[0], 3))
        check(lib().mcpt_sample_radiance(self._h, seed, _p(pix, C.c_int32), _p(k, C.c_int32), pix.shape[0], _p(rgb, C.c_double)))
        return rgb


class MultiDevice:
    """generateImg on several GPUs of the node behind one call (mcpt_multi_*): one host thread per GPU inside the library, tiles
    dealt like rank/world, every rank's pixels gathered into devices[0]'s HBM (peer copies over xGMI, or RCCL)."""

    def __init__(self, scene, devices=None, build=BUILD_HOST, gather=GATHER_PEER):
        self.scene = scene
        self._h = C.c_void_p()
        if devices is None:
            arr, n = None, 0
        else:
            devices = np.ascontiguousarray(devices, dtype=np.int32)
            arr, n = _p(devices, C.c_int32), devices.shape[0]
        check(lib().mcpt_multi_create(scene._h, arr, n, build, gather, C.byref(self._h)))
        i = scene.info
        self.width, self.height = i.width, i.height

    @property
    def num_devices(self):
        return lib().mcpt_multi_num_devices(self._h)

    def generateImg(self, spp, seed=0, tile_w=0, tile_h=0, flags=0, stats=None):
        img = np.zeros((self.height, self.width, 3))
        rp = RenderParams(spp, seed, 0, 1, tile_w, tile_h, flags)
        check(lib().mcpt_multi_render(self._h, C.byref(rp), _p(img, C.c_double), C.byref(stats) if stats is not None else None))
        return img

    def render_device(self, spp, seed=0, tile_w=0, tile_h=0, flags=0, stats=None):
        """The frame stays in devices[0]'s HBM; returns its device address (valid until the next call)."""
        rp = RenderParams(spp, seed, 0, 1, tile_w, tile_h, flags)
        d_img = C.c_void_p()
        check(lib().mcpt_multi_render_device(self._h, C.byref(rp), C.byref(d_img), C.byref(stats) if stats is not None else None))
        return d_img.value

    def collect_stats(self, stats=None):
        """statistics of every RENDER_KEEP_STATS frame since the last call, summed over the GPUs"""
        stats = stats if stats is not None else Stats()
        check(lib().mcpt_multi_collect_stats(self._h, C.byref(stats)))
        return stats

    def last_timing(self):
        """(render_ms per rank, gather_ms, ranks the RCCL communicator reports -- 0 with peer copies) of the last frame"""
        n = self.num_devices
        render_ms = np.zeros(n)
        gather_ms = C.c_double()
        comm = C.c_int32()
        check(lib().mcpt_multi_last_timing(self._h, _p(render_ms, C.c_double), C.byref(gather_ms), C.byref(comm)))
        return render_ms, gather_ms.value, comm.value

    def close(self):
        if getattr(self, "_h", None):
            lib().mcpt_multi_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def imshow_rgb8(img):
    """The 8-bit conversion of imshow (MTPC/MTPC.cpp:22-30)."""
    img = np.ascontiguousarray(img, dtype=np.float64)
    out = np.zeros(img.shape, dtype=np.uint8)
    check(lib().mcpt_quantize_rgb8(_p(img, C.c_double), img.size, _p(out, C.c_uint8)))
    return out


def png_bytes(rgb8):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    cap = 128 + h * (w * 3 + 6)
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().mcpt_png_encode(_p(rgb8, C.c_uint8), w, h, _p(out, C.c_uint8), cap)
    if n < 0:
        check(int(n))
    return out[:n].tobytes()


def write_png(file, rgb8, deflate=False):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    check((lib().mcpt_write_png_deflate if deflate else lib().mcpt_write_png)(file.encode(), _p(rgb8, C.c_uint8), w, h))


def png_bytes_deflate(rgb8):
    """The same picture as a compressed PNG (per-row filters + deflate)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    cap = 1024 + h * (w * 3 + 1) * 2
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().mcpt_png_encode_deflate(_p(rgb8, C.c_uint8), w, h, _p(out, C.c_uint8), cap)
    if n < 0:
        check(int(n))
    return out[:n].tobytes()


def write_pfm(file, img):
    """Linear radiance [H,W,3] as a little-endian fp32 Portable Float Map."""
    img = np.ascontiguousarray(img, dtype=np.float64)
    h, w, _ = img.shape
    check(lib().mcpt_write_pfm(file.encode(), _p(img, C.c_double), w, h))


def checkpoint_save(file, scene, img, spp, seed, done):
    img = np.ascontiguousarray(img, dtype=np.float64)
    done = np.ascontiguousarray(done, dtype=np.uint8)
    check(lib().mcpt_checkpoint_save(file.encode(), scene._h, _p(img, C.c_double), spp, seed, done.shape[0], _p(done, C.c_uint8)))


def checkpoint_load(file, scene, spp, seed, parts):
    """(img [H,W,3], done [parts]) of a matching checkpoint; McptError (code ERR_IO / ERR_PARSE) otherwise."""
    i = scene.info
    img = np.zeros((i.height, i.width, 3))
    done = np.zeros(parts, dtype=np.uint8)
    check(lib().mcpt_checkpoint_load(file.encode(), scene._h, _p(img, C.c_double), spp, seed, parts, _p(done, C.c_uint8)))
    return img, done


def decode_jpeg(file):
    """8-bit BGR raster [rows, cols, 3] of a JPEG file, as cv::imread would hand it to Material::readinMap."""
    w = np.zeros(1, dtype=np.int32)
    h = np.zeros(1, dtype=np.int32)
    check(lib().mcpt_decode_jpeg(file.encode(), _p(w, C.c_int32), _p(h, C.c_int32), None, 0))
    out = np.zeros((int(h[0]), int(w[0]), 3), dtype=np.uint8)
    check(lib().mcpt_decode_jpeg(file.encode(), _p(w, C.c_int32), _p(h, C.c_int32), _p(out, C.c_uint8), out.size))
    return out


def morton_code(x, y, z):
    return lib().mcpt_morton_code(x, y, z)


def render_scene(path, filename, N_ray_per_pixel, seed=0, device=0, width=0, height=0, quiet=True, output_prefix=None, stats=None,
                 load_flags=0, output_flags=0, checkpoint=None, checkpoint_parts=0, devices=None, gather=GATHER_PEER):
    """render_scene(path, filename, N) of MTPC/MTPC.cpp:35; writes <prefix>-SPP<N>.png (default ../result/<filename>).
    devices: list of GPU ordinals, or -1 for every visible GPU (the frame is then rendered by mcpt_multi_*)."""
    dev_arr, ndev = None, 0
    if devices == -1:
        ndev = -1
    elif devices is not None:
        keep = np.ascontiguousarray(devices, dtype=np.int32)
        dev_arr, ndev = _p(keep, C.c_int32), keep.shape[0]
    o = RenderSceneOptions(seed, device, width, height, int(quiet), output_prefix.encode() if output_prefix else None,
                           load_flags, output_flags, checkpoint.encode() if checkpoint else None, checkpoint_parts, 0,
                           ndev, gather, dev_arr)
    check(lib().mcpt_render_scene_opts(path.encode(), filename.encode(), N_ray_per_pixel, C.byref(o), C.sizeof(o),
                                       C.byref(stats) if stats is not None else None))
    return True
