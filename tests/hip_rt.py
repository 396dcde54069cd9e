"""A few HIP runtime calls through ctypes, for GPU tests and bench.py's one-GPU diagnostics that need streams and device buffers
without torch (the torch wheel bundles its own copy of the HIP runtime under the same soname: whichever copy is loaded first is the one
libmcpt.so runs on -- INTEGRATION.md section 1 -- so the GPU test process and bench.py's default path keep torch out).  The library
handle resolves to the runtime libmcpt.so has already loaded."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        import montecarlopathtracing_amd as M
        M.lib()                                     # libmcpt.so pulls the runtime in
        L = C.CDLL("libamdhip64.so")
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        L.hipStreamDestroy.argtypes = [C.c_void_p]
        L.hipStreamSynchronize.argtypes = [C.c_void_p]
        L.hipSetDevice.argtypes = [C.c_int]
        _hip = L
    return _hip


def check(rc):
    assert rc == 0, "HIP error %d" % rc


def set_device(ordinal):
    """the calling thread's current device (buffers and streams below are created on it)"""
    check(hip().hipSetDevice(ordinal))


class DeviceBuffer:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        check(hip().hipMalloc(C.byref(self.ptr), nbytes))
        check(hip().hipMemset(self.ptr, 0, nbytes))

    def to_host_async(self, out, stream):
        check(hip().hipMemcpyAsync(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, 2, stream))

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = None


class Stream:
    def __init__(self):
        self.h = C.c_void_p()
        check(hip().hipStreamCreate(C.byref(self.h)))

    def synchronize(self):
        check(hip().hipStreamSynchronize(self.h))

    def destroy(self):
        if self.h:
            hip().hipStreamDestroy(self.h)
            self.h = None
