"""MI355X-native Monte-Carlo path tracer behind the surface of Arieys/MonteCarloPathTracing's render_scene.

The work is done by csrc/libmcpt.so (host C++ + hand-written HIP kernels for gfx950, C ABI in include/mcpt.h);
this package is the ctypes face of that ABI plus the one-process-per-GPU tile partition / RCCL gather driver."""
from ._lib import McptError, RenderParams, Stats, build, lib  # noqa: F401
from .api import (GATHER_PEER, GATHER_RCCL, MultiDevice, BUILD_DEVICE, BUILD_DEVICE_FAST, BUILD_DEVICE_SAH, BUILD_HOST, LOAD_MORTON_BOUNDS, LOAD_MTLLIB, LOAD_STANDARD_OBJ, OUT_PFM, OUT_PNG_DEFLATE, checkpoint_load, checkpoint_save, png_bytes_deflate, write_pfm, RENDER_DEFAULT, RENDER_MEGAKERNEL, RENDER_KEEP_STATS, RENDER_PIPELINE, TRACE_FAST, TRACE_REFERENCE, Device, Scene, build_id, decode_jpeg, device_count, hip_runtime_path, hip_runtime_info, hip_runtime_check, allow_runtime_mismatch, imshow_rgb8, morton_code, png_bytes, render_scene,  # noqa: F401
                  write_png)
