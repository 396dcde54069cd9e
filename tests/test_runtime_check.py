"""Which HIP runtime do libmcpt.so's kernels run on?  (the likeliest cause of round 2's abort was hipcc-7.2 code objects
on the torch wheel's ROCm-7.0 libamdhip64 -- same soname, whichever is loaded first wins.)  The library now compares the version it was
compiled against with the runtime's at the first device creation and refuses a mismatch unless told otherwise.  The comparison is a
pure function: tested here without a GPU; that a normal process passes it is a -m gpu test."""
import pytest

ERR_HIP = -5


def enc(major, minor, patch=0):
    return major * 10000000 + minor * 100000 + patch


def test_same_release_passes(mcpt):
    rc, msg = mcpt.hip_runtime_check(enc(7, 2, 26015), enc(7, 2, 26015), "/opt/rocm/lib/libamdhip64.so.7")
    assert rc == 0 and msg == ""
    # a patch level is not a release
    rc, msg = mcpt.hip_runtime_check(enc(7, 2, 26015), enc(7, 2, 1), "x")
    assert rc == 0 and msg == ""


@pytest.mark.parametrize("compiled,runtime", [(enc(7, 2, 26015), enc(7, 0, 51831)), (enc(7, 0, 1), enc(7, 2, 1)), (enc(7, 2, 5), enc(6, 2, 5)),
                                              (enc(7, 2, 5), 0), (0, enc(7, 2, 5))])
def test_other_release_is_refused_with_both_versions_named(mcpt, compiled, runtime):
    path = "/usr/local/lib/python3.10/dist-packages/torch/lib/libamdhip64.so"
    rc, msg = mcpt.hip_runtime_check(compiled, runtime, path)
    assert rc == ERR_HIP
    assert "(%d)" % compiled in msg and "(%d)" % runtime in msg and path in msg
    assert "mcpt_allow_runtime_mismatch" in msg
    # the text fits a small buffer without overrunning it
    import ctypes as C
    small = C.create_string_buffer(b"\xff" * 64, 64)
    assert mcpt.lib().mcpt_hip_runtime_check(compiled, runtime, path.encode(), small, 32) == ERR_HIP
    assert small.raw[31:32] == b"\x00" and small.raw[32:] == b"\xff" * 32


def _info_in_fresh_process(prelude):
    """hip_runtime_info() of a new interpreter (this pytest process may have imported torch in an earlier test: whichever runtime came
    first stays)"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    code = "import sys, json; sys.path.insert(0, %r); %s; import montecarlopathtracing_amd as M; i = M.hip_runtime_info(); " \
           "print(json.dumps([i[0], i[1], i[2], M.hip_runtime_check(*i)[0]]))" % (ROOT, prelude)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_runtime_info_reports_what_is_loaded():
    compiled, runtime, path, rc = _info_in_fresh_process("pass")
    assert compiled // 10000000 >= 6 and "libamdhip64" in path
    # nothing but libmcpt.so pulled a HIP runtime into that process: the one it was linked against
    assert rc == 0 and "torch" not in path, (compiled, runtime, path)


def test_a_process_that_imported_torch_first_is_noticed():
    """The case DESIGN 8a is about: torch's wheel bundles a libamdhip64 under the same soname; imported first, it is the runtime
    libmcpt.so's kernels would run on.  Where the wheel's release differs from the one libmcpt.so was compiled against (this image:
    7.0 against 7.2) the check must say so."""
    compiled, runtime, path, rc = _info_in_fresh_process("import torch")
    assert "libamdhip64" in path
    if "torch" not in path:
        pytest.skip("torch did not bring a HIP runtime of its own here")
    same_release = compiled // 100000 == runtime // 100000
    assert (rc == 0) == same_release, (compiled, runtime, path, rc)


@pytest.mark.gpu
def test_device_creation_passes_the_gate_in_a_normal_process(mcpt):
    """... and is refused, with the reason, when the versions are made to differ (the gate is asked before any device is touched)."""
    import os
    from conftest import SCENES
    compiled, runtime, path = mcpt.hip_runtime_info()
    assert (compiled // 100000) == (runtime // 100000), "the GPU test process runs libmcpt.so on another HIP release: %s" % path
    sc = mcpt.Scene(SCENES, "cornell-box", width=32, height=18)
    dev = mcpt.Device(sc, 0)
    assert dev.generateImg(1, seed=1).sum() > 0
    dev.close()
    sc.close()
    assert "MCPT_ALLOW_RUNTIME_MISMATCH" not in os.environ


@pytest.mark.gpu
def test_scene_may_be_freed_before_its_devices(mcpt):
    """A device shares ownership of the scene it was created from (include/mcpt.h: mcpt_scene_free): freeing the caller's handle first
    leaves the device whole -- it renders, and its own release is the one that lets the scene go (that order used to be a
    write into freed memory)."""
    import numpy as np
    from conftest import SCENES
    sc = mcpt.Scene(SCENES, "cornell-box", width=64, height=36)
    dev = mcpt.Device(sc, 0)
    a = dev.generateImg(2, seed=4)
    sc.close()
    b = dev.generateImg(2, seed=4)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)) and a.sum() > 0
    dev.close()
