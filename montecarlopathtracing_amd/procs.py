"""One process per GPU without torch in the process.

`python -m torch.distributed.run --nproc-per-node N bench.py ...` (or any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK) starts the
ranks; each rank renders its tiles with its own mcpt_device and the frames are gathered into rank 0's HBM over an RCCL communicator
that the ranks build themselves (mcpt_comm_*, csrc/proc_comm.cpp).  torch is deliberately NOT imported: a process that imports torch
runs libmcpt.so's kernels on the wheel's bundled HIP runtime instead of the release they were compiled against, and the
library refuses that since version 105.  What the ranks need from a launcher is only their rank, the world size and a way to pass
RCCL's 128-byte unique id from rank 0 to the others -- a file in the node's temporary directory, named after the launcher's process id
(all ranks are children of one launcher process) and the rendezvous port, written atomically by rank 0 and polled by the others.

The torch.distributed form of the same partition / gather is dist.py (kept for the gloo rehearsal on CPU and as `--launcher torch`)."""
import ctypes as C
import os
import tempfile
import time

import numpy as np

from ._lib import RenderParams, check, lib

ID_BYTES = 128


def launch_env():
    """(rank, world, local_rank) from the launcher's environment"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))


def rendezvous_file(tag=None):
    """where rank 0 leaves the unique id: one name per launch (the ranks share their parent, the launcher)"""
    if tag is None:
        tag = os.environ.get("MCPT_RDZV_TAG") or "%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))
    return os.path.join(os.environ.get("MCPT_RDZV_DIR") or tempfile.gettempdir(), "mcpt_rdzv_%s" % tag)


def exchange_id(rank, world, make_id, path=None, timeout=120.0, started=None):
    """rank 0: make_id() -> bytes, published; other ranks: wait for it.  The record carries rank 0's clock so that a file left behind by
    an earlier launch under the same name is not taken for this one's."""
    path = path or rendezvous_file()
    started = time.time() if started is None else started
    if world == 1:
        return make_id()
    if rank == 0:
        ident = make_id()
        assert len(ident) == ID_BYTES
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as fh:
            fh.write(np.float64(time.time()).tobytes() + ident)
        os.replace(tmp, path)                      # atomic: a reader sees nothing or everything
        return ident
    deadline = time.time() + timeout
    while time.time() < deadline:
        try:
            raw = open(path, "rb").read()
        except OSError:
            raw = b""
        if len(raw) == 8 + ID_BYTES and float(np.frombuffer(raw[:8], dtype=np.float64)[0]) >= started - 300.0:
            return raw[8:]
        time.sleep(0.02)
    raise TimeoutError("rank %d: no unique id from rank 0 under %s within %.0f s" % (rank, path, timeout))


class ProcessGroup:
    """The ranks of one launch: RCCL communicator over xGMI, gather of the frame into rank 0, barrier, small all-reduce."""

    def __init__(self, ordinal, rank=None, world=None, rdzv_path=None, started=None):
        env_rank, env_world, _ = launch_env()
        self.rank = env_rank if rank is None else rank
        self.world = env_world if world is None else world
        self.ordinal = ordinal
        self._path = rdzv_path or rendezvous_file()

        def make_id():
            buf = (C.c_uint8 * ID_BYTES)()
            n = lib().mcpt_comm_unique_id(buf, ID_BYTES)
            if n != ID_BYTES:
                check(n if n < 0 else -3)
            return bytes(buf)
        ident = exchange_id(self.rank, self.world, make_id, self._path, started=started)
        self._h = C.c_void_p()
        arr = (C.c_uint8 * ID_BYTES).from_buffer_copy(ident)
        check(lib().mcpt_comm_create(ordinal, self.rank, self.world, arr, ID_BYTES, C.byref(self._h)))

    def size(self):
        return lib().mcpt_comm_size(self._h)

    def gather_frame(self, scene, d_frame_ptr, tile_w=0, tile_h=0, stream=None):
        rp = RenderParams(spp=1, seed=0, rank=self.rank, world=self.world, tile_w=tile_w, tile_h=tile_h, flags=0)
        check(lib().mcpt_comm_gather_frame(self._h, scene._h, C.byref(rp), C.c_void_p(d_frame_ptr), stream))

    def allreduce(self, values, op="sum"):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        assert v.size <= 64
        check(lib().mcpt_comm_allreduce(self._h, v.ctypes.data_as(C.POINTER(C.c_double)), v.size, 0 if op == "sum" else 1))
        return v

    def barrier(self):
        check(lib().mcpt_comm_allreduce(self._h, None, 0, 0))

    def close(self):
        if getattr(self, "_h", None):
            lib().mcpt_comm_free(self._h)
            self._h = None
            if self.rank == 0 and self.world > 1:
                try:
                    os.remove(self._path)
                except OSError:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass
