"""Not gpu: the one-process-per-GPU launcher form of the N>1 path (bench.py --launcher torch / dist.py) with world_size 2 on the
gloo backend.  The tiles come from the CPU oracle (test infrastructure) so that the partition + gather + scatter assembly is
checked end to end without a GPU.

torch is imported inside the functions, not at module level: pytest imports every test module while collecting, also under
`-m gpu`, and the torch wheel bundles its own copy of the HIP runtime (same soname as /opt/rocm's) -- imported first, it is the
runtime libmcpt.so would then run on.  The GPU tests run on the runtime libmcpt.so was built against."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import montecarlopathtracing_amd as M
    from montecarlopathtracing_amd.dist import gather_frame, scatter_into_frame
    import oracle_lib as O
    W, H = 96, 40
    sc = M.Scene(SCENES, "veach-mis", width=W, height=H)
    lists = [sc.owned_pixels(r, world) for r in range(world)]
    counts = [len(l) for l in lists]
    # every rank renders ONLY its own pixels (oracle, deterministic in (pixel, sample) -> independent of the rank count)
    osc = O.OracleScene(SCENES + "veach-mis", texture_dir=SCENES, width=W, height=H)
    full = osc.render(2, seed=11)
    local = torch.zeros((H * W, 3), dtype=torch.float64)
    mine = torch.from_numpy(lists[rank].astype(np.int64))
    local[mine] = torch.from_numpy(full.reshape(-1, 3))[mine]
    bufs = gather_frame(local, mine, counts, rank, world)
    if rank == 0:
        frame = torch.zeros((H * W, 3), dtype=torch.float64)
        scatter_into_frame(frame, bufs, [torch.from_numpy(l.astype(np.int64)) for l in lists])
        np.save(os.path.join(tmp, "frame.npy"), frame.numpy().reshape(H, W, 3))
        np.save(os.path.join(tmp, "full.npy"), full)
    else:
        assert bufs is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gather_assembles_the_frame(world, tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    frame = np.load(tmp_path / "frame.npy")
    full = np.load(tmp_path / "full.npy")
    assert np.array_equal(frame.view(np.uint64), full.view(np.uint64))


def _rdzv_worker(rank, world, path, out):
    import montecarlopathtracing_amd.procs as P
    ident = P.exchange_id(rank, world, lambda: bytes(range(128)), path, timeout=30.0)
    out.put((rank, ident))


def test_unique_id_rendezvous_between_processes(tmp_path):
    """What the one-process-per-GPU launcher form needs besides RCCL itself (montecarlopathtracing_amd/procs.py; no torch, no GPU): rank 0
    publishes the 128-byte unique id atomically under a name every rank of a launch derives for itself, the others wait for it; a
    record left behind by an earlier launch under the same name is not taken for this one's."""
    import multiprocessing as mp
    import time
    import numpy as np
    import montecarlopathtracing_amd.procs as P
    path = str(tmp_path / "rdzv")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, 3, path, q)) for r in (2, 1, 0)]       # (the readers start first)
    for p_ in procs[:2]:
        p_.start()
    time.sleep(0.3)
    procs[2].start()
    got = dict(q.get(timeout=60) for _ in range(3))
    for p_ in procs:
        p_.join(30)
    assert got[0] == got[1] == got[2] == bytes(range(128))
    # a stale record (ten minutes old) is ignored until rank 0 writes a fresh one
    with open(path, "wb") as fh:
        fh.write(np.float64(time.time() - 600.0).tobytes() + bytes(128))
    import pytest
    with pytest.raises(TimeoutError):
        P.exchange_id(1, 2, None, path, timeout=0.3)
    assert P.exchange_id(0, 1, lambda: b"x" * 128, path) == b"x" * 128                            # a launch of one needs no file
    assert P.rendezvous_file("abc").endswith("mcpt_rdzv_abc")
