#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and sec/frame of cornell-box at 1280x720, SPP 256 (BASELINE.json configs[1]).

One step = one whole frame (the per-pixel integrator loop over every pixel and sample) rendered from the scene resident in
HBM; the frame is tile-partitioned over the N GPUs of one node and gathered into the first GPU inside the timed region.
Prints ONE JSON line (last line of stdout) on rank 0.  Two ways to drive N GPUs, the same kernels under both:

  python bench.py --gpus N --steps K --warmup W
      one process, the product's own multi-GPU path behind the C ABI: mcpt_multi_create(..., MCPT_GATHER_RCCL) /
      mcpt_multi_render_device -- a host thread per GPU inside libmcpt.so, ncclSend/ncclRecv of the compact pixel buffers.  This is
      what a caller of render_scene() (MTPC/MTPC.cpp:35-68) gets.  No torch in the process: libmcpt.so runs on the HIP runtime it
      was built against.  Fewer visible GPUs than N: exit code 2.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
      one process per GPU (RANK / WORLD_SIZE / LOCAL_RANK in the environment: --launcher procs): every rank renders its tiles with
      mcpt_render_device and the ranks gather them into rank 0's HBM over an RCCL communicator they build themselves (mcpt_comm_*,
      montecarlopathtracing_amd/procs.py) -- the launcher only starts the processes; torch is NOT imported by the ranks, so they too
      run on the HIP runtime the library was built against (a process that imports torch gets the wheel's: DESIGN.md 8a).
      --launcher torch: the older form, torch.distributed (backend nccl = RCCL) gathers (montecarlopathtracing_amd/dist.py).

Every N renders one frame at a time (frames_in_flight 1: a step's time is a frame's latency); --pipeline (one GPU, diagnostic)
keeps two frames in flight on two streams.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic bytes per unit of work, as the contract defines them: one ray query must move at least
#     32 B per node visited + 48 B per triangle tested + 64 B per ray (ray in, hit out, queue index),
# N_node / N_tri counted by the dominant kernel's own device counters in the same run.  The records this build actually fetches
# are fatter (64-B compressed 4-wide node, 104 B of fp64 triangle): that figure is reported beside it under its own name and is
# NOT the roofline.
BYTES_PER_NODE = 32
BYTES_PER_TRI = 48
BYTES_PER_RAY = 64
RECORD_BYTES_PER_NODE = 64
RECORD_BYTES_PER_TRI = 104
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec

COUNT_KEYS = ("rays", "node_visits", "tri_tests", "samples", "launches", "rays_primary", "rays_shadow", "rays_bounce", "dom_rays",
              "dom_node_visits", "dom_tri_tests")


def committed_profile(name_glob, build_id, kernel=None):
    """(path, json, reason): the newest profiles/rNN_final_<name> whose recorded build_id is the loaded library's.  bench.py cannot
    collect PMC counters of itself; tools/final_profile.sh does, with this same command, and stamps every file with the library's
    build id (a hash of the sources libmcpt.so was compiled from).  A profile of another build is not quoted."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", name_glob)))
    if not files:
        return None, None, "no committed profile matches %s" % name_glob
    for f in reversed(files):
        try:
            j = json.load(open(f))
        except (OSError, ValueError):
            continue
        if j.get("build_id") == build_id and (kernel is None or j.get("kernel") == kernel):
            return f, j, None
    return None, None, ("the committed profiles (%s) were taken with another build of libmcpt.so than the one loaded (build id %s): "
                        "their counters are not quoted" % (", ".join(os.path.relpath(f, ROOT) for f in files[-2:]), build_id))


def write_scene_dir(name, width, height):
    """Scene directory whose .camera has width/height replaced (the parser reads them, sceneManagement.cpp:233-240)."""
    src = os.path.join(ROOT, "scenes")
    d = tempfile.mkdtemp(prefix="mcpt_bench_")
    for ext in (".obj", ".mtl"):
        os.symlink(os.path.join(src, name + ext), os.path.join(d, name + ext))
    for f in os.listdir(src):
        if f.endswith(".ppm") or f.endswith(".jpg"):
            os.symlink(os.path.join(src, f), os.path.join(d, f))
    with open(os.path.join(src, name + ".camera"), "rb") as fh:
        lines = fh.read().decode().replace("\r", "").split("\n")
    out = []
    for ln in lines:
        if ln.startswith("width"):
            ln = "width %d" % width
        elif ln.startswith("height"):
            ln = "height %d" % height
        out.append(ln)
    with open(os.path.join(d, name + ".camera"), "w") as fh:
        fh.write("\n".join(out))
    return d + os.sep


def one_socket_cpus():
    """Logical CPUs that are the first hardware thread of each physical core of socket 0 (within our affinity mask)."""
    allowed = sorted(os.sched_getaffinity(0))
    try:
        seen, pick = set(), []
        for c in allowed:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            pkg = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
            if pkg == 0 and core not in seen:
                seen.add(core)
                pick.append(c)
        return pick or allowed
    except OSError:
        return allowed


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene_dir, name, seed, target_seconds=15.0, frame_spp=256):
    """The oracle (CPU port of the reference, reference-like cost: primary ray re-traced per sample, light CDF
    rebuilt per shade call) on a bounded sample of the same workload: every 8th row of the frame, all columns,
    at an SPP sized for ~target_seconds, on the physical cores of one socket (OpenMP over row x 64-column blocks;
    the reference itself forks <= 8 threads per pixel, MTPC/pathTracing.cpp:303)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    cpus = one_socket_cpus()
    old = os.sched_getaffinity(0)
    os.sched_setaffinity(0, cpus)
    try:
        cores = len(cpus)
        osc = O.OracleScene(scene_dir + name, texture_dir=scene_dir)
        stride = 8
        img = np.zeros((osc.height, osc.width, 3))
        # calibrate: SPP 1 pass, then size the timed pass
        st0 = O.Stats()
        t0 = time.time()
        osc.render_strided(1, seed, stride, faithful_cost=True, nthreads=cores, stats=st0, img=img)
        dt0 = max(time.time() - t0, 1e-3)
        spp = int(min(256, max(2, round(target_seconds / dt0))))
        st = O.Stats()
        t0 = time.time()
        osc.render_strided(spp, seed, stride, faithful_cost=True, nthreads=cores, stats=st, img=img)
        dt = time.time() - t0
        # The reference-style figure SURVEY 8(d) asks for beside it: the reference's own parallel structure -- one pixel at a time, its
        # samples on min(SPP, 8) OpenMP threads, a team forked and joined per pixel (MTPC/pathTracing.cpp:300-320) -- on a pixel lattice
        # of the same frame at the frame's SPP, sized for about a third of the time budget.
        ref_threads = min(frame_spp, 8)
        rs, cs = 48, 64
        sr0 = O.Stats()
        t0 = time.time()
        osc.render_reference_style(frame_spp, seed, rs * 4, cs * 4, stats=sr0, img=img)
        dtr0 = max(time.time() - t0, 1e-3)
        # (the calibration lattice is a sixteenth of the timed one)
        scale = max(1.0, min(16.0, (target_seconds / 2.0) / (dtr0 * 16.0)))
        rs2 = max(8, int(rs / scale ** 0.5)); cs2 = max(8, int(cs / scale ** 0.5))
        sr = O.Stats()
        t0 = time.time()
        osc.render_reference_style(frame_spp, seed, rs2, cs2, stats=sr, img=img)
        dtr = time.time() - t0
    finally:
        os.sched_setaffinity(0, old)
    rays = st.rays
    nrows = (osc.height + stride - 1) // stride
    reference_style = {
        "value": sr.rays / dtr / 1e6, "unit": "Mrays/s", "threads": ref_threads,
        "sample": "%s %dx%d, pixels (%d k, %d m) at SPP %d, seed %d: %d samples, %d rays in %.1f s; the oracle in reference-cost mode with the "
                  "reference's parallel structure: one pixel at a time, min(SPP, 8) = %d OpenMP threads over its samples, a team forked and "
                  "joined per pixel (MTPC/pathTracing.cpp:300-320)" % (name, osc.width, osc.height, rs2, cs2, frame_spp, seed, sr.samples, sr.rays, dtr, ref_threads),
        "samples_per_s": sr.samples / dtr, "seconds": dtr}
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "reference_style": reference_style,
            "sample": "%s %dx%d, rows 0,%d,%d,.. (%d rows, all columns) at SPP %d, seed %d: %d samples, %d rays in %.1f s; oracle in "
                      "reference-cost mode (primary ray re-traced per sample, light CDF rebuilt per shade call); OpenMP over "
                      "row x 64-column blocks on %d threads = physical cores of socket 0 (%s)"
                      % (name, osc.width, osc.height, stride, 2 * stride, nrows, spp, seed, st.samples, rays, dt, cores, cpu_model()),
            "samples_per_s": st.samples / dt, "rays_per_sample": rays / max(1, st.samples), "seconds": dt}


def make_scene(M, args, local_rank=0, talk=True):
    """(scene, scene_dir or None, build mode for the device handles)"""
    build_mode = {"default": None, "host": M.BUILD_HOST, "device": M.BUILD_DEVICE, "device_fast": M.BUILD_DEVICE_FAST, "device_sah": M.BUILD_DEVICE_SAH}[args.build]
    if args.scene == "synthetic":
        from montecarlopathtracing_amd import synthetic
        t_gen = time.perf_counter()
        scene = synthetic.make_scene(M, args.tris, defer_build=True, width=args.width, height=args.height)
        if talk:
            print("synthetic scene: %d triangles, generate+create %.1f s" % (scene.info.num_faces, time.perf_counter() - t_gen), file=sys.stderr)
        return scene, None, build_mode if build_mode is not None else M.BUILD_DEVICE
    if args.scene == "interior":
        # substitute for configs[3] (bedroom.obj is not shipped by the reference): generated textured interior, ~204 k triangles
        from montecarlopathtracing_amd import synthetic
        scene_dir = tempfile.mkdtemp(prefix="mcpt_interior_") + os.sep
        synthetic.write_interior(scene_dir, "interior", width=args.width, height=args.height)
        return M.Scene(scene_dir, "interior"), scene_dir, build_mode if build_mode is not None else M.BUILD_HOST
    scene_dir = write_scene_dir(args.scene, args.width, args.height)
    return M.Scene(scene_dir, args.scene), scene_dir, build_mode if build_mode is not None else M.BUILD_HOST


def add_stats(tot, st):
    for k in COUNT_KEYS:
        tot[k] += getattr(st, k)
    tot["ms_trace"] += st.ms_trace


def run_capi(args):
    """One process; N GPUs behind the C ABI (mcpt_multi_*), or one rank's share / a two-deep frame pipeline on one GPU."""
    import numpy as np
    import montecarlopathtracing_amd as M
    n = args.gpus
    visible = M.device_count()
    if visible < n:
        print("bench.py: --gpus %d but %d HIP device(s) visible" % (n, visible), file=sys.stderr)
        sys.exit(2)
    scene, scene_dir, build_mode = make_scene(M, args)
    H, W = scene.info.height, scene.info.width
    tot = {k: 0 for k in COUNT_KEYS}
    tot["ms_trace"] = 0.0
    extra = {"launcher": "capi: one process, mcpt_multi_* (host thread per GPU inside libmcpt.so)", "hip_runtime": M.hip_runtime_path()}
    frame_host = None
    single = args.sim_world > 1 or args.pipeline
    if single:
        # one device handle driven directly: one rank's tiles of an N-way partition (no exchange), or two frames in flight
        if n != 1:
            print("bench.py: --sim-world / --pipeline are one-GPU diagnostics", file=sys.stderr)
            sys.exit(2)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hip_rt
        t_dev = time.perf_counter()
        dev = M.Device(scene, 0, build=build_mode)
        print("device create %.2f s" % (time.perf_counter() - t_dev), file=sys.stderr)
        rank, world = (args.sim_rank, args.sim_world) if args.sim_world > 1 else (0, 1)
        nbuf = 2 if args.pipeline else 1
        bufs = [hip_rt.DeviceBuffer(H * W * 24) for _ in range(nbuf)]
        streams = [hip_rt.Stream() for _ in range(nbuf)]
        flags = (M.RENDER_PIPELINE | M.RENDER_KEEP_STATS) if args.pipeline else M.RENDER_KEEP_STATS
        turn = [0]

        def frame(stats):
            t = turn[0] = (turn[0] + 1) % nbuf
            dev.render_device(bufs[t].ptr.value, args.spp, args.seed, rank, world, args.tile_w, args.tile_h, flags=flags, stats=None, stream=streams[t].h.value)
            if not args.pipeline:
                streams[t].synchronize()

        def sync():
            for s in streams:
                s.synchronize()

        for _ in range(max(args.warmup, nbuf if args.pipeline else 0)):   # (both frame slots get their workspace outside the timed region)
            frame(M.Stats())
        sync()
        dev.collect_stats()
        st = M.Stats()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            frame(st)
        sync()
        elapsed = time.perf_counter() - t0
        dev.collect_stats(st)
        add_stats(tot, st)
        extra["frames_in_flight"] = nbuf
        if args.save_png:
            frame_host = np.zeros((H, W, 3))
            bufs[turn[0]].to_host_async(frame_host, streams[turn[0]].h)
            sync()
        n_ranks_timed = 1
    else:
        t_dev = time.perf_counter()
        gather = M.GATHER_RCCL
        try:
            md = M.MultiDevice(scene, list(range(n)), build=build_mode, gather=gather)
        except M.McptError as e:
            if n > 1:
                raise
            # one GPU: there is no exchange, the communicator only exists to be counted -- say so and go on without it
            extra["rccl_error"] = str(e)
            gather = M.GATHER_PEER
            md = M.MultiDevice(scene, [0], build=build_mode, gather=gather)
        print("scene on %d GPU(s) %.2f s" % (n, time.perf_counter() - t_dev), file=sys.stderr)
        st = M.Stats()
        for _ in range(args.warmup):
            md.render_device(args.spp, args.seed, args.tile_w, args.tile_h, flags=M.RENDER_KEEP_STATS)
        md.collect_stats()                            # (discard what the untimed frames left)
        t0 = time.perf_counter()
        for _ in range(args.steps):                   # every call returns with the frame complete in GPU 0's HBM: all streams synchronised
            md.render_device(args.spp, args.seed, args.tile_w, args.tile_h, flags=M.RENDER_KEEP_STATS)
        elapsed = time.perf_counter() - t0
        md.collect_stats(st)                          # counters and the event pairs recorded inside the timed region, read after it
        add_stats(tot, st)
        render_ms, gather_ms, comm_ranks = md.last_timing()
        extra.update({"rccl_ranks": comm_ranks, "gather": "rccl" if gather == M.GATHER_RCCL else "peer",
                      "per_rank_render_ms": [float(x) for x in render_ms], "gather_ms": gather_ms, "frames_in_flight": 1})
        if args.save_png:
            frame_host = md.generateImg(args.spp, args.seed)
        n_ranks_timed = n
    # ms_trace of a multi-GPU frame is the slowest rank's; launches are summed over ranks
    return {"elapsed": elapsed, "tot": tot, "world": n, "scene": scene, "scene_dir": scene_dir, "frame": frame_host, "extra": extra,
            "build_id": M.build_id(), "launch_ranks": n_ranks_timed, "M": M, "engine": scene.trace_engine()}


def run_procs(args):
    """One process per GPU under a launcher (torch.distributed.run sets RANK / WORLD_SIZE / LOCAL_RANK), WITHOUT torch in the process:
    every rank renders its tiles with its own mcpt_device on the HIP runtime libmcpt.so was compiled against, the frames are gathered
    into rank 0's HBM over an RCCL communicator the ranks build themselves (mcpt_comm_*, montecarlopathtracing_amd/procs.py), the timed
    region is bracketed by that communicator's barrier and a stream synchronisation, and the time is the maximum over the ranks."""
    import numpy as np
    import montecarlopathtracing_amd as M
    from montecarlopathtracing_amd.procs import ProcessGroup, launch_env
    t_start = time.time()
    rank, world, local_rank = launch_env()
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)" % (args.gpus, world, args.gpus), file=sys.stderr)
        sys.exit(2)
    if M.device_count() <= local_rank:
        print("bench.py: rank %d wants GPU %d, %d visible" % (rank, local_rank, M.device_count()), file=sys.stderr)
        sys.exit(2)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hip_rt
    pg = ProcessGroup(local_rank, rank, world, started=t_start)
    scene, scene_dir, build_mode = make_scene(M, args, talk=rank == 0)
    dev = M.Device(scene, local_rank, build=build_mode)
    H, W = dev.height, dev.width
    hip_rt.set_device(local_rank)                   # (the frame buffer and the stream belong to this rank's GPU)
    frame_buf = hip_rt.DeviceBuffer(H * W * 24)
    stream = hip_rt.Stream()
    flags = M.RENDER_KEEP_STATS

    def frame():
        dev.render_device(frame_buf.ptr.value, args.spp, args.seed, rank, world, args.tile_w, args.tile_h, flags=flags, stats=None, stream=stream.h.value)
        pg.gather_frame(scene, frame_buf.ptr.value, args.tile_w, args.tile_h, stream=stream.h.value)      # (returns when this rank's part is done)

    def sync():
        stream.synchronize()
        pg.barrier()
        stream.synchronize()

    for _ in range(args.warmup):
        frame()
    sync()
    dev.collect_stats()
    st = M.Stats()
    tot = {k: 0 for k in COUNT_KEYS}
    tot["ms_trace"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    sync()
    elapsed = time.perf_counter() - t0
    dev.collect_stats(st)
    add_stats(tot, st)
    mx = pg.allreduce([elapsed, tot["ms_trace"]], op="max")
    sm = pg.allreduce([float(tot[k]) for k in COUNT_KEYS], op="sum")
    elapsed, tot["ms_trace"] = float(mx[0]), float(mx[1])
    for i, k in enumerate(COUNT_KEYS):
        tot[k] = float(sm[i])
    frame_host = None
    if args.save_png and rank == 0:
        frame_host = np.zeros((H, W, 3))
        frame_buf.to_host_async(frame_host, stream.h)
        stream.synchronize()
    comm_ranks = pg.size()
    pg.barrier()
    pg.close()
    if rank != 0:
        sys.exit(0)
    extra = {"launcher": "procs: one process per GPU, mcpt_render_device per rank, gather over an RCCL communicator of the processes (mcpt_comm_*); no torch in the process",
             "hip_runtime": M.hip_runtime_path(), "rccl_ranks": comm_ranks, "gather": "rccl", "frames_in_flight": 1}
    return {"elapsed": elapsed, "tot": tot, "world": world, "scene": scene, "scene_dir": scene_dir, "frame": frame_host, "extra": extra,
            "build_id": M.build_id(), "launch_ranks": world, "M": M, "engine": scene.trace_engine()}


def run_torch(args):
    """One process per GPU under torch.distributed.run: mcpt_render_device per rank, torch.distributed gather (dist.py)."""
    import torch
    import torch.distributed as dist
    import montecarlopathtracing_amd as M
    from montecarlopathtracing_amd.dist import DistributedRenderer
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)" % (args.gpus, world, args.gpus), file=sys.stderr)
        sys.exit(2)
    # MCPT_BENCH_SHARE_GPU=1: rehearsal of the N-rank path on a one-GPU box -- every rank uses GPU 0 and the gather runs over
    # gloo on host copies (RCCL refuses two ranks on one device).  Not a measurement configuration.
    share_gpu = os.environ.get("MCPT_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        print("bench.py: rank %d wants GPU %d, %d visible" % (rank, local_rank, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    tdev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=tdev)
    # This process imported torch first, so libmcpt.so's kernels run on the wheel's bundled HIP runtime, not on the release they were
    # compiled against: the library refuses that unless told (mcpt_allow_runtime_mismatch); this launcher is the one place that tells
    # it, and the line reports both versions under "hip_runtime".
    M.allow_runtime_mismatch(True)
    scene, scene_dir, build_mode = make_scene(M, args, talk=rank == 0)
    dev = M.Device(scene, local_rank, build=build_mode)
    rr = DistributedRenderer(scene, dev, rank, world, torch_device=tdev, stage_on_cpu=share_gpu, pipeline=False)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        rr.render(args.spp, args.seed)
    sync()
    stats = M.Stats()
    tot = {k: 0 for k in COUNT_KEYS}
    tot["ms_trace"] = 0.0
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):
        frame = rr.render(args.spp, args.seed, stats=stats)
        add_stats(tot, stats)
    sync()
    elapsed = time.perf_counter() - t0
    red_dev = torch.device("cpu") if share_gpu else tdev
    vals = torch.tensor([elapsed, tot["ms_trace"]] + [float(tot[k]) for k in COUNT_KEYS], dtype=torch.float64, device=red_dev)
    if world > 1:
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        tot["ms_trace"] = float(mx[1])
        for i, k in enumerate(COUNT_KEYS):
            tot[k] = float(sm[2 + i])
    frame_host = frame.cpu().numpy() if (args.save_png and rank == 0 and frame is not None) else None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        sys.exit(0)
    extra = {"launcher": "torch: one process per GPU, mcpt_render_device per rank, torch.distributed gather (backend %s)" % ("gloo, shared GPU rehearsal" if share_gpu else "nccl = RCCL"),
             "hip_runtime": M.hip_runtime_path(), "hip_versions": "compiled %d, runtime %d (mismatch allowed explicitly)" % M.hip_runtime_info()[:2],
             "frames_in_flight": 1}
    return {"elapsed": elapsed, "tot": tot, "world": world, "scene": scene, "scene_dir": scene_dir, "frame": frame_host, "extra": extra,
            "build_id": M.build_id(), "launch_ranks": world, "M": M, "engine": scene.trace_engine()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell-box")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--launcher", default="auto", choices=["auto", "capi", "procs", "torch"],
                    help="auto: procs when started by torch.distributed.run (RANK / WORLD_SIZE set: one process per GPU, RCCL communicator of the "
                         "processes, no torch in them), else capi (one process, mcpt_multi_*); torch: the ranks gather through torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="size of the CPU baseline's bounded sample")
    ap.add_argument("--save-png", default=None)
    ap.add_argument("--tris", type=int, default=10_000_000, help="--scene synthetic: number of lattice triangles")
    ap.add_argument("--build", default="default", choices=["default", "host", "device", "device_fast", "device_sah"],
                    help="where the BVHs are built (mcpt_device_create_ex); device_fast / device_sah: the fast walk's hierarchy on the GPU too (Morton clusters / locally-ordered clustering)")
    ap.add_argument("--pipeline", action="store_true", help="one GPU, diagnostic: two frames in flight on two streams (a step is then 1/throughput, not a latency)")
    ap.add_argument("--sim-world", type=int, default=0, help="one GPU, diagnostic: render only one rank's tiles of an N-rank partition")
    ap.add_argument("--sim-rank", type=int, default=0, help="with --sim-world: which rank's tiles")
    ap.add_argument("--tile-w", type=int, default=0, help="tile width of the rank partition (0: the library's default, 32)")
    ap.add_argument("--tile-h", type=int, default=0, help="tile height of the rank partition (0: the library's default, 8)")
    args = ap.parse_args()

    # ONE JSON line on stdout: libraries that print banners to file descriptor 1 (RCCL's version block, Gloo's connection chatter)
    # are sent to stderr for the length of the run; the line itself goes to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    launcher = args.launcher
    if launcher == "auto":
        launcher = "procs" if int(os.environ.get("WORLD_SIZE", "0")) >= 1 and "RANK" in os.environ else "capi"
    res = run_torch(args) if launcher == "torch" else (run_procs(args) if launcher == "procs" else run_capi(args))
    M = res["M"]
    tot, elapsed, world, scene_dir, scene = res["tot"], res["elapsed"], res["world"], res["scene_dir"], res["scene"]
    steps = max(1, args.steps)
    sec_per_frame = elapsed / steps
    rays = float(tot["rays"])
    value = rays / elapsed / 1e6

    # dominant kernel = k_wf_trace (one launch per bounce iteration); algorithmic bytes of its launches from its own counters
    def alg_bytes(per_node, per_tri, per_ray):
        return per_node * tot["dom_node_visits"] + per_tri * tot["dom_tri_tests"] + per_ray * tot["dom_rays"]
    n_launch = max(1, tot["launches"])
    # average launch duration: the per-rank sums of HIP-event times / the per-rank launch counts (several GPUs: slowest rank's sum)
    avg_ms = tot["ms_trace"] / (n_launch / max(1, res["launch_ranks"]))
    per_launch = alg_bytes(BYTES_PER_NODE, BYTES_PER_TRI, BYTES_PER_RAY) / n_launch
    per_launch_rec = alg_bytes(RECORD_BYTES_PER_NODE, RECORD_BYTES_PER_TRI, BYTES_PER_RAY) / n_launch
    achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    record_rate = per_launch_rec / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch and issue utilisation: NOT measured by this run (a process cannot read PMC counters of itself) -- quoted from
    # the committed rocprofv3 --pmc passes of this same command, and only when they were taken with the build that is loaded now
    traffic, traffic_src, issue = None, None, None
    # the dominant kernel: one launch per bounce iteration of the closest-hit engine the library picked for this scene
    dom_kernel = "k_wf_trace_pool" if res.get("engine") == "pool" else "k_wf_trace"
    # workloads whose command tools/final_profile.sh also runs under rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes): tag of the file
    PROFILED = {("cornell-box", 1280, 720, 256): "final", ("veach-mis", 1280, 720, 100): "veach_mis", ("interior", 1280, 720, 256): "interior",
                ("synthetic", 1280, 720, 16): "synthetic10m"}
    tag = PROFILED.get((args.scene, args.width, args.height, args.spp)) if (world == 1 and args.sim_world <= 1 and not args.pipeline and
                                                                            (args.scene != "synthetic" or args.tris == 10_000_000)) else None
    headline = tag == "final"
    if tag is not None:
        tf, tj, why = committed_profile("r*_%s_hbm_traffic.json" % tag, res["build_id"], dom_kernel)
        if tj is not None:
            traffic = tj["bytes_per_launch"]
            traffic_src = ("from_committed_profile: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled per the gfx950 "
                           "note), taken with this build (%s); not measured in this run" % (os.path.relpath(tf, ROOT), res["build_id"]))
        else:
            traffic_src = why
        if headline:
            uf, uj, why = committed_profile("r*_final_issue_utilisation.json", res["build_id"], dom_kernel)
            if uj is not None:
                issue = uj
                issue["source"] = "from_committed_profile: %s, taken with this build; not measured in this run" % os.path.relpath(uf, ROOT)
            else:
                issue = {"source": why}
    else:
        traffic_src = "not a profiled workload: no PMC pass of this command is committed"
    config = {"workload": "%s %dx%d SPP=%d" % (args.scene, args.width, args.height, args.spp), "seed": args.seed,
              "partition": "32x8-pixel tiles dealt along a shifted diagonal over the ranks, compact pixel buffers gathered into rank 0's HBM",
              "primary_rays": "traced once per pixel (identical for every sample: the reference has no jitter)",
              "frames_in_flight": res["extra"].pop("frames_in_flight", 1), "launcher": res["extra"].pop("launcher")}
    if args.sim_world > 1:
        config["sim"] = "rank %d of %d rendered alone on one GPU, no exchange" % (args.sim_rank, args.sim_world)
    out = {
        "metric": "Mrays/s on %s %dx%d SPP=%d (closest-hit queries actually traced / wall time incl. gather)" % (args.scene, args.width, args.height, args.spp),
        "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sec_per_frame * 1e3, "sec_per_frame": sec_per_frame, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": ("reference scene %s.obj (in repo under scenes/), .camera rewritten to %dx%d, seed %d" % (args.scene, args.width, args.height, args.seed))
        if scene_dir is not None and args.scene != "interior" else
        ("generated textured interior (stand-in for the unshipped bedroom scene), %d triangles" % scene.info.num_faces if args.scene == "interior" else
         "synthetic lattice scene, %d triangles, generator seed 42 (montecarlopathtracing_amd/synthetic.py)" % scene.info.num_faces),
        "config": config,
        "latency_ms_per_frame": sec_per_frame * 1e3 if config["frames_in_flight"] == 1 else None,
        "rays_per_frame": rays / steps, "samples_per_frame": tot["samples"] / steps,
        "nodes_per_ray": tot["node_visits"] / max(1.0, rays), "tris_per_ray": tot["tri_tests"] / max(1.0, rays),
        "build_id": res["build_id"],
        "roofline": {"bound": "hbm", "kernel": dom_kernel, "engine": res.get("engine"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_ms": avg_ms, "launches": tot["launches"],
                     "algorithmic_bytes_per_launch": per_launch,
                     "bytes_per_unit": {"node_visit": BYTES_PER_NODE, "triangle_test": BYTES_PER_TRI, "ray": BYTES_PER_RAY,
                                        "source": "SURVEY.md 8(d); node visits and triangle tests counted by k_wf_trace's device counters in this run"},
                     "dom_nodes_per_ray": tot["dom_node_visits"] / max(1.0, tot["dom_rays"]), "dom_tris_per_ray": tot["dom_tri_tests"] / max(1.0, tot["dom_rays"]),
                     "record_bytes_rate_GBs": record_rate,
                     "record_bytes_note": "the same counts priced at this build's record sizes (64-B compressed 4-wide node, 104 B of an fp64 triangle): a cache-side rate, not a roofline",
                     "real_bound": "not HBM: the walk's nodes and triangles are served by L1/L2, fabric traffic is the streamed ray and hit records; the kernel is "
                                   "bound by instruction issue of a per-lane state machine (DESIGN.md 6)",
                     "issue_utilisation": issue,
                     "note": "launches of k_wf_trace, timed with HIP events on the launching stream"},
    }
    out.update(res["extra"])
    if args.save_png and res["frame"] is not None:
        M.write_png(args.save_png, M.imshow_rgb8(res["frame"]))
    if world == 1 and not args.no_cpu_baseline and scene_dir is not None and args.scene != "interior":
        cb = cpu_baseline(scene_dir, args.scene, args.seed, args.cpu_seconds, args.spp)
        out["cpu_baseline"] = cb
        out["gpu_over_cpu_mrays"] = value / cb["value"]
        # frame-time ratio: CPU seconds for the full frame extrapolated linearly in samples
        cpu_frame_s = (args.width * args.height * args.spp) / cb["samples_per_s"]
        out["cpu_sec_per_frame_extrapolated"] = cpu_frame_s
        out["gpu_over_cpu_frame_time"] = cpu_frame_s / sec_per_frame
        out["cpu_baseline"]["note"] = ("kind 'port' = the repo's C restatement of the reference in reference-cost mode, OpenMP over pixel blocks on every core of "
                                       "the socket (value); reference_style = the same code under the reference's own parallel structure, min(SPP, 8) threads over "
                                       "the samples of one pixel at a time, measured in this run.  Per sample the port is cheaper than the reference binary itself "
                                       "(no std::string copies, hardware popcount: SURVEY 6 measured the binary at 0.31 Mrays/s on 8 vCPUs, where the port in "
                                       "reference style does 0.29).  It traces the reference's 2.7 rays per sample, the GPU 1.6 (primary ray once per pixel, unused "
                                       "shadow rays skipped): compare frame times (gpu_over_cpu_frame_time), not Mrays/s")
        out["gpu_over_reference_style_mrays"] = value / cb["reference_style"]["value"]
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
