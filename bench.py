#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and sec/frame of cornell-box at 1280x720, SPP 256 (BASELINE.json configs[1]).

One step = one whole frame (the per-pixel integrator loop over every pixel and sample) rendered from the scene
resident in HBM; the frame is tile-partitioned over the N GPUs of one node and gathered to rank 0 (RCCL) inside
the timed region.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
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


def committed_profile(name_glob):
    """newest profiles/rNN_final_<name> (rocprofv3 --pmc passes of this same command; bench.py cannot collect PMC itself)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", name_glob)))
    return files[-1] if files else None


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


def cpu_baseline(scene_dir, name, seed, target_seconds=15.0):
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
    finally:
        os.sched_setaffinity(0, old)
    rays = st.rays
    nrows = (osc.height + stride - 1) // stride
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%s %dx%d, rows 0,%d,%d,.. (%d rows, all columns) at SPP %d, seed %d: %d samples, %d rays in %.1f s; oracle in "
                      "reference-cost mode (primary ray re-traced per sample, light CDF rebuilt per shade call); OpenMP over "
                      "row x 64-column blocks on %d threads = physical cores of socket 0 (%s)"
                      % (name, osc.width, osc.height, stride, 2 * stride, nrows, spp, seed, st.samples, rays, dt, cores, cpu_model()),
            "samples_per_s": st.samples / dt, "rays_per_sample": rays / max(1, st.samples), "seconds": dt}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="size of the CPU baseline's bounded sample")
    ap.add_argument("--save-png", default=None)
    ap.add_argument("--tris", type=int, default=10_000_000, help="--scene synthetic: number of lattice triangles")
    ap.add_argument("--build", default="default", choices=["default", "host", "device", "device_fast"],
                    help="where the BVHs are built (mcpt_device_create_ex); device_fast: the fast walk's hierarchy on the GPU too")
    ap.add_argument("--no-pipeline", action="store_true", help="one frame at a time (no overlap of a frame's tail with the next frame's head)")
    ap.add_argument("--sim-world", type=int, default=0, help="diagnostic: render only one rank's tiles of an N-rank partition on this GPU")
    ap.add_argument("--sim-rank", type=int, default=0, help="with --sim-world: which rank's tiles")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import montecarlopathtracing_amd as M
    from montecarlopathtracing_amd.dist import DistributedRenderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    # MCPT_BENCH_SHARE_GPU=1: rehearsal of the N-rank path on a one-GPU box -- every rank uses GPU 0 and the gather runs over
    # gloo on host copies (RCCL refuses two ranks on one device).  Not a measurement configuration.
    share_gpu = os.environ.get("MCPT_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    tdev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=tdev)

    build_mode = {"default": None, "host": M.BUILD_HOST, "device": M.BUILD_DEVICE, "device_fast": M.BUILD_DEVICE_FAST}[args.build]
    if args.scene == "synthetic":
        from montecarlopathtracing_amd import synthetic
        t_gen = time.perf_counter()
        scene = synthetic.make_scene(M, args.tris, defer_build=True, width=args.width, height=args.height)
        t_dev = time.perf_counter()
        dev = M.Device(scene, local_rank, build=build_mode)   # Morton + sort + BVH levels on the GPU; fast hierarchy: host SAH or GPU (--build device_fast)
        if rank == 0:
            print("synthetic scene: %d triangles, generate+create %.1f s, device build %.1f s" %
                  (scene.info.num_faces, t_dev - t_gen, time.perf_counter() - t_dev), file=sys.stderr)
        scene_dir = None
    elif args.scene == "interior":
        # substitute for configs[3] (bedroom.obj is not shipped by the reference): generated textured interior, ~204 k triangles
        from montecarlopathtracing_amd import synthetic
        scene_dir = tempfile.mkdtemp(prefix="mcpt_interior_") + os.sep
        synthetic.write_interior(scene_dir, "interior", width=args.width, height=args.height)
        scene = M.Scene(scene_dir, "interior")
        dev = M.Device(scene, local_rank, build=build_mode)
    else:
        scene_dir = write_scene_dir(args.scene, args.width, args.height)
        scene = M.Scene(scene_dir, args.scene)
        dev = M.Device(scene, local_rank, build=build_mode)
    # A renderer of a sequence of frames: two frames in flight (the device's two frame slots, two streams, two frame tensors), so
    # the latency-bound tail of frame i overlaps the head of frame i+1, and no frame waits for its statistics (they stay on the
    # device until the timed region is over).  --no-pipeline: one frame at a time, statistics read back after every frame.
    pipeline = not args.no_pipeline and not share_gpu and world == 1     # (with a gather per frame the ranks render one frame at a time)
    if args.sim_world > 1:          # one rank's share of an N-way partition, no communication
        rr = DistributedRenderer(scene, dev, args.sim_rank, args.sim_world, torch_device=tdev, pipeline=pipeline, gather=False)
    else:
        rr = DistributedRenderer(scene, dev, rank, world, torch_device=tdev, stage_on_cpu=share_gpu, pipeline=pipeline)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if pipeline:                    # both frame slots get their workspace (83 GB each for the headline frame) outside the timed region
        try:
            for _ in range(2):
                rr.render(args.spp, args.seed)
            torch.cuda.synchronize()
        except M.McptError as e:    # not enough free HBM for two frames in flight: one at a time
            print("pipelining off: %s" % e, file=sys.stderr)
            pipeline = False
            dev.close()
            dev = M.Device(scene, local_rank, build=build_mode)
            rr = DistributedRenderer(scene, dev, args.sim_rank if args.sim_world > 1 else rank, args.sim_world if args.sim_world > 1 else world,
                                     torch_device=tdev, stage_on_cpu=share_gpu, pipeline=False, gather=args.sim_world <= 1)
    for _ in range(args.warmup):
        rr.render(args.spp, args.seed)
    sync()
    if pipeline:
        dev.collect_stats()         # discard what the untimed frames left
    stats = M.Stats()
    tot = {"rays": 0, "node_visits": 0, "tri_tests": 0, "samples": 0, "ms_trace": 0.0, "launches": 0, "rays_primary": 0,
           "rays_shadow": 0, "rays_bounce": 0, "dom_rays": 0, "dom_node_visits": 0, "dom_tri_tests": 0}
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):
        frame = rr.render(args.spp, args.seed, stats=None if pipeline else stats)
        if not pipeline:
            for k in tot:
                tot[k] += getattr(stats, k)
    sync()
    elapsed = time.perf_counter() - t0
    if pipeline:                    # the events were recorded inside the timed region; they are read here, after it
        dev.collect_stats(stats)
        for k in tot:
            tot[k] = getattr(stats, k)

    red_dev = torch.device("cpu") if share_gpu else tdev
    vals = torch.tensor([elapsed] + [float(tot[k]) for k in ("rays", "node_visits", "tri_tests", "samples", "launches")] + [tot["ms_trace"]],
                        dtype=torch.float64, device=red_dev)
    if world > 1:
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        rays, nodes, tris, samples, launches = (float(sm[i]) for i in range(1, 6))
        ms_trace_max = float(mx[6])
    else:
        rays, nodes, tris, samples, launches = (float(vals[i]) for i in range(1, 6))
        ms_trace_max = tot["ms_trace"]

    if rank == 0:
        steps = max(1, args.steps)
        sec_per_frame = elapsed / steps
        value = rays / elapsed / 1e6
        # dominant kernel = k_wf_trace (one launch per bounce iteration); algorithmic bytes of its launches from its own counters
        def alg_bytes(per_node, per_tri, per_ray):
            return per_node * tot["dom_node_visits"] + per_tri * tot["dom_tri_tests"] + per_ray * tot["dom_rays"]
        alg_bytes_rank0 = alg_bytes(BYTES_PER_NODE, BYTES_PER_TRI, BYTES_PER_RAY)
        record_bytes_rank0 = alg_bytes(RECORD_BYTES_PER_NODE, RECORD_BYTES_PER_TRI, BYTES_PER_RAY)
        n_launch = max(1, tot["launches"])
        # HBM bytes per launch and issue utilisation: NOT measured by this run (a process cannot read PMC counters of itself) --
        # taken from the committed rocprofv3 --pmc passes of this same command and labelled as such; null for any other workload
        traffic, traffic_src, issue = None, None, None
        headline = args.scene == "cornell-box" and (args.width, args.height, args.spp) == (1280, 720, 256) and world == 1 and args.sim_world <= 1
        tf = committed_profile("r*_final_hbm_traffic.json")
        if headline and tf:
            tj = json.load(open(tf))
            traffic = tj["bytes_per_launch"]
            traffic_src = "from_committed_profile: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled per the gfx950 note); not measured in this run" % os.path.relpath(tf, ROOT)
        uf = committed_profile("r*_final_issue_utilisation.json")
        if headline and uf:
            issue = json.load(open(uf))
            issue["source"] = "from_committed_profile: %s; not measured in this run" % os.path.relpath(uf, ROOT)
        avg_ms = tot["ms_trace"] / n_launch
        achieved = (alg_bytes_rank0 / n_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        record_rate = (record_bytes_rank0 / n_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s on %s %dx%d SPP=%d (closest-hit queries actually traced / wall time incl. gather)" % (args.scene, args.width, args.height, args.spp),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec_per_frame * 1e3, "sec_per_frame": sec_per_frame, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": ("reference scene %s.obj (in repo under scenes/), .camera rewritten to %dx%d, seed %d" % (args.scene, args.width, args.height, args.seed))
            if scene_dir is not None else "synthetic lattice scene, %d triangles, generator seed 42 (montecarlopathtracing_amd/synthetic.py)" % scene.info.num_faces,
            "config": {"workload": "%s %dx%d SPP=%d" % (args.scene, args.width, args.height, args.spp), "seed": args.seed,
                       "partition": "32x8-pixel tiles round-robin over ranks, RCCL gather to rank 0",
                       "primary_rays": "traced once per pixel (identical for every sample: the reference has no jitter)",
                       "frames_in_flight": 2 if pipeline else 1},
            "rays_per_frame": rays / steps, "samples_per_frame": samples / steps,
            "nodes_per_ray": nodes / max(1.0, rays), "tris_per_ray": tris / max(1.0, rays),
            "roofline": {"bound": "hbm", "kernel": "k_wf_trace", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": avg_ms, "launches": tot["launches"],
                         "algorithmic_bytes_per_launch": alg_bytes_rank0 / n_launch,
                         "bytes_per_unit": {"node_visit": BYTES_PER_NODE, "triangle_test": BYTES_PER_TRI, "ray": BYTES_PER_RAY,
                                            "source": "SURVEY.md 8(d); node visits and triangle tests counted by k_wf_trace's device counters in this run"},
                         "record_bytes_rate_GBs": record_rate,
                         "record_bytes_note": "the same counts priced at this build's record sizes (64-B compressed 4-wide node, 104 B of an fp64 triangle): a cache-side rate, not a roofline",
                         "real_bound": "not HBM: the walk's 2.5 MB of nodes and triangles are served by L1/L2, fabric traffic is the streamed ray and hit records (traffic / peak ~ 7 %); "
                                       "the kernel is bound by instruction issue (VALU ~90 % busy at 4 waves per SIMD) of a state machine at ~57 % lane occupancy (DESIGN.md 6)",
                         "issue_utilisation": issue,
                         "note": "rank-0 launches of k_wf_trace, timed with HIP events on the launching stream"},
        }
        if args.save_png and frame is not None:
            img = frame.cpu().numpy()
            M.write_png(args.save_png, M.imshow_rgb8(img))
        if world == 1 and not args.no_cpu_baseline and scene_dir is not None and args.scene != "interior":
            cb = cpu_baseline(scene_dir, args.scene, args.seed, args.cpu_seconds)
            out["cpu_baseline"] = cb
            out["gpu_over_cpu_mrays"] = value / cb["value"]
            # frame-time ratio: CPU seconds for the full frame extrapolated linearly in samples
            cpu_frame_s = (args.width * args.height * args.spp) / cb["samples_per_s"]
            out["cpu_sec_per_frame_extrapolated"] = cpu_frame_s
            out["gpu_over_cpu_frame_time"] = cpu_frame_s / sec_per_frame
            out["cpu_baseline"]["note"] = ("kind 'port' = the repo's C restatement of the reference, OpenMP over pixel blocks; it is ~4x faster per core than the "
                                           "reference binary itself (no per-pixel fork/join, no std::string copies, hardware popcount; SURVEY 6: 0.31 Mrays/s on 8 vCPUs "
                                           "where the port does 1.2).  It traces the reference's 2.7 rays per sample, the GPU 1.6 (primary ray once per pixel, unused "
                                           "shadow rays skipped): compare frame times (gpu_over_cpu_frame_time), not Mrays/s")
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
