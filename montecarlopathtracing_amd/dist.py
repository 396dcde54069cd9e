"""One process per GPU: pixel-tile partition of a frame and the gather of the framebuffer onto rank 0.

Pixels are independent and the scene is read-only, so the path shards with no data-path collective: the frame is
cut into 32x8 tiles dealt round-robin to ranks (mcpt_owned_pixels), every rank keeps its own resident copy of the
scene, and every (pixel, sample) has its own counter-RNG key, so the image does not depend on the rank count.
The only exchange is the end-of-frame gather of each rank's compact pixel buffer to rank 0 (RCCL over xGMI when the
tensors are on GPUs: backend "nccl"; "gloo" on CPU for tests).  1280x720x3 fp64 / 8 ranks = 2.8 MB per rank."""
import numpy as np
import torch
import torch.distributed as dist


def owned_counts(scene, world, tile_w=0, tile_h=0):
    return [int(scene.owned_pixels(r, world, tile_w, tile_h).shape[0]) for r in range(world)]


def gather_frame(local_flat, pixels, counts, rank, world, group=None, stage_on_cpu=False):
    """local_flat: [H*W, 3] tensor holding this rank's pixels at their frame positions; pixels: LongTensor of the
    owned indices (same device).  Returns the per-rank compact buffers on rank 0 (None elsewhere).
    stage_on_cpu: run the collective on host copies (gloo rehearsals with several ranks sharing one GPU)."""
    if world == 1:
        return local_flat
    nmax = max(counts)
    send = torch.zeros((nmax, 3), dtype=local_flat.dtype, device=local_flat.device)
    send[: pixels.shape[0]] = local_flat.index_select(0, pixels)
    if stage_on_cpu:
        send = send.cpu()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, bufs, dst=0, group=group)
    if rank != 0:
        return None
    if stage_on_cpu:
        bufs = [b.to(local_flat.device) for b in bufs]
    return bufs


def scatter_into_frame(frame_flat, bufs, pixel_lists):
    """rank 0: place every rank's compact buffer at its pixel positions."""
    for r, (buf, pix) in enumerate(zip(bufs, pixel_lists)):
        frame_flat.index_copy_(0, pix, buf[: pix.shape[0]])
    return frame_flat


class DistributedRenderer:
    """generateImg over `world` GPUs of one node; rank r drives GPU `local_rank`.
    pipeline=True: a renderer of frame SEQUENCES -- consecutive frames alternate between two frame buffers, two streams and the
    device's two frame slots (RENDER_PIPELINE | RENDER_KEEP_STATS), so the latency-bound tail of one frame overlaps the head of
    the next and no frame waits for its statistics; read them with device.collect_stats() after a torch.cuda.synchronize()."""

    def __init__(self, scene, device, rank=0, world=1, tile_w=0, tile_h=0, torch_device=None, stage_on_cpu=False, pipeline=False, gather=True):
        self.scene, self.device, self.rank, self.world = scene, device, rank, world
        self.stage_on_cpu = stage_on_cpu
        self.gather = gather                # False: this rank's tiles only, no exchange (one rank's share of an N-way frame, for timing)
        self.tile_w, self.tile_h = tile_w, tile_h
        self.torch_device = torch_device
        self.H, self.W = device.height, device.width      # the frame the device writes (fixed when it was created)
        self.pipeline = bool(pipeline) and torch_device is not None and torch_device.type == "cuda"
        n_frames = 2 if self.pipeline else 1
        self.frames = [torch.zeros((self.H * self.W, 3), dtype=torch.float64, device=torch_device) for _ in range(n_frames)]
        self.streams = [torch.cuda.Stream(device=torch_device) for _ in range(2)] if self.pipeline else None
        self.turn = 0
        self.frame = self.frames[0]
        self.pixel_lists = None
        self.counts = [self.H * self.W]
        if world > 1 and gather:
            lists = [scene.owned_pixels(r, world, tile_w, tile_h) for r in range(world)]
            self.counts = [int(l.shape[0]) for l in lists]
            self.pixels = torch.from_numpy(lists[rank].astype(np.int64)).to(torch_device)
            if rank == 0:
                self.pixel_lists = [torch.from_numpy(l.astype(np.int64)).to(torch_device) for l in lists]

    def _render_on_current_stream(self, frame, spp, seed, stats, flags):
        stream = torch.cuda.current_stream(self.torch_device).cuda_stream
        self.device.render_device(frame.data_ptr(), spp, seed, self.rank, self.world, self.tile_w, self.tile_h, flags, stats, stream)
        if self.world == 1 or not self.gather:
            return frame.view(self.H, self.W, 3)
        bufs = gather_frame(frame, self.pixels, self.counts, self.rank, self.world, stage_on_cpu=self.stage_on_cpu)
        if self.rank != 0:
            return None
        scatter_into_frame(frame, bufs, self.pixel_lists)
        return frame.view(self.H, self.W, 3)

    def render(self, spp, seed=0, stats=None, flags=0):
        """Renders this rank's tiles and gathers; returns the [H,W,3] tensor on rank 0 (pipelined: valid once its stream has
        finished -- torch.cuda.synchronize() -- and until the frame after next is started)."""
        if not self.pipeline:
            return self._render_on_current_stream(self.frame, spp, seed, stats, flags)
        from .api import RENDER_KEEP_STATS, RENDER_PIPELINE
        self.turn ^= 1
        with torch.cuda.stream(self.streams[self.turn]):
            return self._render_on_current_stream(self.frames[self.turn], spp, seed, None, flags | RENDER_KEEP_STATS | RENDER_PIPELINE)
