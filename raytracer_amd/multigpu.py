"""Sharding of the render path over the GPUs of one node.

Pixels and samples are independent (the seed is a pure function of x, y, sample index and
frameCount, GPURaytracer.cu:94), so the scene is replicated and there is no exchange inside a
frame.  Two partitionings, both pure host logic over a `render(rect, frame_index)` callback so
that they can be exercised on CPU with gloo:

 * tiles:   one frame, row bands dealt round-robin to the ranks (sky bands are cheap, so
            interleaving balances the load); composite = all_gather of each rank's bands.
            Seeds use global pixel coordinates, so the composite is bit-identical to the
            single-GPU frame.
 * frames:  progressive accumulation; frame index f belongs to rank f % world.  Every rank
            sums its own clamped frames, the sums are combined with one reduce(SUM), and the
            mean is sum / frames.  (The reference's running mean (c + prev*n)/(n+1),
            GPURaytracer.cu:129-134, differs from sum/N only by rounding.)
"""
from typing import List, Tuple

BAND_ROWS = 16  # multiple of the 8-row tiles.  A rank's bands run as one pass of the pipeline (rt_render_bands), so thin bands cost
                # nothing and balance better: 4K over 8 ranks is 135 bands, 17 or 16 per rank (64-row bands: 5 or 4)


def band_rects(width: int, height: int, rank: int, world: int, band_rows: int = BAND_ROWS) -> List[Tuple[int, int, int, int]]:
    """Row bands [y0, y1) owned by `rank`: band b belongs to rank b % world."""
    if world == 1:
        return [(0, 0, width, height)]
    rects = []
    nb = (height + band_rows - 1) // band_rows
    for b in range(rank, nb, world):
        rects.append((0, b * band_rows, width, min(height, (b + 1) * band_rows)))
    return rects


def frame_indices(first: int, count: int, rank: int, world: int) -> List[int]:
    """Frame indices in [first, first+count) owned by `rank`."""
    return [f for f in range(first, first + count) if f % world == rank]


def band_layout(height: int, world: int, band_rows: int = BAND_ROWS):
    """(bands in the frame, bands per rank's packet, rows of the padded frame): the padded frame holds
    per_rank * world whole bands, so that band b = k * world + r sits at [k, r] of a 5-D view."""
    nb = (height + band_rows - 1) // band_rows
    per_rank = (nb + world - 1) // world
    return nb, per_rank, per_rank * world * band_rows


class TileCompositor:
    """Composite of one frame rendered as interleaved row bands (rank r owns bands r, r + world, ...;
    rt_render_bands with (firstBand, bandStride) = (rank, world)).  The frame lives in a tensor padded to a
    whole number of band rounds, `frame` is its first `height` rows (wrap that as the framebuffer):

        pack    one strided copy: the rank's bands -> a contiguous packet (1/world of a frame)
        gather  one all_gather of equal-sized packets into [world, per_rank, band_rows, W, 4]
        unpack  one strided copy: gathered.permute(1, 0, ...) -> the padded frame

    No per-band Python loop, no allocation per frame.  Seeds use frame coordinates, so the composite is
    bit-identical to the single-GPU frame.

    `buffers` = 2 gives the pipelined form for a sequence of frames: frame i is rendered into buffer i % 2 and its
    composite runs on a side stream (GPU) while frame i + 1 renders into the other buffer, the way FrameAccumulator
    hides the reduce:

        b = tc.acquire(i)        # buffer index for frame i, its previous composite finished (a stream dependency)
        render(tc.frames[b])     # enqueue the rank's bands
        tc.submit(b)             # pack + all_gather + unpack, behind the render, beside the next one
        ...
        tc.drain()               # everything finished; tc.frames[b] hold complete frames
    """

    def __init__(self, torch, dist, width, height, rank, world, device=None, band_rows: int = BAND_ROWS, dtype=None, buffers: int = 1,
                 exchange=None):
        """exchange: run the pack / all_gather / unpack even with one rank (default: only when world > 1).  A one-rank RCCL group
        makes the same calls on the same streams as N ranks do, which is how the GPU tests reach this code on a one-GPU box."""
        self.torch, self.dist, self.rank, self.world, self.band_rows = torch, dist, rank, world, band_rows
        self.exchange = world > 1 if exchange is None else bool(exchange)
        self.width, self.height = width, height
        self.nb, self.per_rank, rows = band_layout(height, world, band_rows)
        dtype = dtype or torch.float32
        self.padded = [torch.zeros((rows, width, 4), dtype=dtype, device=device) for _ in range(buffers)]
        self.frames = [p[:height] for p in self.padded]
        self.views = [p.view(self.per_rank, world, band_rows, width, 4) for p in self.padded]
        self.packets = [torch.empty((self.per_rank, band_rows, width, 4), dtype=dtype, device=device) for _ in range(buffers)]
        self.gathered = [torch.empty((world, self.per_rank, band_rows, width, 4), dtype=dtype, device=device) for _ in range(buffers)]
        self.frame, self.view = self.frames[0], self.views[0]
        self.on_gpu = device is not None and torch.device(device).type == "cuda"
        # the side stream is for RCCL, whose collectives are stream-ordered.  gloo's own device-tensor path stalls for seconds
        # at a time on the GPU box when called off the default stream (tools/rehearse_probe.py), so under gloo (bench.py's
        # one-GPU rehearsal) the packet is staged through pinned host memory here, in line
        self.host = None
        self.side = torch.cuda.Stream(device=device) if self.on_gpu and buffers > 1 and self.exchange and dist.get_backend() == "nccl" else None
        self.done = [None] * buffers      # per buffer: event recorded on the side stream after the unpack

    def _composite(self, b):
        self.packets[b].copy_(self.views[b][:, self.rank])
        if self.dist.get_backend() == "nccl":   # RCCL: straight into the contiguous [world, ...] tensor
            self.dist.all_gather_into_tensor(self.gathered[b], self.packets[b])
        elif self.on_gpu:                       # gloo under device tensors: staged through pinned host memory by hand
            if self.host is None:
                self.host = (self.packets[b].cpu().pin_memory(), self.gathered[b].cpu().pin_memory())
            self.host[0].copy_(self.packets[b])
            self.dist.all_gather(list(self.host[1].unbind(0)), self.host[0])
            self.gathered[b].copy_(self.host[1])
        else:
            self.dist.all_gather(list(self.gathered[b].unbind(0)), self.packets[b])
        self.views[b].copy_(self.gathered[b].permute(1, 0, 2, 3, 4))

    def composite(self, b: int = 0):
        """All ranks call this after rendering their bands into frames[b]; on return (in stream order) it is complete."""
        if self.exchange:
            self._composite(b)
        return self.frames[b]

    def acquire(self, i: int) -> int:
        b = i % len(self.frames)
        if self.done[b] is not None:
            self.torch.cuda.current_stream().wait_event(self.done[b])
            self.done[b] = None
        return b

    def submit(self, b: int):
        if not self.exchange:
            return
        if self.side is None:
            self._composite(b)
            return
        torch = self.torch
        rendered = torch.cuda.Event()
        rendered.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(rendered)
            self._composite(b)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.done[b] = ev

    def drain(self):
        for b in range(len(self.frames)):
            if self.done[b] is not None:
                self.torch.cuda.current_stream().wait_event(self.done[b])
                self.done[b] = None


def composite_tiles(dist, local_fb, width, height, rank, world, band_rows: int = BAND_ROWS):
    """One-shot form of TileCompositor for a plain [H, W, 4] tensor (copies in and out of a padded frame)."""
    import torch

    if world == 1:
        return local_fb
    tc = TileCompositor(torch, dist, width, height, rank, world, local_fb.device, band_rows, local_fb.dtype)
    tc.frame.copy_(local_fb)
    return tc.composite().clone()


def reduce_frames(dist, local_sum, world, dst=0):
    """Combine per-rank sums of clamped frames: one reduce(SUM) to `dst`."""
    if world > 1:
        dist.reduce(local_sum, dst=dst, op=dist.ReduceOp.SUM)
    return local_sum


class FrameAccumulator:
    """frames partitioning with the exchange hidden: frame i is rendered into buffer i % 2 and its
    reduce(SUM) to `dst` is launched asynchronously, so it runs while the next frame renders into the
    other buffer.  Before a buffer is rendered into again its reduce is waited for (with RCCL that is a
    stream dependency, not a host wait) and, on `dst`, the reduced frame is added to `acc`.

        b = fa.acquire(i)        # buffer index for frame i, free to be written
        render(buffers[b])       # enqueue the frame
        fa.submit(b)             # launch its reduce
        ...
        fa.drain()               # all reduces finished, acc complete on dst
    """

    def __init__(self, dist, world, rank, buffers, acc, dst=0, exchange=None, streams=None):
        """exchange: launch the reduce even with one rank (default: only when world > 1); see TileCompositor.
        streams: two torch streams, one per buffer, when the two frames in flight are rendered on streams of their own
        (rt_set_option RT_OPT_WF_LANE): buffer b's reduce and its addition then run on streams[b], into a partial sum per
        buffer (two streams adding into one tensor would race), and drain() folds the partial sums into acc on the
        caller's stream.  The caller renders frame i under torch.cuda.stream(streams[i & 1])."""
        assert len(buffers) == 2
        self.dist, self.world, self.rank, self.buffers, self.acc, self.dst = dist, world, rank, buffers, acc, dst
        self.exchange = world > 1 if exchange is None else bool(exchange)
        self.pending = [None, None]
        self.dirty = [False, False]
        self.host = None
        self.streams = streams
        self.partials = None
        if streams is not None:
            assert len(streams) == 2
            import torch
            self.torch = torch
            self.partials = [torch.zeros_like(acc), torch.zeros_like(acc)] if rank == dst else None

    def _finish(self, b):
        if self.streams is not None:
            with self.torch.cuda.stream(self.streams[b]):
                self._finish_on_current(b, self.partials[b] if self.partials is not None else None)
        else:
            self._finish_on_current(b, self.acc)

    def _finish_on_current(self, b, into):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        if self.dirty[b]:
            if self.rank == self.dst:
                into.add_(self.buffers[b])
            self.dirty[b] = False

    def acquire(self, i):
        b = i & 1
        self._finish(b)
        return b

    def submit(self, b):
        if self.exchange:
            buf = self.buffers[b]
            if buf.is_cuda and self.dist.get_backend() != "nccl":
                # gloo under device tensors (bench.py's one-GPU rehearsal): staged through pinned host memory by hand, in line;
                # gloo's own device-tensor path stalls for seconds at a time off the default stream (tools/rehearse_probe.py)
                if self.host is None:
                    self.host = buf.cpu().pin_memory()
                self.host.copy_(buf)
                self.dist.reduce(self.host, dst=self.dst, op=self.dist.ReduceOp.SUM)
                if self.rank == self.dst:
                    buf.copy_(self.host)
            else:   # RCCL (stream-ordered) or gloo on host tensors: in flight while the next frame renders
                self.pending[b] = self.dist.reduce(buf, dst=self.dst, op=self.dist.ReduceOp.SUM, async_op=True)
        self.dirty[b] = True

    def drain(self):
        self._finish(0)
        self._finish(1)
        if self.streams is not None:
            cur = self.torch.cuda.current_stream()
            for s in self.streams:
                cur.wait_stream(s)
            if self.partials is not None:
                for p in self.partials:
                    self.acc.add_(p)
                    p.zero_()
                for s in self.streams:   # the partial sums are not touched again before this
                    s.wait_stream(cur)
