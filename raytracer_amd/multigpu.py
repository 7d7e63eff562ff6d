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
    bit-identical to the single-GPU frame."""

    def __init__(self, torch, dist, width, height, rank, world, device=None, band_rows: int = BAND_ROWS, dtype=None):
        self.dist, self.rank, self.world, self.band_rows = dist, rank, world, band_rows
        self.width, self.height = width, height
        self.nb, self.per_rank, rows = band_layout(height, world, band_rows)
        dtype = dtype or torch.float32
        self.padded = torch.zeros((rows, width, 4), dtype=dtype, device=device)
        self.frame = self.padded[:height]
        self.view = self.padded.view(self.per_rank, world, band_rows, width, 4)
        self.packet = torch.empty((self.per_rank, band_rows, width, 4), dtype=dtype, device=device)
        self.gathered = torch.empty((world, self.per_rank, band_rows, width, 4), dtype=dtype, device=device)

    def composite(self):
        """All ranks call this after rendering their bands into `frame`; on return `frame` is complete."""
        if self.world == 1:
            return self.frame
        self.packet.copy_(self.view[:, self.rank])
        if self.dist.get_backend() == "nccl":   # RCCL: straight into the contiguous [world, ...] tensor
            self.dist.all_gather_into_tensor(self.gathered, self.packet)
        else:
            self.dist.all_gather(list(self.gathered.unbind(0)), self.packet)
        self.view.copy_(self.gathered.permute(1, 0, 2, 3, 4))
        return self.frame


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

    def __init__(self, dist, world, rank, buffers, acc, dst=0):
        assert len(buffers) == 2
        self.dist, self.world, self.rank, self.buffers, self.acc, self.dst = dist, world, rank, buffers, acc, dst
        self.pending = [None, None]
        self.dirty = [False, False]

    def _finish(self, b):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        if self.dirty[b]:
            if self.rank == self.dst:
                self.acc.add_(self.buffers[b])
            self.dirty[b] = False

    def acquire(self, i):
        b = i & 1
        self._finish(b)
        return b

    def submit(self, b):
        if self.world > 1:
            self.pending[b] = self.dist.reduce(self.buffers[b], dst=self.dst, op=self.dist.ReduceOp.SUM, async_op=True)
        self.dirty[b] = True

    def drain(self):
        self._finish(0)
        self._finish(1)
