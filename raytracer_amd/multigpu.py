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

BAND_ROWS = 64  # multiple of the 8-row tiles; a band of a 1080p x 16 spp frame is 2 M samples, enough for one wavefront chunk


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


def composite_tiles(dist, local_fb, width, height, rank, world, band_rows: int = BAND_ROWS):
    """Composite of one frame rendered as interleaved row bands: every rank packs the bands it owns into one
    contiguous tensor, one all_gather moves them (1/world of a frame per rank instead of a whole frame), and
    each rank unpacks every owner's bands into place.  `local_fb` is a torch tensor [H, W, 4]; rows this rank
    does not own are ignored on input and filled on output.  Ranks that own one band fewer (or a short last
    band) pad their packet with zeros so that the collective stays a plain equal-sized all_gather."""
    import torch

    if world == 1:
        return local_fb
    nb = (height + band_rows - 1) // band_rows
    per_rank = (nb + world - 1) // world                      # bands in a packet
    def owned(r):
        return list(range(r, nb, world))
    packet = torch.zeros((per_rank * band_rows, width, 4), dtype=local_fb.dtype, device=local_fb.device)
    for k, b in enumerate(owned(rank)):
        y0, y1 = b * band_rows, min(height, (b + 1) * band_rows)
        packet[k * band_rows:k * band_rows + (y1 - y0)] = local_fb[y0:y1]
    parts = [torch.empty_like(packet) for _ in range(world)]
    dist.all_gather(parts, packet)
    out = torch.empty_like(local_fb)
    for r in range(world):
        for k, b in enumerate(owned(r)):
            y0, y1 = b * band_rows, min(height, (b + 1) * band_rows)
            out[y0:y1] = parts[r][k * band_rows:k * band_rows + (y1 - y0)]
    return out


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
