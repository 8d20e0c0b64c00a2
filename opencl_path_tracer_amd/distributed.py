"""Multi-GPU tiling of the framebuffer -- host-side plumbing over torch.distributed.

The path shards embarrassingly: all mutable state is per pixel (prog.cl:306,313,319,330,379) and
the scene is read-only, so rank r owns the rows y with (y // rows_per_block) % world == r of the
GLOBAL frame and renders them with no communication.  Seeds and pixel ids are those of the global
frame, hence the assembled image is bit-identical for any number of ranks.  The only exchange is
one all-gather of the rank-major radiance slabs (RCCL over xGMI when the backend is "nccl")
followed by a de-interleave; it is a gather, not an arithmetic reduce, because tiles are disjoint.
"""
import numpy as np
import torch
import torch.distributed as dist


class TileMap:
    """Which global pixels each rank owns, in the rank's local (buffer) order."""

    def __init__(self, width, height, world, rows_per_block=8):
        self.width, self.height, self.world, self.rows_per_block = width, height, world, rows_per_block
        rows = np.arange(height)
        owner = (rows // rows_per_block) % world
        self._rows = [rows[owner == r] for r in range(world)]
        self.counts = [int(r.size) * width for r in self._rows]
        self.max_count = max(self.counts)

    def count(self, rank):
        return self.counts[rank]

    def ids(self, rank):
        """Global pixel id of every local pixel of `rank` (same order as pt_local_pixel_ids)."""
        r = self._rows[rank]
        return (r[:, None].astype(np.int64) * self.width + np.arange(self.width)[None, :]).reshape(-1)

    def gather_index(self, device):
        """For the rank-major, max_count-padded gather buffer: destination row in a frame of
        width*height (+1 trash row that absorbs the padding)."""
        idx = np.full(self.world * self.max_count, self.width * self.height, dtype=np.int64)
        for r in range(self.world):
            idx[r * self.max_count: r * self.max_count + self.counts[r]] = self.ids(r)
        return torch.from_numpy(idx).to(device)


def exchange_frame(slab, tmap, index=None, gathered=None, frame=None, group=None):
    """All-gather every rank's (max_count, C) slab and de-interleave into the full frame.
    Returns a (width*height, C) tensor on every rank."""
    world = tmap.world
    C = slab.shape[1]
    if world == 1:
        return slab[: tmap.width * tmap.height]
    if gathered is None:
        gathered = torch.empty((world * tmap.max_count, C), dtype=slab.dtype, device=slab.device)
    if frame is None:
        frame = torch.empty((tmap.width * tmap.height + 1, C), dtype=slab.dtype, device=slab.device)
    if index is None:
        index = tmap.gather_index(slab.device)
    # one collective, chosen by the backend and never switched at run time: a RuntimeError of the collective
    # is a real error and propagates (a rank-local fallback would post a mismatched collective)
    if dist.get_backend(group) == "gloo":
        parts = list(gathered.view(world, tmap.max_count, C).unbind(0))
        dist.all_gather(parts, slab, group=group)          # writes into the views of `gathered`
    else:
        dist.all_gather_into_tensor(gathered, slab, group=group)
    frame.index_copy_(0, index, gathered)
    return frame[: tmap.width * tmap.height]
