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


# ------------------------------------------------------------------------------------------------
# Self-validation of an N-rank run (bench.py --gpus N): the frame is bit-identical for any N by construction
# (seeds and pixel ids are those of the global frame), so a run can prove it on the spot: rank 0 renders a band of
# the frame as a ONE-rank context and compares it bit for bit with the same rows of the gathered frame.
def band_context(height, band_rows=24, at=0.3):
    """A band of `band_rows` consecutive rows of the global frame as the tiling arguments of a one-rank context:
    Scene(W, H, rank=k, world=n, rows_per_block=band_rows) owns exactly rows [k * band_rows, (k + 1) * band_rows)
    when n = ceil(H / band_rows) (every block index below n belongs to a different "rank").  `at` places the band
    (fraction of the height; 0.3 is where the Cornell box's spheres are -- row 0 is the bottom of the view)."""
    band_rows = max(1, min(int(band_rows), int(height)))
    n = (height + band_rows - 1) // band_rows
    k = min(n - 1, max(0, int(at * height) // band_rows))
    return {"rank": k, "world": n, "rows_per_block": band_rows}


def band_pixel_ids(width, height, band):
    """Global pixel ids of the band's pixels in its context's local order (== pt_local_pixel_ids of that context)."""
    r0 = band["rank"] * band["rows_per_block"]
    rows = np.arange(r0, min(r0 + band["rows_per_block"], height))
    return (rows[:, None].astype(np.int64) * width + np.arange(width)[None, :]).reshape(-1)


def frame_matches_band(frame, band_colors, ids):
    """Bit-for-bit comparison of the gathered frame's rows with the one-rank band render (RGB; the pad lane is not
    part of the contract).  frame: (W*H, 4) float32, band_colors: (len(ids), 4) float32."""
    a = np.ascontiguousarray(np.asarray(frame, dtype=np.float32)[ids][:, :3]).view(np.uint32)
    b = np.ascontiguousarray(np.asarray(band_colors, dtype=np.float32)[:, :3]).view(np.uint32)
    return bool(a.shape == b.shape and np.array_equal(a, b))


class Watchdog:
    """Bounds a blocking phase of a multi-process run (rendezvous, ncclCommInitRank, a collective + synchronize): if
    the block has not finished after `seconds`, the process says which phase hung and exits non-zero -- a missing rank
    then fails the job instead of hanging it.  (os._exit: the stuck call cannot be interrupted, and no replacement
    process is started from one that has touched the GPU.)"""

    def __init__(self, seconds, what, rank=0, code=3):
        self.seconds, self.what, self.rank, self.code = float(seconds), what, rank, code
        self._t = None

    def _fire(self):
        import os
        import sys
        print("[watchdog] rank %d: '%s' did not finish within %.0f s -- a rank is missing or stuck; exiting with code %d"
              % (self.rank, self.what, self.seconds, self.code), file=sys.stderr, flush=True)
        os._exit(self.code)

    def __enter__(self):
        import threading
        if self.seconds > 0:
            self._t = threading.Timer(self.seconds, self._fire)
            self._t.daemon = True
            self._t.start()
        return self

    def __exit__(self, *exc):
        if self._t is not None:
            self._t.cancel()
        return False
