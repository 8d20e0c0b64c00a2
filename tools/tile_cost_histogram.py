"""The bound on strong scaling at one tile per resident wave: per 8x8 tile, the time its wave spends on
its 64 samples (shader-clock cycles / 64 between fetching the work item and finishing it; a pixel's samples are one sequential
LCG stream, so a tile is one wave's work from start to end), measured by the counting instance of the kernel -- for the tile
set of rank 0 of an 8-rank 1080p job (4,080 tiles for 4,096 resident waves, whole tiles, lockstep: the shape bench.py --gpus 8
launches) and for the whole frame on one GPU.  A launch with one tile per wave ends with its dearest tile, so its efficiency
is capped at mean / max of this distribution whatever the schedule.
  python tools/tile_cost_histogram.py [W H spp]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

W, H, SPP = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 64)
for world in (8, 1):
    sc = api.Scene(W, H, rank=0, world=world, rows_per_block=8).load(scenes.cornell_box())
    sc.iterations = 8
    sc.set_option("count_work", 1)
    if world > 1:
        sc.set_option("schedule", 0)
        sc.set_option("chunk_spp", 0)          # whole tiles: one work item per tile
    sc.render(SPP)
    sc.sync()
    sc.current_sample = 0
    sc.seed_default()
    sc.render(SPP)
    sc.sync()
    c = sc.debug_tile_cost().astype(np.float64) * 64.0 / 2.4e9 * 1e3       # ms at 2.4 GHz
    print("rank 0 of %d, %dx%d, %d spp, %d tiles (lds_block %d, schedule %d): wave time per tile: mean %.2f ms  median %.2f  p90 %.2f  p99 %.2f  max %.2f  -> mean / max = %.3f, mean / p99 = %.3f" % (
        world, W, H, SPP, c.size, sc.stat("lds_bytes") > 70000 and 768 or 512, 0 if world > 1 else 1, c.mean(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max(), c.mean() / c.max(), c.mean() / np.percentile(c, 99)))
    hist, edges = np.histogram(c, bins=14)
    for h, a, b in zip(hist, edges[:-1], edges[1:]):
        print("   %7.2f - %7.2f ms: %6d tiles %s" % (a, b, h, "#" * int(60 * h / hist.max())))
