"""Strong-scaling probe on ONE GPU: time the tile set of rank r of an N-rank job (the ranks of a
real job run concurrently on N GPUs, so the job time is the max over ranks + the exchange)."""
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

spec = scenes.cornell_box()
W, H, B, SPP, STEPS = 1920, 1080, 8, 64, 4


def t_rank(world, rank, rb=8, **opts):
    sc = api.Scene(W, H, rank=rank, world=world, rows_per_block=rb).load(spec)
    for k, v in opts.items():
        sc.set_option(k, v)
    sc.iterations = B
    sc.render(SPP)
    sc.sync()
    t = time.time()
    for _ in range(STEPS):
        sc.render(SPP)
    sc.sync()
    return time.time() - t


base = t_rank(1, 0)
print("1 rank : %.3f s  (%.1f Msamples/s)" % (base, W * H * SPP * STEPS / base / 1e6), flush=True)
b1 = t_rank(1, 0, chunk_spp=4)
print("1 rank chunk 4: %.3f s  (%.1f Msamples/s)" % (b1, W * H * SPP * STEPS / b1 / 1e6), flush=True)
for world in (2, 4, 8):
    for opts in ({}, {"chunk_spp": 4}, {"chunk_spp": 2}):
        ts = [t_rank(world, r, **opts) for r in sorted(set([0, world // 2, world - 1]))]
        worst = max(ts)
        print("%d ranks %-18s: rank times %s -> efficiency %.1f%% vs tile-map N=1 (render only)" % (world, opts, ["%.3f" % x for x in ts], 100 * base / (world * worst)), flush=True)
