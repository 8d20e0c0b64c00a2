"""Strong-scaling probe on ONE GPU: time the tile set of rank r of an N-rank job (the ranks of a
real job run concurrently on N GPUs, so the job time is the max over ranks + the exchange).
  python tools/scaling_probe.py [W H] [matrix]"""
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

spec = scenes.cornell_box()
W, H, B, SPP, STEPS = 1920, 1080, 8, 64, 4
args = [a for a in sys.argv[1:] if a != "matrix"]
if len(args) >= 2:
    W, H = int(args[0]), int(args[1])


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
print("%dx%d 1 rank : %.4f s  (%.1f Msamples/s)" % (W, H, base, W * H * SPP * STEPS / base / 1e6), flush=True)
if "matrix" in sys.argv:
    cases = [dict(schedule=0), dict(schedule=1), dict(schedule=1, chunk_spp=0), dict(schedule=1, chunk_spp=16), dict(schedule=0, chunk_spp=0), dict(schedule=0, chunk_spp=8)]
    worlds = (2, 4, 8)
else:
    cases = [{}]
    worlds = (2, 4, 8)
for world in worlds:
    for opts in cases:
        ts = [t_rank(world, r, **opts) for r in sorted(set([0, world // 2, world - 1]))]
        worst = max(ts)
        print("%d ranks %-52s: rank times %s -> efficiency %.1f%% vs N=1 (render only)" % (world, opts, ["%.4f" % x for x in ts], 100 * base / (world * worst)), flush=True)
