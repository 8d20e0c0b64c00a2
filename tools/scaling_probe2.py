"""Strong-scaling probe on ONE GPU, by kernel instance and schedule: times the tile sets of ranks 0, N/2, N-1 of an N-rank
job (render only; the ranks of a real job run concurrently, so the job time is the slowest rank's) for the two LDS-tree
instances of k_render (lds_block 512 = 4 waves per SIMD at 128 VGPRs, 768 = 6 waves at 80 VGPRs), the three schedules and
several pass lengths.  Prints the projected whole-job rate and the efficiency against the best N = 1 time.
  python tools/scaling_probe2.py [W H] [full]"""
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

spec = scenes.cornell_box()
W, H, B, SPP, STEPS = 1920, 1080, 8, 64, 3
args = [a for a in sys.argv[1:] if a != "full"]
if len(args) >= 2:
    W, H = int(args[0]), int(args[1])


def t_rank(world, rank, **opts):
    sc = api.Scene(W, H, rank=rank, world=world, rows_per_block=8).load(spec)
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


best1 = None
for world in (1, 2, 4, 8):
    cases = [dict(lds_block=b) for b in (512, 768)]
    if world > 1 or "full" in sys.argv:
        cases += [dict(lds_block=b, schedule=s, chunk_spp=c) for b in (512, 768) for s, c in ((0, 0), (0, 8), (1, 16), (1, 32), (2, 2), (2, 4), (2, 8), (2, 16))]
    for opts in cases:
        ts = [t_rank(world, r, **opts) for r in sorted(set([0, world // 2, world - 1]))]
        worst = max(ts)
        if world == 1:
            best1 = worst if best1 is None else min(best1, worst)
        print("%d ranks %-48s: slowest rank %.4f s -> %8.1f Msamples/s whole job, efficiency %5.1f%% vs best N=1" % (
            world, opts, worst, W * H * SPP * STEPS / worst / 1e6, 100 * best1 / (world * worst)), flush=True)
