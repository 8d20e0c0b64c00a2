"""Phase-switch thresholds of the while-while rounds (options node_min_lanes / leaf_min_lanes) and the suspension threshold
(suspend_lanes) on the BASELINE scenes.
usage: python tools/sweep_phase.py <scene[:spp],...> [--nm 0,4,8] [--lm 0] [--sl -1] [--reps 3]     scene: cornell mesh100k mesh1m wf wf100k"""
import argparse
import itertools
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scenes")
ap.add_argument("--nm", default="0,4,8")
ap.add_argument("--lm", default="0")
ap.add_argument("--sl", default="-1")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
CASES = {"cornell": (scenes.cornell_box, 8, 64, 0), "mesh100k": (lambda: scenes.displaced_grid_mesh(100000), 8, 16, 0),
         "mesh1m": (lambda: scenes.displaced_grid_mesh(1000000), 16, 8, 0), "wf": (scenes.cornell_box, 8, 8, 1),
         "wf100k": (lambda: scenes.displaced_grid_mesh(100000), 8, 4, 1)}
ints = lambda s: [int(x) for x in s.split(",")]
for item in args.scenes.split(","):
    name, _, spp_s = item.partition(":")          # scene[:samples per launch]
    make, bounces, spp, variant = CASES[name]
    spp = int(spp_s) if spp_s else spp
    sc = api.Scene(1920, 1080).load(make())
    sc.iterations = bounces
    sc.set_option("variant", variant)
    sc.set_option("timing", 1)
    sc.render(spp)
    sc.sync()
    for sl, nm, lm in itertools.product(ints(args.sl), ints(args.nm), ints(args.lm)):
        sc.set_option("suspend_lanes", sl)
        sc.set_option("node_min_lanes", nm)
        sc.set_option("leaf_min_lanes", lm)
        sc.render(spp)
        sc.sync()
        sc.set_option("reset_stats", 1)
        t = time.time()
        for _ in range(args.reps):
            sc.render(spp)
        sc.sync()
        dt = time.time() - t
        wall = 1920 * 1080 * spp * args.reps / dt / 1e6
        print("%-9s suspend %2d node_min %2d leaf_min %2d: %8.1f Msamples/s" % (name, sl, nm, lm, wall if variant else sc.stat("samples") / sc.stat("kernel_ms") / 1e3), flush=True)
    sc.close()
