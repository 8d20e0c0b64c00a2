"""Megakernel schedules on the BASELINE scenes at 1920x1080: restart + tail suspension (schedule 1) against the same with lanes
moving on to the wave's next work item (schedule 2), over samples per work item (chunk_spp; -1 automatic, 0 whole tiles) and the
number of lanes that move together (migrate_lanes).
usage: python tools/sweep_sched.py <scene[:spp],...> [--sched=1,2] [--chunk=-1] [--ml=1] [--reps 2] [--opt key=value ...]     scene: cornell mesh100k mesh1m"""
import argparse
import itertools
import sys

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scenes")
ap.add_argument("--sched", default="1,2")
ap.add_argument("--chunk", default="-1")
ap.add_argument("--ml", default="1")
ap.add_argument("--taper", default="-1", help="chunk_taper values: shortest pass at the end of the launch (0 off)")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--opt", action="append", default=[], help="key=value set before the scene is loaded (lds_block=512, lds_scene=0 ...)")
args = ap.parse_args()
CASES = {"cornell": (scenes.cornell_box, 8, 64), "mesh100k": (lambda: scenes.displaced_grid_mesh(100000), 8, 16),
         "mesh1m": (lambda: scenes.displaced_grid_mesh(1000000), 16, 8)}
ints = lambda s: [int(x) for x in s.split(",")]
for item in args.scenes.split(","):
    name, _, spp_s = item.partition(":")
    make, bounces, spp = CASES[name]
    spp = int(spp_s) if spp_s else spp
    sc = api.Scene(1920, 1080)
    for kv in args.opt:
        sc.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    sc.load(make())
    sc.iterations = bounces
    sc.set_option("timing", 1)
    sc.render(spp)
    sc.sync()
    for sched, chunk, ml, taper in itertools.product(ints(args.sched), ints(args.chunk), ints(args.ml), ints(args.taper)):
        if sched != 2 and ml != ints(args.ml)[0]:
            continue
        sc.set_option("schedule", sched)
        sc.set_option("chunk_spp", chunk)
        sc.set_option("migrate_lanes", ml)
        sc.set_option("chunk_taper", taper)
        sc.render(spp)
        sc.sync()
        sc.set_option("reset_stats", 1)
        for _ in range(args.reps):
            sc.render(spp)
        sc.sync()
        print("%-9s spp %3d schedule %d chunk_spp %3d migrate_lanes %2d chunk_taper %2d: %8.1f Msamples/s" % (name, spp, sched, chunk, ml, taper, sc.stat("samples") / sc.stat("kernel_ms") / 1e3), flush=True)
    sc.close()
