"""Throughput sweep of the render kernel variants on one GPU (development tool)."""
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402


def run(W, H, bounces, spp, spec, reps=2, count=False, **opts):
    sc = api.Scene(W, H)
    pre = ("bvh_policy", "treelet", "flat_list", "wide_nodes", "wide_lds_entries", "sah_visit_cost")    # options the upload depends on
    for k in pre:
        if k in opts:
            sc.set_option(k, opts[k])
    sc.load(spec)
    for k, v in opts.items():
        if k not in pre:
            sc.set_option(k, v)
    sc.set_option("timing", 1)
    sc.set_option("count_work", 1 if count else 0)
    sc.iterations = bounces
    sc.render(2)
    sc.sync()
    sc.set_option("reset_stats", 1)
    t = time.time()
    for _ in range(reps):
        sc.render(spp)
    sc.sync()
    dt = time.time() - t
    segs, samples, kms = sc.stat("segments"), sc.stat("samples"), sc.stat("kernel_ms")
    extra = ""
    if count:
        nv, tt, wn, wt = sc.stat("node_visits"), sc.stat("tri_tests"), sc.stat("wave_node_steps"), sc.stat("wave_tri_steps")
        wsegs = segs / 64.0
        print("   executions for <= 8 lanes: node body %.0f%%, triangle body %.0f%%, its exact part %.0f%% (x%.1f per wave-segment), shade %.0f%%" % (
            100 * sc.stat("low_node") / max(wn, 1), 100 * sc.stat("low_tri") / max(wt, 1), 100 * sc.stat("low_exact") / max(sc.stat("low_exact_all"), 1),
            sc.stat("low_exact_all") / wsegs, 100 * sc.stat("low_shade") / max(sc.stat("wave_shade_steps"), 1)))
        print("   per-tile lane balance (lane segments / 64 x busiest lane): %.1f%%" % (100 * segs / max(sc.stat("tile_lane_steps"), 1)))
        extra = "  nodes/seg=%.2f tris/seg=%.2f | per wave-segment: node body x%.1f (util %.0f%%), tri body x%.1f (util %.0f%%), shade x%.2f, trips x%.2f, rounds x%.2f" % (
            nv / segs, tt / segs, wn / wsegs, 100 * nv / (64 * wn), wt / wsegs, 100 * tt / (64 * wt),
            sc.stat("wave_shade_steps") / wsegs, sc.stat("wave_trips") / wsegs, sc.stat("wave_rounds") / wsegs)
    print("%dx%d b%d spp%d %-34s nodes=%d mode=%d treelet=%d lds_bytes=%6d: %8.1f Msamples/s (kernel %8.1f)  dbar=%.3f  Mseg/s=%.1f%s" % (
        W, H, bounces, spp, str(opts), sc.stat("bvh_nodes"), sc.stat("node_mode"), sc.stat("treelet_nodes"), sc.stat("lds_bytes"), samples / dt / 1e6, samples / kms / 1e3, segs / samples, segs / kms / 1e3, extra), flush=True)


def checksum(W, H, bounces, spp, spec, **opts):
    import numpy as np
    sc = api.Scene(W, H)
    sc.load(spec)
    for k, v in opts.items():
        sc.set_option(k, v)
    sc.iterations = bounces
    sc.render(spp)
    sc.sync()
    return float(np.asarray(sc.read_colors(), dtype=np.float64).sum()), int(np.asarray(sc.read_rnds(), dtype=np.int64).sum())


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", action="store_true", help="also the counting kernel instance (lane utilisation)")
    ap.add_argument("--what", default="cb,mesh100k,mesh1m")
    args = ap.parse_args()
    what = args.what.split(",")
    if "cb" in what:
        cb = scenes.cornell_box()
        run(1920, 1080, 8, 64, cb, reps=3)
        run(1920, 1080, 8, 64, cb, reps=3, lds_scene=0)
        run(1920, 1080, 8, 16, cb, reps=2, variant=1)
        run(1920, 1080, 8, 16, cb, reps=2, variant=1, lds_scene=0)
        if args.count:
            run(1920, 1080, 8, 16, cb, reps=1, count=True)
    if "c1c4" in what:
        cb = scenes.cornell_box()
        run(256, 256, 4, 64, cb, reps=4)
        run(3840, 2160, 8, 16, cb, reps=2)
    for name, n, b, spp in (("mesh100k", 100000, 8, 16), ("mesh1m", 1000000, 16, 8)):
        if name not in what:
            continue
        m = scenes.displaced_grid_mesh(n)
        run(1920, 1080, b, spp, m, reps=2)
        run(1920, 1080, b, spp, m, reps=2, lds_scene=0)
        run(1920, 1080, b, spp, m, reps=2, treelet=256)
        run(1920, 1080, b, spp, m, reps=2, variant=1)
        run(1920, 1080, b, spp, m, reps=2, variant=1, lds_scene=0)
        if args.count:
            run(1920, 1080, b, spp, m, reps=1, count=True)
            run(1920, 1080, b, spp, m, reps=1, count=True, lds_scene=0)
