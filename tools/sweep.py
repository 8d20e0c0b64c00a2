"""Throughput sweep of the render kernel variants on one GPU (development tool)."""
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402


def run(W, H, bounces, spp, spec, reps=2, count=False, **opts):
    sc = api.Scene(W, H)
    if "bvh_policy" in opts:
        sc.set_option("bvh_policy", opts["bvh_policy"])
    sc.load(spec)
    for k, v in opts.items():
        if k != "bvh_policy":
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
        print("   per-tile lane balance (lane segments / 64 x busiest lane): %.1f%%" % (100 * segs / max(sc.stat("tile_lane_steps"), 1)))
        extra = "  nodes/seg=%.2f tris/seg=%.2f | per wave-segment: node body x%.1f (util %.0f%%), tri body x%.1f (util %.0f%%)" % (
            nv / segs, tt / segs, wn / wsegs, 100 * nv / (64 * wn), wt / wsegs, 100 * tt / (64 * wt))
    print("%dx%d b%d spp%d %-45s nodes=%d lds_bytes=%6d: %8.1f Msamples/s (kernel %8.1f)  dbar=%.3f  Mseg/s=%.1f%s" % (
        W, H, bounces, spp, str(opts), sc.stat("bvh_nodes"), sc.stat("lds_bytes"), samples / dt / 1e6, samples / kms / 1e3, segs / samples, segs / kms / 1e3, extra), flush=True)


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
    cb = scenes.cornell_box()
    run(256, 256, 4, 64, cb, reps=4)
    m100 = scenes.displaced_grid_mesh(100000)
    run(1920, 1080, 8, 16, m100, reps=2)
    run(1920, 1080, 8, 16, m100, reps=2, variant=1)
    run(3840, 2160, 8, 16, cb, reps=2)
    m1m = scenes.displaced_grid_mesh(1000000)
    run(1920, 1080, 16, 8, m1m, reps=2)
    run(1920, 1080, 16, 8, m1m, reps=2, variant=1)
