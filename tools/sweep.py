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
        extra = "  nodes/seg=%.2f tris/seg=%.2f" % (sc.stat("node_visits") / segs, sc.stat("tri_tests") / segs)
    print("%dx%d b%d spp%d %-45s nodes=%d lds_bytes=%6d: %8.1f Msamples/s (kernel %8.1f)  dbar=%.3f  Mseg/s=%.1f%s" % (
        W, H, bounces, spp, str(opts), sc.stat("bvh_nodes"), sc.stat("lds_bytes"), samples / dt / 1e6, samples / kms / 1e3, segs / samples, segs / kms / 1e3, extra), flush=True)


if __name__ == "__main__":
    spec = scenes.cornell_box()
    W, H = 1920, 1080
    run(W, H, 8, 16, spec, count=True, bvh_policy=1)
    for mw in (1, 4, 5):
        run(W, H, 8, 16, spec, min_waves=mw, bvh_policy=1)
    run(W, H, 8, 16, spec, min_waves=4, bvh_policy=2)
    run(W, H, 8, 16, spec, block=64, bvh_policy=1)
    run(W, H, 8, 16, spec, lds_scene=1, block=1024, bvh_policy=2)
