"""Throughput sweep of the render kernel variants on one GPU (development tool)."""
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402


def run(W, H, bounces, spp, lds, block, spec, reps=2):
    sc = api.Scene(W, H).load(spec)
    sc.set_option("lds_scene", lds)
    sc.set_option("block", block)
    sc.set_option("timing", 1)
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
    print("%dx%d b%d spp%d lds=%d block=%4d lds_bytes=%6d: %8.1f Msamples/s (wall) %8.1f (kernel)  dbar=%.3f  Mseg/s=%.1f" % (
        W, H, bounces, spp, lds, block, sc.stat("lds_bytes"), samples / dt / 1e6, samples / kms / 1e3, segs / samples, segs / kms / 1e3), flush=True)


if __name__ == "__main__":
    spec = scenes.cornell_box()
    W, H = 1920, 1080
    for lds, block in ((1, 256), (1, 128), (0, 256), (0, 512), (0, 1024), (0, 128)):
        run(W, H, 8, 16, lds, block, spec)
    run(256, 256, 4, 16, 1, 256, spec)
