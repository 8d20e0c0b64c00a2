"""Traversal-only throughput: realistic secondary rays (the rays buffer after k bounces of the
split API) pushed through k_debug_closest_hit repeatedly."""
import sys
sys.path.insert(0, ".")
import numpy as np
from opencl_path_tracer_amd import api, scenes

spec = scenes.cornell_box()
W, H = 1920, 1080
sc = api.Scene(W, H).load(spec)
for bounces in (0, 1, 3):
    sc.iterations = bounces
    sc.current_sample = 0
    sc.seed_default()
    sc.generate_rays()
    if bounces:
        sc.trace_rays()
    rays = sc.read_rays()
    ok = np.isfinite(rays["D"][:, 0])
    rays = rays[ok]
    sc.set_option("reset_stats", 1)
    sc.set_option("debug_repeat", 5)
    t, tri = sc.debug_closest_hit(rays)
    ms = sc.stat("kernel_ms") / 5
    print("rays after %d bounce(s): n=%d  hit %.1f%%  %.3f ms -> %.2f Grays/s" % (bounces, rays.shape[0], 100 * (tri >= 0).mean(), ms, rays.shape[0] / ms / 1e6), flush=True)
    # shuffled (incoherent order)
    perm = np.random.RandomState(1).permutation(rays.shape[0])
    sc.set_option("reset_stats", 1)
    t, tri = sc.debug_closest_hit(rays[perm])
    ms = sc.stat("kernel_ms") / 5
    print("   same rays, shuffled order:                      %.3f ms -> %.2f Grays/s" % (ms, rays.shape[0] / ms / 1e6), flush=True)
