"""Traversal-only: nodes from L1/L2 vs nodes staged in LDS (persistent 512-thread blocks)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from opencl_path_tracer_amd import api, scenes

spec = scenes.cornell_box()
W, H = 1920, 1080
sc = api.Scene(W, H).load(spec)
sc.iterations = 3
sc.generate_rays()
sc.trace_rays()
rays = sc.read_rays()
rays = rays[np.isfinite(rays["D"][:, 0])]
ref = None
for name, pad in (("one ray per thread, nodes global (8 waves/SIMD)", 0), ("persistent 512-thr blocks, nodes global", 1), ("persistent 512-thr blocks, nodes in LDS", 2)):
    sc.set_option("reset_stats", 1)
    sc.set_option("debug_repeat", 5)
    sc.set_option("debug_lds_pad", pad)
    t, tri = sc.debug_closest_hit(rays)
    ms = sc.stat("kernel_ms") / 5
    print("%-50s: %.3f ms -> %.2f Grays/s" % (name, ms, rays.shape[0] / ms / 1e6), flush=True)
