"""Traversal-only throughput versus occupancy (LDS padding limits blocks per CU)."""
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
stack = 16 * 4 * 256
for blocks_per_cu in (10, 8, 6, 4, 3, 2, 1):
    pad = max(0, 160 * 1024 // blocks_per_cu - stack - 512) if blocks_per_cu < 10 else 0
    sc.set_option("reset_stats", 1)
    sc.set_option("debug_repeat", 5)
    sc.set_option("debug_lds_pad", pad)
    t, tri = sc.debug_closest_hit(rays)
    ms = sc.stat("kernel_ms") / 5
    print("~%2d blocks/CU (%2d waves/CU, lds pad %6d): %.3f ms -> %.2f Grays/s" % (blocks_per_cu, blocks_per_cu * 4, pad, ms, rays.shape[0] / ms / 1e6), flush=True)
