"""Host SAH builder vs device LBVH builder: build time and render throughput."""
import sys
import time
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes

W, H = 1920, 1080
for name, spec, bounces in (("cornell", scenes.cornell_box(), 8), ("mesh100k", scenes.displaced_grid_mesh(100000), 8), ("mesh1M", scenes.displaced_grid_mesh(1000000), 16)):
    for policy, cluster in ((0, 0), (4, 0), (4, 8), (4, 16), (4, 32), (4, 64), (4, 256)):
        sc = api.Scene(W, H)
        sc.set_option("bvh_policy", policy)
        sc.set_option("lbvh_cluster", cluster)
        sc.load(spec)
        sc.upload_Triangles()       # second build: excludes first-touch allocations
        bms = sc.stat("bvh_build_ms")
        sc.set_option("timing", 1)
        sc.iterations = bounces
        sc.render(2)
        sc.sync()
        sc.set_option("reset_stats", 1)
        sc.render(8)
        sc.sync()
        kms = sc.stat("kernel_ms")
        print("%-9s %-24s tris=%8d nodes=%8d depth=%2d build %9.2f ms   render %7.1f Msamples/s" % (
            name, ("device LBVH, SAH top over clusters of %d" % cluster if cluster else "device LBVH") if sc.stat("bvh_on_device") else "host SAH", spec.ntris, sc.stat("bvh_nodes"), sc.stat("bvh_depth"), bms, W * H * 8 / kms / 1e3), flush=True)
