"""Host SAH builder vs device builders (the same SAH tree built on the device; Morton radix tree / PLOC, with and without the
SAH top over clusters): build time and render throughput."""
import sys
import time
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes

W, H = 1920, 1080
which = sys.argv[1:] or ["cornell", "mesh100k", "mesh1M"]
cases = ((0, 0, 0), (5, 0, 0), (4, 0, 0), (4, 0, 64), (4, 8, 0), (4, 16, 0), (4, 32, 0), (4, 16, 64), (4, 16, 8))
for name, spec, bounces in (("cornell", scenes.cornell_box(), 8), ("mesh100k", scenes.displaced_grid_mesh(100000), 8), ("mesh1M", scenes.displaced_grid_mesh(1000000), 16)):
    if name not in which:
        continue
    for policy, ploc, cluster in cases:
        sc = api.Scene(W, H)
        sc.set_option("bvh_policy", policy)
        sc.set_option("bvh_device", 1 if policy >= 4 else 0)
        sc.set_option("lbvh_ploc", ploc)
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
        what = "host SAH"
        if policy == 5:
            what = "device SAH (the host builder's tree)" if sc.stat("bvh_on_device") else "device SAH -> handed back to the host"
        elif sc.stat("bvh_on_device"):
            what = ("device PLOC r=%d" % ploc if ploc else "device Morton radix tree") + (", SAH top over clusters of %d" % cluster if cluster else "")
        print("%-9s %-52s tris=%8d nodes=%8d depth=%2d build %9.2f ms   render %7.1f Msamples/s" % (
            name, what, spec.ntris, sc.stat("bvh_nodes"), sc.stat("bvh_depth"), bms, W * H * 8 / kms / 1e3), flush=True)
