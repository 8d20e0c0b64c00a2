"""Host SAH builder: time and identity of the tree against the number of build threads (development tool)."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
from opencl_path_tracer_amd import api, scenes

for n in (100000, 1000000):
    spec = scenes.displaced_grid_mesh(n)
    ref = None
    for th in (1, 2, 4, 8, 16, 0):
        sc = api.Scene(16, 16, device=None)
        sc.set_option("build_threads", th)
        t = time.time()
        sc.load(spec)
        dt = time.time() - t
        nodes, tris, meta, orig = sc.debug_bvh()
        same = "" if ref is None else "  same tree: %s" % (np.array_equal(nodes.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(orig, ref[1]))
        if ref is None:
            ref = (nodes, orig)
        print("%d triangles, build_threads %2d: upload_Triangles %.1f ms (whole load %.2f s)%s" % (n, th, sc.stat("bvh_build_ms"), dt, same), flush=True)
