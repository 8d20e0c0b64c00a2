"""The device SAH builder (bvh_policy 5) against the host builder, scene by scene: same nodes / order, build times.
PTAMD_TRACE=1 prints the phases."""
import sys
import time
sys.path.insert(0, ".")
import numpy as np
from opencl_path_tracer_amd import api, scenes

cases = [("cornell", 64), ("cornell", 4096), ("mesh6k", 512), ("mesh100k", 512), ("mesh100k", 64), ("mesh1M", 512)]
want = sys.argv[1:]
for name, grain in cases:
    if want and name not in want:
        continue
    spec = scenes.cornell_box() if name == "cornell" else scenes.displaced_grid_mesh({"mesh6k": 6000, "mesh100k": 100000, "mesh1M": 1000000}[name])
    print("== %s grain %d" % (name, grain), flush=True)
    ref = api.Scene(32, 32)
    ref.set_option("bvh_device", 0)
    ref.load(spec)
    ref.upload_Triangles()
    dev = api.Scene(32, 32)
    dev.set_option("bvh_policy", 5)
    dev.set_option("sah_grain", grain)
    dev.load(spec)
    dev.upload_Triangles()
    print("   host %.2f ms  device %.2f ms (on device: %d)  nodes %d / %d  depth %d / %d" % (
        ref.stat("bvh_build_ms"), dev.stat("bvh_build_ms"), dev.stat("bvh_on_device"), ref.stat("bvh_nodes"), dev.stat("bvh_nodes"),
        ref.stat("bvh_depth"), dev.stat("bvh_depth")), flush=True)
    a, b = ref.debug_bvh(), dev.debug_bvh()
    same_order = np.array_equal(a[3], b[3])
    an, bn = a[0].view(np.uint32), b[0].view(np.uint32)
    same_nodes = an.shape == bn.shape and np.array_equal(an, bn)
    wa, wb = ref.debug_wide_nodes(), dev.debug_wide_nodes()
    same_wide = len(wa) == len(wb) and wa.tobytes() == wb.tobytes()
    print("   same order %s  same nodes %s  4-wide nodes %d / %d same %s" % (same_order, same_nodes, len(wa), len(wb), same_wide), flush=True)
    if not same_wide and len(wa) == len(wb) and len(wa):
        bad = [i for i in range(len(wa)) if wa[i].tobytes() != wb[i].tobytes()]
        print("   %d wide nodes differ, first %d:\n%s\n%s" % (len(bad), bad[0], wa[bad[0]], wb[bad[0]]), flush=True)
    if not same_nodes and an.shape == bn.shape:
        bad = np.nonzero((an != bn).any(axis=1))[0]
        print("   %d nodes differ, first %d:\n%s\n%s" % (len(bad), bad[0], a[0][bad[0]], b[0][bad[0]]), flush=True)
        print(an[bad[0]], bn[bad[0]])
