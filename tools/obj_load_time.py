"""Time the pieces of BASELINE config 5's scene path as the reference words it (main.cpp:552-617, 618-630): write the
1M-triangle OBJ+MTL, pt_add_obj (parse + transform + add_Triangle + end_Obj), pt_upload_triangles (BVH build + packing +
copy).  Works without a GPU (a host-only context authors scenes and builds the tree; the upload then fails at the copy).
usage: python tools/obj_load_time.py [ntris]"""
import os
import sys
import tempfile
import time

import torch  # noqa: F401  (the HIP runtime)

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = tempfile.mkdtemp(prefix="ptobj_")
t = time.time()
pos, scale, pitch, yaw = (40.0, -15.0, 25.0), (2.0, 2.0, 2.0), 10.0, 30.0
path, local, faces, band = scenes.write_grid_mesh_obj(n, d, pos, scale, pitch, yaw)
print("wrote %s: %.1f MB, %d vertices, %d faces in %.1f s (python)" % (path, os.path.getsize(path) / 1e6, local.shape[0], faces.shape[0], time.time() - t))
dev = 0 if torch.cuda.is_available() else None
sc = api.Scene(1920, 1080, device=dev)
for m in scenes.BUILTIN_MATERIALS:
    sc.add_Material(*m)
wv, wm = scenes.cornell_walls()
sc.add_Triangles(api.triangles_from_vertices(wv, wm))
sc.end_Obj()
for rep in range(3):
    s2 = api.Scene(64, 64, device=dev)
    for m in scenes.BUILTIN_MATERIALS:
        s2.add_Material(*m)
    t = time.time()
    s2.add_Obj(path, pos, scale, pitch, yaw)
    dt = time.time() - t
    print("pt_add_obj: %.1f ms (%.1f MB/s, %.2f Mtriangles/s)" % (dt * 1e3, os.path.getsize(path) / 1e6 / dt, faces.shape[0] / 1e6 / dt))
    del s2
t = time.time()
sc.add_Obj(path, pos, scale, pitch, yaw)
print("pt_add_obj (scene with walls): %.1f ms" % ((time.time() - t) * 1e3))
t = time.time()
try:
    sc.upload_Triangles()
    print("pt_upload_triangles: %.1f ms" % ((time.time() - t) * 1e3))
except api.PtError as e:
    print("pt_upload_triangles: %.1f ms up to '%s'" % ((time.time() - t) * 1e3, str(e)[:80]))
