import sys, time
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes
spec = scenes.displaced_grid_mesh(1000000)
for ploc, cluster in ((16, 64), (16, 0), (32, 64)):
    sc = api.Scene(1920, 1080)
    sc.set_option("bvh_policy", 4); sc.set_option("lbvh_ploc", ploc); sc.set_option("lbvh_cluster", cluster)
    sc.load(spec)
    print("=== ploc", ploc, "cluster", cluster, flush=True)
    t = time.time(); sc.upload_Triangles(); print("upload_Triangles %.1f ms" % ((time.time() - t) * 1e3), flush=True)
