"""Two device SAH builds (bvh_policy 5) of the 1M-triangle mesh, for `rocprofv3 --kernel-trace --stats -- python3 tools/sahdev_profile.py`."""
import sys
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
grain = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spec = scenes.displaced_grid_mesh(n)
sc = api.Scene(32, 32)
sc.set_option("bvh_policy", 5)
sc.set_option("sah_grain", grain)
sc.load(spec)
for _ in range(3):
    sc.upload_Triangles()
    print("upload_Triangles %.2f ms (on device %d)" % (sc.stat("bvh_build_ms"), sc.stat("bvh_on_device")), flush=True)
