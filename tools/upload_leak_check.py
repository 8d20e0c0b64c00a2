import sys
sys.path.insert(0, ".")
import torch
from opencl_path_tracer_amd import api, scenes
spec = scenes.displaced_grid_mesh(100000)
sc = api.Scene(64, 64)
sc.load(spec)
free0 = torch.cuda.mem_get_info()[0]
for k in range(60):
    sc.upload_Triangles()
    if k % 20 == 19:
        print(k, "free MB delta", (torch.cuda.mem_get_info()[0] - free0) / 1e6, flush=True)
sc.render(1); sc.sync()
print("ok", sc.stat("bvh_on_device"))
