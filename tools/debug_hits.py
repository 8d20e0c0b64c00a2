import sys
sys.path.insert(0, ".")
import numpy as np
from opencl_path_tracer_amd import api, scenes
from oracle import oracle_py as O

spec = scenes.cornell_box()
osc = O.load_scene(spec)
rng = np.random.RandomState(5)
n = 20000
P = np.stack([rng.uniform(-90, 1090, n), rng.uniform(10, 990, n), rng.uniform(-990, 990, n)], 1).astype(np.float32)
D = rng.normal(size=(n, 3)); D /= np.linalg.norm(D, axis=1)[:, None]; D = D.astype(np.float32)
D[:200, 0] = 0.0; D[200:400, 1] = 0.0; D[400:600] = np.array([0, 0, 1], np.float32)
verts = spec.objects[1][0]
eye = np.array([500.0, 500.0, -1299.037842], np.float32)
targets = np.concatenate([verts[:300, 0], (verts[:300, 0] + verts[:300, 1]) * np.float32(0.5)])
P[600:1200] = eye
d = (targets - eye).astype(np.float64)
D[600:1200] = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
rays = np.zeros(n, dtype=api.RAY); rays["P"][:, :3] = P; rays["D"][:, :3] = D
sc = api.Scene(16, 16).load(spec)
t, tri = sc.debug_closest_hit(rays)
h2 = osc.closest_hit(rays.view(O.RAY), mode=2)
ot2 = np.where(h2["t"] > 0, h2["t"], np.float32(-1))
bad = np.nonzero(t.view(np.uint32) != ot2.view(np.uint32))[0]
print("mismatches:", bad.size, bad[:40])
np.save("gpurun_out/bad_rays.npy", rays[bad])
for i in bad[:12]:
    print(i, "P", rays["P"][i, :3], "D", repr(rays["D"][i, :3]), "gpu t", t[i], "tri", tri[i], "| oracle t", ot2[i], "N", h2["N"][i, :3], "mati", h2["mati"][i])
