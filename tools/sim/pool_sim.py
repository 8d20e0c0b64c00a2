"""Would pooling the rays of a WORKGROUP pay?  Every lane keeps its pixel and its path state, but posts its ray to
an LDS pool; rays that miss both children of the root never enter it; the pooled rays are traced by as many waves
as they fill (64 per wave, run to completion), results go back to the owners.  Replayed on the recorded rays of
the Cornell box (spheres in the tree, walls in the flat list): wave-level executions of the node / triangle body
per 8 waves x 64 lane-segments, against waves tracing their own 64 rays."""
import copy
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tools/sim")
import wave_sim as ws  # noqa: E402
from opencl_path_tracer_amd import scenes  # noqa: E402

NODE, TRI = 35.0, 55.0
spec = scenes.cornell_box()
rays, alive = ws.get_rays(spec)
W, H, B = ws.W, ws.H, ws.B
flat = rays.reshape(-1, 8)
sph = copy.deepcopy(spec)
sph.objects = spec.objects[1:]
ph, nr, _ = ws.traces(spec, flat, 0, bvh_spec=sph)
ph = ph.reshape(B, W * H, ws.MAXR, 3).astype(np.int64)
depth = alive.sum(0)
ty, tx = H // 8, W // 8
rng = np.random.RandomState(7)


def cost(M):            # M: (lanes, rounds, 3) -> wave-level node, tri executions
    return M[:, :, 0].max(0).sum(), M[:, :, 1].max(0).sum()


tot = dict(own=np.zeros(2), pool=np.zeros(2), sorted=np.zeros(2), cls=np.zeros(2), pooled=0, waves_pool=0, groups=0)
WAVES = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for g in range(600):
    t0 = rng.randint(0, ty * (tx - WAVES))
    Ms = []
    for w in range(WAVES):
        t_ = t0 + w
        y0, x0 = (t_ // tx) * 8, (t_ % tx) * 8
        ys, xs = np.meshgrid(np.arange(y0, y0 + 8), np.arange(x0, x0 + 8), indexing="ij")
        pix = (ys * W + xs).reshape(-1)
        b = rng.randint(0, 1 << 30, 64) % depth[pix]          # lanes at mixed bounce depths (restart schedule)
        M = ph[b, pix]
        Ms.append(M)
        tot["own"] += cost(M)
    A = np.concatenate(Ms)                                     # (WAVES * 64, rounds, 3)
    nodes = A[:, :, 0].sum(1)
    inpool = nodes > 1
    P = A[inpool]
    tot["pooled"] += inpool.sum()
    tot["groups"] += 1
    # what stays with the owners: the root visit, one node-body execution per wave
    base = np.array([WAVES, 0.0])
    for name, order in (("pool", np.arange(P.shape[0])), ("sorted", np.argsort(nodes[inpool])), ("cls", np.argsort(P[:, 0, 0], kind="stable"))):
        c = base.copy()
        Q = P[order]
        for i in range(0, Q.shape[0], 64):
            c += cost(Q[i:i + 64])
        tot[name] += c
    tot["waves_pool"] += (P.shape[0] + 63) // 64
n = tot["groups"]
print("workgroups of %d waves, %d groups; %.0f %% of the rays enter the pool, %.2f of %d waves trace" % (WAVES, n, 100.0 * tot["pooled"] / (n * WAVES * 64), tot["waves_pool"] / n, WAVES))
ref = tot["own"][0] * NODE + tot["own"][1] * TRI
for name, label in (("own", "every wave traces its own 64 rays (no suspension)"), ("pool", "pooled, in lane order"), ("cls", "pooled, ordered by the length of the first node phase"), ("sorted", "pooled, ordered by total node visits (bound)")):
    v = tot[name]
    print("%-58s node body x%.1f tri body x%.1f per wave  -> %.0f VALU (x%.2f)" % (label, v[0] / n / WAVES, v[1] / n / WAVES, (v[0] * NODE + v[1] * TRI) / n / WAVES, ref / (v[0] * NODE + v[1] * TRI)))
