"""Two restructurings the round-1 review asked to be measured, replayed on the recorded rays (lockstep waves
of the Cornell box, the production schedule's statistics):
 (a) leaf phase with 4 lanes per ray: a leaf's <= 4 packets tested by 4 lanes at once (16 rays per
     execution) instead of one lane looping over them;
 (b) two-level scheme: the 12 wall / lamp triangles tested as a flat wave-uniform list, the BVH (spheres
     only) entered on a hit of its root box."""
import copy
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tools/sim")
import wave_sim as ws  # noqa: E402
from opencl_path_tracer_amd import scenes  # noqa: E402

NODE, TRI = 43.0, 55.0
spec = scenes.cornell_box()
rays, alive = ws.get_rays(spec)
W, H, B = ws.W, ws.H, ws.B
flat = rays.reshape(-1, 8)
ph, nr, _ = ws.traces(spec, flat, 0)
sph = copy.deepcopy(spec)
sph.objects = spec.objects[1:]                       # the two spheres only
ph2, nr2, _ = ws.traces(spec, flat, 0, bvh_spec=sph)
ph = ph.reshape(B, W * H, ws.MAXR, 3)
ph2 = ph2.reshape(B, W * H, ws.MAXR, 3)
ty, tx = H // 8, W // 8
rng = np.random.RandomState(3)
tiles = rng.choice(ty * tx, 1200, replace=False)
tot = dict(n=0, t=0, coop=0, coop_pass=0, n2=0, t2=0, trips=0)
for b in range(B):
    for t_ in tiles:
        y0, x0 = (t_ // tx) * 8, (t_ % tx) * 8
        ys, xs = np.meshgrid(np.arange(y0, y0 + 8), np.arange(x0, x0 + 8), indexing="ij")
        pix = (ys * W + xs).reshape(-1)
        a = alive[b, pix]
        if not a.any():
            continue
        M = ph[b, pix][a].astype(np.int64)
        tot["trips"] += 1
        tot["n"] += M[:, :, 0].max(0).sum()
        tot["t"] += M[:, :, 1].max(0).sum()
        # (a): per round, every lane holding leaves needs 4 lanes per leaf; the wave runs max(leaves per lane) steps
        #      when <= 16 lanes hold leaves, proportionally more otherwise
        leaves = M[:, :, 2]
        holders = (leaves > 0).sum(0)
        tot["coop"] += (leaves.max(0) * np.ceil(np.maximum(holders, 1) / 16.0)).sum()
        M2 = ph2[b, pix][a].astype(np.int64)
        tot["n2"] += M2[:, :, 0].max(0).sum()
        tot["t2"] += M2[:, :, 1].max(0).sum()
n = tot["trips"]
print("per wave-trip (lockstep waves, %d trips): node body x%.1f, tri body x%.1f -> %.0f VALU" % (n, tot["n"] / n, tot["t"] / n, (tot["n"] * NODE + tot["t"] * TRI) / n))
COOP = TRI + 30.0        # ray broadcast (7 ds_bpermute), 2-step min reduction with the rank tie-break, write-back
print("(a) 4 lanes per ray in the leaf phase: cooperative tri body x%.1f at ~%.0f VALU each -> %.0f VALU (x%.3f)" % (
    tot["coop"] / n, COOP, (tot["n"] * NODE + tot["coop"] * COOP) / n, (tot["n"] * NODE + tot["t"] * TRI) / (tot["n"] * NODE + tot["coop"] * COOP)))
FLAT = 12 * 30.0 + 20.0  # 12 exact tests without early-outs on every lane + the root box test
print("(b) walls as a flat list + sphere BVH: node body x%.1f, tri body x%.1f, + %.0f VALU flat -> %.0f VALU (x%.3f)" % (
    tot["n2"] / n, tot["t2"] / n, FLAT, (tot["n2"] * NODE + tot["t2"] * TRI) / n + FLAT,
    (tot["n"] * NODE + tot["t"] * TRI) / ((tot["n2"] * NODE + tot["t2"] * TRI) + FLAT * n)))
