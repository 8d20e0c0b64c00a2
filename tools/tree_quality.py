"""SAH cost and leaf statistics of the BVH2 the host builder and the device builders produce for the same scene
(area-weighted expected node visits / triangle tests of a random ray; read from pt_debug_bvh_copy)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from opencl_path_tracer_amd import api, scenes


def quality(nodes):
    q = nodes[:, :12].reshape(-1, 3, 4).astype(np.float64)
    refs = nodes[:, 12:14].copy().view(np.int32)
    lo = np.stack([q[:, :, 0], q[:, :, 2]], axis=1)     # [node][child][axis]
    hi = np.stack([q[:, :, 1], q[:, :, 3]], axis=1)
    d = np.maximum(hi - lo, 0.0)
    live = np.all(hi >= lo, axis=2)
    area = (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]) * live
    rlo = np.min(np.where(live[0][:, None], lo[0], np.inf), axis=0)
    rhi = np.max(np.where(live[0][:, None], hi[0], -np.inf), axis=0)
    rd = rhi - rlo
    root = rd[0] * rd[1] + rd[1] * rd[2] + rd[2] * rd[0]
    leaf = (refs < 0) & live
    inner = (refs >= 0) & live
    count = ((~refs) & 7) + 1
    # overlap of the two child boxes relative to the smaller of them
    olo = np.maximum(lo[:, 0], lo[:, 1])
    ohi = np.minimum(hi[:, 0], hi[:, 1])
    od = np.maximum(ohi - olo, 0.0)
    oarea = od[:, 0] * od[:, 1] + od[:, 1] * od[:, 2] + od[:, 2] * od[:, 0]
    both = live[:, 0] & live[:, 1]
    return dict(
        inner_cost=1.0 + float(np.sum(area * inner) / root),
        leaf_visits=float(np.sum(area * leaf) / root),
        tri_cost=float(np.sum(area * leaf * count) / root),
        leaves=int(leaf.sum()),
        tris_per_leaf=float((count * leaf).sum() / max(1, leaf.sum())),
        hist=np.bincount(count[leaf], minlength=9)[1:].tolist(),
        overlap=float(np.sum(oarea * both) / root),
    )


host_only = "--host" in sys.argv          # no GPU: the host builder only
which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["mesh100k", "mesh1M"]
cases = ((0, 0, 0), (4, 0, 0), (4, 16, 0), (4, 32, 0), (4, 16, 64), (4, 16, 8))
if host_only:
    cases = cases[:1]
for name, spec in (("cornell", scenes.cornell_box()), ("mesh100k", scenes.displaced_grid_mesh(100000)), ("mesh1M", scenes.displaced_grid_mesh(1000000))):
    if name not in which:
        continue
    for policy, ploc, cluster in cases:
        sc = api.Scene(64, 64, device=None if host_only else 0)
        sc.set_option("bvh_policy", policy)
        sc.set_option("lbvh_ploc", ploc)
        sc.set_option("lbvh_cluster", cluster)
        sc.load(spec)
        nodes, tris, meta, orig = sc.debug_bvh()
        r = quality(nodes)
        print("%-9s policy %d ploc %2d cluster %2d: node visits %7.2f  leaf visits %6.2f  triangle tests %7.2f  overlap %7.2f  leaves %7d  tris/leaf %.2f  hist %s" % (
            name, policy, ploc, cluster, r["inner_cost"], r["leaf_visits"], r["tri_cost"], r["overlap"], r["leaves"], r["tris_per_leaf"], r["hist"]), flush=True)
