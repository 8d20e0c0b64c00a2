"""Would a 4-wide BVH halve the dependent fetches of the mesh scenes?  Collapses the product's BVH2 into a BVH4
(split the child with the largest box until a node has 4 children) and counts, for the oracle's rays, wide-node
visits / leaf visits / triangle tests / stack depth against the BVH2 numbers of wave_sim.traces."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tools/sim")
import wave_sim as ws  # noqa: E402
from opencl_path_tracer_amd import api, scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NODE4 = np.dtype([("lo", "<f4", (4, 3)), ("hi", "<f4", (4, 3)), ("ref", "<i4", 4), ("n", "<i4"), ("pad", "<i4", 3)])


def collapse(nodes):
    left, right = nodes[:, 12].view(np.int32), nodes[:, 13].view(np.int32)

    def child_boxes(i):
        q = nodes[i]
        return [(np.array([q[0], q[4], q[8]]), np.array([q[1], q[5], q[9]]), int(left[i])), (np.array([q[2], q[6], q[10]]), np.array([q[3], q[7], q[11]]), int(right[i]))]

    def area(lo, hi):
        d = np.maximum(hi - lo, 0)
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]

    out = []
    index = {}
    work = [0]
    index[0] = 0
    out.append(None)
    while work:
        i = work.pop()
        kids = child_boxes(i)
        while len(kids) < 4:
            cand = [(area(lo, hi), k) for k, (lo, hi, r) in enumerate(kids) if r >= 0]
            if not cand:
                break
            _, k = max(cand)
            lo, hi, r = kids.pop(k)
            kids.extend(child_boxes(r))
        rec = np.zeros((), NODE4)
        rec["n"] = len(kids)
        for c, (lo, hi, r) in enumerate(kids):
            rec["lo"][c], rec["hi"][c] = lo, hi
            if r >= 0:
                if r not in index:
                    index[r] = len(out)
                    out.append(None)
                    work.append(r)
                rec["ref"][c] = index[r]
            else:
                rec["ref"][c] = r
        out[index[i]] = rec
    return np.array(out, dtype=NODE4)


if __name__ == "__main__":
    ntris = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    ws.W, ws.H = 480, 270
    spec = scenes.displaced_grid_mesh(ntris)
    rays, alive = ws.get_rays(spec)
    flat = rays.reshape(-1, 8)[alive.reshape(-1)]
    flat = np.ascontiguousarray(flat[np.random.RandomState(1).choice(flat.shape[0], min(200000, flat.shape[0]), replace=False)])
    sc = api.Scene(16, 16, device=None)
    sc.set_option("treelet", 0)
    sc.load(spec)
    nodes, tris, meta, orig = sc.debug_bvh()
    n_flat = int(sc.stat("flat_triangles"))
    # BVH2 counts with the same emulator family (flat list handled by starting best_t... here simply: trace the tree as is)
    L2 = C.CDLL(os.path.join(HERE, "libtravtrace.so"))
    ph = np.zeros((flat.shape[0], ws.MAXR, 3), np.uint16)
    nr = np.zeros(flat.shape[0], np.int32)
    t = np.zeros(flat.shape[0], np.float32)
    L2.trav_trace(nodes.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p), flat.ctypes.data_as(C.c_void_p), C.c_int64(flat.shape[0]), C.c_int(ws.MAXR), C.c_int(1),
                  ph.ctypes.data_as(C.c_void_p), nr.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
    n4 = collapse(nodes)
    L4 = C.CDLL(os.path.join(HERE, "libbvh4trace.so"))
    out = np.zeros((flat.shape[0], 4), np.int32)
    L4.bvh4_trace(n4.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p), C.c_int(n_flat), flat.ctypes.data_as(C.c_void_p), C.c_int64(flat.shape[0]), out.ctypes.data_as(C.c_void_p))
    print("%d triangles, BVH2 %d nodes -> BVH4 %d nodes (%.1f MB at 64 B quantized, %.1f MB at 128 B)" % (tris.shape[0], nodes.shape[0], n4.shape[0], n4.shape[0] * 64 / 1e6, n4.shape[0] * 128 / 1e6))
    print("BVH2 (no flat pruning): node visits %.2f, leaves %.2f, tri tests %.2f per ray" % (ph[:, :, 0].sum() / len(flat), ph[:, :, 2].sum() / len(flat), ph[:, :, 1].sum() / len(flat)))
    print("BVH4 (flat list first): node visits %.2f, leaves %.2f, tri tests %.2f per ray; max stack %d (99.9th pct %d)" % (
        out[:, 0].mean(), out[:, 1].mean(), out[:, 2].mean(), out[:, 3].max(), np.percentile(out[:, 3], 99.9)))
