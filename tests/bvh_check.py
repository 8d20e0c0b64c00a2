"""Test helper: walk the product's packed BVH (pt_debug_bvh_copy) in numpy.  Checker code only."""
import numpy as np


def tri_test(tri12, P, D):
    """prog.cl:94-112 in float64 (tolerant checker, not the bit-defined spec)."""
    r1, r2, r3, N = tri12[0:3], tri12[3:6], tri12[6:9], tri12[9:12]
    den = np.dot(D, N)
    if den == 0 or not np.isfinite(den):
        return -1.0
    t = np.dot(r1 - P, N) / den
    if not (t > 0):
        return -1.0
    p = P + D * t
    eps = -1e-6 * max(1.0, np.abs(p).max())
    if np.dot(np.cross(r2 - r1, p - r1), N) < eps:
        return -1.0
    if np.dot(np.cross(r3 - r2, p - r2), N) < eps:
        return -1.0
    if np.dot(np.cross(r1 - r3, p - r3), N) < eps:
        return -1.0
    return t


def leaves_reaching(nodes, P, D):
    """All leaf (first,count) ranges whose box chain the ray passes (plain slab test, no pruning)."""
    out = []
    stack = [0]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / D
    while stack:
        n = stack.pop()
        q = nodes[n]
        refs = q[12:14].view(np.int32)
        for side in (0, 1):
            lo = np.array([q[0 + 2 * side], q[4 + 2 * side], q[8 + 2 * side]], dtype=np.float64)
            hi = np.array([q[1 + 2 * side], q[5 + 2 * side], q[9 + 2 * side]], dtype=np.float64)
            if not np.all(hi >= lo):
                continue
            with np.errstate(invalid="ignore"):
                t0 = (lo - P) * inv
                t1 = (hi - P) * inv
            tn = np.nanmax(np.minimum(t0, t1))
            tf = np.nanmin(np.maximum(t0, t1))
            if tf >= tn and tf >= 0:
                ref = int(refs[side])
                if ref >= 0:
                    stack.append(ref)
                else:
                    v = ~ref
                    out.append((v >> 3, (v & 7) + 1))
    return out


def validate_structure(nodes, tris, ntris, n_flat=0):
    """Every packed triangle behind the flat list [0, n_flat) is in exactly one leaf and inside that leaf's
    box; the flat list is in no leaf."""
    seen = np.zeros(ntris, dtype=int)
    seen[:n_flat] = 1
    depth_max = 0
    stack = [(0, 0)]
    while stack:
        n, d = stack.pop()
        depth_max = max(depth_max, d)
        q = nodes[n]
        refs = q[12:14].view(np.int32)
        for side in (0, 1):
            lo = np.array([q[0 + 2 * side], q[4 + 2 * side], q[8 + 2 * side]])
            hi = np.array([q[1 + 2 * side], q[5 + 2 * side], q[9 + 2 * side]])
            ref = int(refs[side])
            if ref >= 0:
                stack.append((ref, d + 1))
                continue
            if not np.all(hi >= lo):
                continue          # empty child
            v = ~ref
            first, count = v >> 3, (v & 7) + 1
            for k in range(first, first + count):
                assert k >= n_flat or ntris == n_flat, "a flat-list triangle is referenced by a leaf"
                seen[k] += k >= n_flat
                pts = tris[k, :9].reshape(3, 3)
                assert np.all(pts >= lo - 0) and np.all(pts <= hi + 0), "triangle %d outside its leaf box" % k
    assert (seen == 1).all()
    return depth_max


def decode_wide(w):
    """Planes of the 4-wide quantised nodes exactly as the kernel decodes them: fma((float)q, 2^(exp - 127), origin),
    evaluated in float64 (exact: an 8-bit q times a power of two plus a float) and rounded once to float32.
    Returns lo[n, 4, 3], hi[n, 4, 3]."""
    step = np.ldexp(1.0, w["exp"].astype(np.int64) - 127)               # [n, 3], float64
    o = w["origin"].astype(np.float64)
    q = w["q"].astype(np.float64)                                        # [n, 6, 4]: lo_x, hi_x, lo_y, hi_y, lo_z, hi_z
    lo = np.stack([(q[:, 2 * a, :] * step[:, a, None] + o[:, a, None]).astype(np.float32) for a in range(3)], -1)
    hi = np.stack([(q[:, 2 * a + 1, :] * step[:, a, None] + o[:, a, None]).astype(np.float32) for a in range(3)], -1)
    return lo, hi


def validate_wide(nodes, wide, ntris, n_flat=0):
    """The wide tree holds exactly the leaves of the BVH2, each once, and every child's decoded box contains the
    BVH2 boxes of all leaves below it (so culling with it can only visit more, never less).  Returns the most
    entries a traversal can have pending when it visits an interior node."""
    NOCHILD = -1                # the leaf {packet 0, 1 triangle} behind an inverted box
    leaf_box = {}
    left, right = nodes[:, 12].view(np.int32), nodes[:, 13].view(np.int32)
    for i in range(nodes.shape[0]):
        q = nodes[i]
        for side, ref in ((0, int(left[i])), (1, int(right[i]))):
            lo = np.array([q[0 + 2 * side], q[4 + 2 * side], q[8 + 2 * side]])
            hi = np.array([q[1 + 2 * side], q[5 + 2 * side], q[9 + 2 * side]])
            if ref < 0 and np.all(hi >= lo):
                assert ref not in leaf_box
                leaf_box[ref] = (lo, hi)
    lo, hi = decode_wide(wide)
    seen = set()
    # bottom-up union of the true leaf boxes: children have larger indices than their parent (built top-down)
    tlo = np.full((wide.shape[0], 3), np.inf)
    thi = np.full((wide.shape[0], 3), -np.inf)
    pending = np.zeros(wide.shape[0], dtype=int)
    for i in range(wide.shape[0]):
        nc = int(wide["nchild"][i])
        assert 1 <= nc <= 4 or wide.shape[0] == 1
        for k in range(4):
            ref = int(wide["ref"][i, k])
            if k >= nc:
                assert ref == NOCHILD and np.all(lo[i, k] >= hi[i, k])
                continue
            if ref >= 0:
                assert ref > i
                pending[ref] = pending[i] + nc - 1
    for i in range(wide.shape[0] - 1, -1, -1):
        for k in range(int(wide["nchild"][i])):
            ref = int(wide["ref"][i, k])
            if ref >= 0:
                clo, chi = tlo[ref], thi[ref]
            else:
                assert ref in leaf_box and ref not in seen
                seen.add(ref)
                clo, chi = leaf_box[ref]
            assert np.all(lo[i, k] <= clo) and np.all(hi[i, k] >= chi), "wide node %d child %d does not contain its subtree" % (i, k)
            tlo[i] = np.minimum(tlo[i], clo)
            thi[i] = np.maximum(thi[i], chi)
    assert seen == set(leaf_box)
    return int(pending.max())
