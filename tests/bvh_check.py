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
