"""What would a different wave schedule execute?  CPU model of the megakernel's closest_hit phase.

Rays: the oracle (CPU) traces the Cornell box at 960x540 for b = 0..7 iterations; a wave = one 8x8 pixel
tile whose lanes sit at different bounce depths (a lane whose path ended restarts at bounce 0), as in the
persistent megakernel.  tools/sim/trav_trace.c walks the product's own BVH for every ray and records the
while-while phase structure (node visits, triangle tests per round).  The schedules below are then
replayed on those traces, counting wave-level executions of the node body and of the triangle body
(the quantity the VALU-bound kernel's time follows: profiles/r01/o_final_sq_counters.json)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
W, H, B = 960, 540, 8
MAXR = 48


def get_rays(spec):
    cache = "/tmp/wave_sim_rays_%s_%dx%d.npz" % (spec.name, W, H)
    if os.path.exists(cache):
        z = np.load(cache)
        return z["rays"], z["alive"]
    osc = O.load_scene(spec)
    cam = O.make_camera(spec.fov, spec.yaw, spec.pitch, spec.shift, W, H)
    rays = np.zeros((B, W * H, 8), np.float32)
    for b in range(B):
        fr = O.OracleFrame(W, H)
        fr.generate_rays(cam, nthreads=8)
        if b:
            fr.trace_rays(osc, cam, b, 0, nthreads=8)
        r = fr.rays()
        rays[b, :, 0:3] = r["P"][:, :3]
        rays[b, :, 4:7] = r["D"][:, :3]
        print("bounce", b, "done", flush=True)
    alive = np.ones((B, W * H), bool)
    for b in range(1, B):
        alive[b] = alive[b - 1] & (rays[b] != rays[b - 1]).any(1)
    np.savez(cache, rays=rays, alive=alive)
    return rays, alive


def traces(spec, rays, defer=0, bvh_spec=None):
    """bvh_spec: trace the rays against the BVH of ANOTHER scene (e.g. the spheres without the walls)."""
    spec = bvh_spec or spec
    sc = api.Scene(16, 16, device=None)
    sc.set_option("treelet", 0)
    sc.set_option("flat_list", 0)        # the traces describe ONE tree over everything it is given
    sc.load(spec)
    nodes, tris, meta, orig = sc.debug_bvh()
    L = C.CDLL(os.path.join(HERE, "libtravtrace.so"))
    n = rays.shape[0]
    ph = np.zeros((n, MAXR, 3), np.uint16)
    nr = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32)
    rays = np.ascontiguousarray(rays)
    L.trav_trace(nodes.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p),
                 C.c_int64(n), C.c_int(MAXR), C.c_int(defer), ph.ctypes.data_as(C.c_void_p), nr.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
    return ph, nr, t


def wave_rays(rays, alive, rng, k=1):
    """k rays per lane for every 8x8 tile: (tiles, 64, k) indices into the (bounce, pixel) ray table.
    Lane l of tile t owns pixel(s) of the tile; its bounce depth is uniform over the path's length."""
    ty, tx = H // 8, W // 8
    depth = alive.sum(0)                                   # segments the path executes (>= 1)
    out = []
    for t in range(ty * tx * 0 + (ty // k) * tx):          # k vertically adjacent tiles share a wave
        y0, x0 = (t // tx) * 8 * k, (t % tx) * 8
        sel = []
        for j in range(k):
            ys, xs = np.meshgrid(np.arange(y0 + 8 * j, y0 + 8 * j + 8), np.arange(x0, x0 + 8), indexing="ij")
            pix = (ys * W + xs).reshape(-1)
            b = rng.randint(0, 1 << 30, 64) % depth[pix]
            sel.append(b * (W * H) + pix)
        out.append(np.stack(sel, 1))
    return np.asarray(out)


def replay(ph, nr, idx):
    """idx: (waves, 64, k) ray ids.  Every lane runs its k rays back to back, one while-while round of its
    current ray per wave round (a lane that finishes a ray starts its next one in the next round).
    Returns wave-level executions of (node body, tri body) and the per-lane totals."""
    waves, lanes, k = idx.shape
    wn = wt = 0
    for w in range(waves):
        # per lane: concatenated phase list of its k rays
        seqs = [np.concatenate([ph[i, :nr[i]] for i in idx[w, l]]) for l in range(lanes)]
        R = max(len(s) for s in seqs)
        M = np.zeros((lanes, R, 3), np.int64)
        for l, s in enumerate(seqs):
            M[l, :len(s)] = s
        wn += M[:, :, 0].max(0).sum()
        wt += M[:, :, 1].max(0).sum()
    return wn, wt


if __name__ == "__main__":
    spec = scenes.cornell_box()
    rays, alive = get_rays(spec)
    flat = rays.reshape(-1, 8)
    ph, nr, t = traces(spec, flat, defer=0)
    live = alive.reshape(-1)
    print("rays %d, alive %d; per ray: node visits %.2f, tri tests %.2f, rounds %.2f (alive rays only)" % (
        flat.shape[0], live.sum(), ph[live, :, 0].sum() / live.sum(), ph[live, :, 1].sum() / live.sum(), nr[live].mean()))
    rng = np.random.RandomState(1)
    NODE, TRI, SHADE = 43.0, 55.0, 900.0            # VALU instructions per wave-level execution
    for k in (1, 2, 3, 4, 8):
        idx = wave_rays(rays, alive, rng, k)
        idx = idx[rng.choice(idx.shape[0], min(1500, idx.shape[0]), replace=False)]
        wn, wt = replay(ph, nr, idx)
        nseg = idx.shape[0] * k                       # wave-segments (64 rays each)
        lane_n = ph[idx.reshape(-1), :, 0].sum() / (idx.size)
        lane_t = ph[idx.reshape(-1), :, 1].sum() / (idx.size)
        valu = (wn * NODE + wt * TRI) / nseg + SHADE
        print("k=%d rays per lane: per 64 rays node body x%.1f (lane mean %.2f, util %.0f%%), tri body x%.1f (lane mean %.2f, util %.0f%%) -> ~%.0f VALU per 64 rays" % (
            k, wn / nseg, lane_n, 100 * lane_n * nseg / wn, wt / nseg, lane_t, 100 * lane_t * nseg / wt, valu), flush=True)
