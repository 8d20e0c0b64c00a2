"""Unified wave-schedule simulator for the megakernel (see wave_sim.py for the traces).

A wave renders S samples of an 8x8 tile.  Every lane walks its pixel's true paths (the oracle's rays per
bounce); a "trip" = every lane that has a ray traverses (while-while rounds; the wave leaves the traversal
when at most K lanes are unfinished and at least one has finished), then the finished lanes shade.  A lane
whose path ended starts its next sample only when at least M lanes are waiting to start (or nobody is doing
anything else): M = 1 is the restart scheme, M = 64 is lockstep per sample.  Counts wave-level executions."""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tools/sim")
import wave_sim as ws  # noqa: E402
from opencl_path_tracer_amd import scenes  # noqa: E402

NODE, TRI, SHADE, ROUND, GEN = 43.0, 55.0, 900.0, 12.0, 150.0


def load():
    spec = scenes.cornell_box()
    rays, alive = ws.get_rays(spec)
    W, H, B = ws.W, ws.H, ws.B
    ph, nr, t = ws.traces(spec, rays.reshape(-1, 8), 0)
    return ph.reshape(B, W * H, ws.MAXR, 3)[..., :2], nr.reshape(B, W * H), alive


def sim(ph, nr, alive, K, M, S=8, ntiles=150, seed=5):
    W, H = ws.W, ws.H
    ty, tx = H // 8, W // 8
    rng = np.random.RandomState(seed)
    tiles = rng.choice(ty * tx, ntiles, replace=False)
    tn = tt = rounds = trips = shades = gens = 0
    for t_ in tiles:
        y0, x0 = (t_ // tx) * 8, (t_ % tx) * 8
        ys, xs = np.meshgrid(np.arange(y0, y0 + 8), np.arange(x0, x0 + 8), indexing="ij")
        pix = (ys * W + xs).reshape(-1)
        plen = alive[:, pix].sum(0)
        s = np.zeros(64, int)          # samples started
        b = np.full(64, -1)            # bounce in flight (-1: waiting to start a sample)
        src = np.zeros(64, int)        # which pixel's path this lane currently follows
        cur = [None] * 64
        while True:
            waiting = [l for l in range(64) if b[l] < 0 and s[l] < S]
            busy = [l for l in range(64) if b[l] >= 0]
            if not waiting and not busy:
                break
            if waiting and (len(waiting) >= M or not busy):
                gens += 1
                for l in waiting:
                    src[l] = pix[(l + 17 * s[l]) % 64]      # another pixel's path for later samples: keeps the tile's mix
                    b[l] = 0
                    s[l] += 1
                busy = [l for l in range(64) if b[l] >= 0]
            for l in busy:
                if cur[l] is None:
                    cur[l] = [(int(x[0]), int(x[1])) for x in ph[b[l], src[l], :nr[b[l], src[l]]]]
            trips += 1
            while True:
                act = [l for l in busy if cur[l]]
                if not act or (len(act) <= K and len(act) < len(busy)):
                    break
                tn += max(cur[l][0][0] for l in act)
                tt += max(cur[l][0][1] for l in act)
                rounds += 1
                for l in act:
                    cur[l].pop(0)
            fin = [l for l in busy if not cur[l]]
            if fin:
                shades += 1
            for l in fin:
                cur[l] = None
                b[l] += 1
                if b[l] >= alive[:, src[l]].sum():
                    b[l] = -1
    n = ntiles * S
    valu = (tn * NODE + tt * TRI + rounds * ROUND + shades * SHADE + gens * GEN) / n
    return tn / n, tt / n, trips / n, shades / n, gens / n, valu


if __name__ == "__main__":
    ph, nr, alive = load()
    base = None
    for K, M in ((0, 64), (0, 1), (8, 1), (16, 1), (24, 1), (32, 1), (0, 32), (0, 48), (8, 48), (16, 48), (8, 32), (16, 32), (24, 32), (8, 64), (16, 16), (24, 16)):
        n_, t_, tr, sh, g, v = sim(ph, nr, alive, K, M)
        if base is None:
            base = v
        print("K=%2d M=%2d: per tile-sample node x%.0f tri x%.0f trips %.2f shades %.2f starts %.2f -> %.0f VALU (x%.3f vs lockstep)" % (K, M, n_, t_, tr, sh, g, v, base / v), flush=True)
