"""How much of a one-sample launch is its ragged end, and what handing the dear tiles out first would recover: per-tile costs of
one-sample launches from the counting instance, then greedy list scheduling of those items onto the resident waves in the
kernel's order (tile index) and in longest-first order.  (Item durations are taken as fixed; contention is not modelled.)"""
import heapq
import sys

import numpy as np

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

W, H = 1920, 1080
sc = api.Scene(W, H).load(scenes.cornell_box())
sc.iterations = 8
sc.set_option("count_work", 1)
sc.set_option("schedule", 0)
acc = None
for k in range(8):
    sc.render(1)
    sc.sync()
    c = sc.debug_tile_cost().astype(np.float64)
    acc = c if acc is None else acc + c
cost = acc / 8 * 64.0 / 2.4e9 * 1e6          # us per tile-sample at 2.4 GHz
print("tiles %d: cost per tile-sample mean %.1f us  min %.1f  median %.1f  p99 %.1f  max %.1f" % (cost.size, cost.mean(), cost.min(), np.median(cost), np.percentile(cost, 99), cost.max()))


def makespan(items, waves):
    heap = [0.0] * waves
    for d in items:
        t = heapq.heappop(heap)
        heapq.heappush(heap, t + d)
    return max(heap)


for waves in (4096, 6144):
    load = cost.sum() / waves
    a = makespan(cost, waves)
    b = makespan(np.sort(cost)[::-1], waves)
    print("%d waves: mean load %.1f us; kernel order: makespan %.1f us (x%.3f); dear tiles first: %.1f us (x%.3f) -> %.1f %% shorter" % (
        waves, load, a, a / load, b, b / load, 100.0 * (1 - b / a)))
