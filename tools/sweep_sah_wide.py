"""SAH visit price and leaf policy once more, now that a node visit is a 4-wide one (development tool)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from opencl_path_tracer_amd import scenes
from sweep import run
for name, n, b, spp in (("mesh100k", 100000, 8, 16), ("mesh1m", 1000000, 16, 8)):
    m = scenes.displaced_grid_mesh(n)
    for cost in (5, 10, 15, 20, 30):
        run(1920, 1080, b, spp, m, reps=2, sah_visit_cost=cost)
    for pol in (2, 3):
        run(1920, 1080, b, spp, m, reps=2, bvh_policy=pol)
