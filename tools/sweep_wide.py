"""4-wide quantised nodes against the BVH2 for the scenes read from global memory (development tool)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from opencl_path_tracer_amd import scenes  # noqa: E402
from sweep import run  # noqa: E402

what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["mesh100k", "mesh1m"]
for name, n, b, spp in (("mesh100k", 100000, 8, 16), ("mesh1m", 1000000, 16, 8)):
    if name not in what:
        continue
    m = scenes.displaced_grid_mesh(n)
    for wide in (0, 1):
        for w in (4, 5, 6, 7):
            run(1920, 1080, b, spp, m, reps=2, wide_nodes=wide, waves_per_simd=w)
    run(1920, 1080, b, spp, m, reps=2, wide_nodes=1, wide_lds_entries=16)
    run(1920, 1080, b, spp, m, reps=2, wide_nodes=1, suspend_lanes=16)
    run(1920, 1080, b, spp, m, reps=2, wide_nodes=1, suspend_lanes=32)
    run(1920, 1080, b, 4 * spp, m, reps=1, wide_nodes=1)
    run(1920, 1080, b, spp, m, reps=1, wide_nodes=0, count=True)
    run(1920, 1080, b, spp, m, reps=1, wide_nodes=1, count=True)
    run(1920, 1080, b, spp, m, reps=2, variant=1, wide_nodes=0)
    run(1920, 1080, b, spp, m, reps=2, variant=1, wide_nodes=1)
