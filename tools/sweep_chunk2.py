"""Pass length of chained launches for the mesh scenes at 6 waves per SIMD (development tool)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from opencl_path_tracer_amd import scenes  # noqa: E402
from sweep import run  # noqa: E402

for name, n, b, spp in (("mesh100k", 100000, 8, 64), ("mesh1m", 1000000, 16, 32)):
    m = scenes.displaced_grid_mesh(n)
    for c in (-1, 8, 16, 32, 64):
        run(1920, 1080, b, spp, m, reps=1, chunk_spp=c)
    run(1920, 1080, b, spp, m, reps=1, chunk_spp=32, suspend_lanes=32)
