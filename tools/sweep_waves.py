"""Waves per SIMD x suspension threshold for the kernels that read nodes from global memory (development tool)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from opencl_path_tracer_amd import scenes  # noqa: E402
from sweep import run  # noqa: E402

cb = scenes.cornell_box()
for w in (4, 5, 6):
    run(1920, 1080, 8, 64, cb, reps=2, lds_scene=0, waves_per_simd=w)
for name, n, b, spp in (("mesh100k", 100000, 8, 16), ("mesh1m", 1000000, 16, 8)):
    m = scenes.displaced_grid_mesh(n)
    for w in (4, 5, 6, 7):
        for k in (16, 24, 32):
            run(1920, 1080, b, spp, m, reps=2, waves_per_simd=w, suspend_lanes=k)
    run(1920, 1080, b, spp, m, reps=2, schedule=0)
    run(1920, 1080, b, 4 * spp, m, reps=1)
