"""Re-tune: suspension threshold, schedule, pass length."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from sweep import run  # noqa: E402
from opencl_path_tracer_amd import scenes  # noqa: E402

cb = scenes.cornell_box()
for k in (8, 12, 16, 24, 32):
    run(1920, 1080, 8, 64, cb, reps=3, schedule=1, suspend_lanes=k)
for ch in (16, 32, 64):
    run(1920, 1080, 8, 64, cb, reps=3, chunk_spp=ch)
    run(1920, 1080, 8, 256, cb, reps=1, chunk_spp=ch)
m = scenes.displaced_grid_mesh(100000)
for k in (16, 24, 32):
    run(1920, 1080, 8, 64, m, reps=1, suspend_lanes=k)
m = scenes.displaced_grid_mesh(1000000)
for k in (16, 24, 32):
    run(1920, 1080, 16, 32, m, reps=1, suspend_lanes=k)
