"""Re-tune after the flat list: suspension threshold, schedule, SAH visit price."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import sweep  # noqa: E402
from sweep import run  # noqa: E402
from opencl_path_tracer_amd import scenes  # noqa: E402

cb = scenes.cornell_box()
run(1920, 1080, 8, 64, cb, reps=3, schedule=0)
for k in (4, 8, 16, 24, 32):
    run(1920, 1080, 8, 64, cb, reps=3, schedule=1, suspend_lanes=k)
for m100, b, spp in ((scenes.displaced_grid_mesh(100000), 8, 16), (scenes.displaced_grid_mesh(1000000), 16, 8)):
    run(1920, 1080, b, spp, m100, reps=2, schedule=0)
    for k in (24, 32, 48, 56):
        run(1920, 1080, b, spp, m100, reps=2, schedule=1, suspend_lanes=k)
