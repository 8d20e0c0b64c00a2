"""suspend_lanes sweep (tail suspension of the megakernel)."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from sweep import run  # noqa: E402
from opencl_path_tracer_amd import scenes  # noqa: E402

what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cb", "mesh100k", "mesh1m"]
if "cb" in what:
    cb = scenes.cornell_box()
    run(1920, 1080, 8, 64, cb, reps=3, schedule=0)
    for k in (0, 16, 24, 32):
        run(1920, 1080, 8, 64, cb, reps=3, schedule=1, suspend_lanes=k)
    run(1920, 1080, 8, 16, cb, reps=1, count=True, schedule=0)
    run(1920, 1080, 8, 16, cb, reps=1, count=True, schedule=1, suspend_lanes=0)
    run(1920, 1080, 8, 16, cb, reps=1, count=True, schedule=1, suspend_lanes=24)
    run(1920, 1080, 8, 64, cb, reps=2, schedule=0, lds_scene=0)
    run(1920, 1080, 8, 64, cb, reps=2, schedule=1, suspend_lanes=16, lds_scene=0)
    run(1920, 1080, 8, 64, cb, reps=2, schedule=1, suspend_lanes=32, lds_scene=0)
for name, n, b, spp in (("mesh100k", 100000, 8, 16), ("mesh1m", 1000000, 16, 8)):
    if name in what:
        m = scenes.displaced_grid_mesh(n)
        run(1920, 1080, b, spp, m, reps=2, schedule=0)
        for k in (16, 24, 32, 40, 48, 56):
            run(1920, 1080, b, spp, m, reps=2, schedule=1, suspend_lanes=k)
        run(1920, 1080, b, spp, m, reps=1, count=True, schedule=0)
        run(1920, 1080, b, spp, m, reps=1, count=True, schedule=1, suspend_lanes=32)
        run(1920, 1080, b, spp, m, reps=2, schedule=1, suspend_lanes=32, lds_scene=0)
