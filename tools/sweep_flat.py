"""flat_list (big-triangle list in front of the tree) on / off."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from sweep import run  # noqa: E402
from opencl_path_tracer_amd import scenes  # noqa: E402

cb = scenes.cornell_box()
for fl in (16, 0):
    run(1920, 1080, 8, 64, cb, reps=3, flat_list=fl)
    run(1920, 1080, 8, 64, cb, reps=3, flat_list=fl, schedule=0)
    run(1920, 1080, 8, 16, cb, reps=1, count=True, flat_list=fl)
    run(1920, 1080, 8, 16, cb, reps=2, flat_list=fl, variant=1)
for name, n, b, spp in (("mesh100k", 100000, 8, 16), ("mesh1m", 1000000, 16, 8)):
    m = scenes.displaced_grid_mesh(n)
    for fl in (16, 0):
        run(1920, 1080, b, spp, m, reps=2, flat_list=fl)
    run(1920, 1080, b, spp, m, reps=1, count=True)
