import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from sweep import run
from opencl_path_tracer_amd import scenes
cb = scenes.cornell_box()
run(1920, 1080, 8, 64, cb, reps=3)
run(1920, 1080, 8, 64, cb, reps=3, schedule=0)
run(1920, 1080, 8, 16, cb, reps=1, count=True)
run(1920, 1080, 8, 16, cb, reps=2, variant=1)
m = scenes.displaced_grid_mesh(100000)
run(1920, 1080, 8, 64, m, reps=1)
m = scenes.displaced_grid_mesh(1000000)
run(1920, 1080, 16, 32, m, reps=1)
