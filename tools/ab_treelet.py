import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from sweep import run
from opencl_path_tracer_amd import scenes
for n, b, spp in ((100000, 8, 64), (1000000, 16, 32)):
    m = scenes.displaced_grid_mesh(n)
    for rep in range(2):
        run(1920, 1080, b, spp, m, reps=1)
        run(1920, 1080, b, spp, m, reps=1, lds_scene=0)
    run(1920, 1080, b, spp, m, reps=1, treelet=512)
