import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from sweep import run  # noqa: E402
from opencl_path_tracer_amd import scenes  # noqa: E402
cb = scenes.cornell_box()
for spp in (64, 256):
    for ch in (8, 16, 32, 64, 0):
        run(1920, 1080, 8, spp, cb, reps=2 if spp > 64 else 3, chunk_spp=ch)
for ch, k in ((32, 8), (32, 24), (64, 16), (64, 8)):
    run(1920, 1080, 8, 128, cb, reps=2, chunk_spp=ch, suspend_lanes=k)
run(1920, 1080, 8, 128, cb, reps=2, chunk_spp=32, schedule=0)
m = scenes.displaced_grid_mesh(100000)
for ch in (8, 16, 32, 0):
    run(1920, 1080, 8, 64, m, reps=1, chunk_spp=ch)
m = scenes.displaced_grid_mesh(1000000)
for ch in (8, 16, 0):
    run(1920, 1080, 16, 32, m, reps=1, chunk_spp=ch)
