import sys
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes
spec = scenes.cornell_box()
for spp in (4, 16, 64):
    sc = api.Scene(1920, 1080).load(spec)
    sc.set_option("count_work", 1)
    sc.iterations = 8
    sc.render(spp)
    segs = sc.stat("segments"); steps = sc.stat("tile_lane_steps")
    print("spp %3d: lane segments %.4g, wave segment-steps x64 %.4g -> per-tile lane balance %.1f%%" % (spp, segs, steps, 100 * segs / steps))
