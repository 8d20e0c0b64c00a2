"""Render a few launches with given options (target for rocprofv3 runs).
  scene=cornell|mesh100k|mesh1m  W= H= spp= reps= bounces= bvh_policy=  + any pt_set_option key=value"""
import sys
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
W, H = int(opts.pop("W", 1920)), int(opts.pop("H", 1080))
spp, reps, bounces = int(opts.pop("spp", 16)), int(opts.pop("reps", 3)), int(opts.pop("bounces", 8))
scene = opts.pop("scene", "cornell")
spec = {"cornell": scenes.cornell_box, "mesh100k": lambda: scenes.displaced_grid_mesh(100000),
        "mesh1m": lambda: scenes.displaced_grid_mesh(1000000)}[scene]()
sc = api.Scene(W, H)
pre = ("bvh_policy", "treelet", "lds_scene", "flat_list", "wide_nodes", "wide_lds_entries", "sah_visit_cost")          # options the upload depends on
for k in pre:
    if k in opts:
        sc.set_option(k, int(opts[k]))
sc.load(spec)
for k, v in opts.items():
    if k not in pre:
        sc.set_option(k, int(v))
sc.set_option("timing", 1)
sc.iterations = bounces
sc.render(spp)
sc.sync()
sc.set_option("reset_stats", 1)
for _ in range(reps):
    sc.render(spp)
sc.sync()
segs, samples, kms = sc.stat("segments"), sc.stat("samples"), sc.stat("kernel_ms")
print("scene %s %dx%d b%d spp%d x%d: segments %d samples %d kernel_ms %.3f -> %.1f Msamples/s, %.1f Msegments/s, nodes %d, lds %d" % (
    scene, W, H, bounces, spp, reps, segs, samples, kms, samples / kms / 1e3, segs / kms / 1e3, sc.stat("bvh_nodes"), sc.stat("lds_bytes")))
