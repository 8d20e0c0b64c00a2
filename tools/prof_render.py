"""Render a few launches with given options (target for rocprofv3 runs)."""
import sys
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
W, H = int(opts.pop("W", 1920)), int(opts.pop("H", 1080))
spp, reps, bounces = int(opts.pop("spp", 16)), int(opts.pop("reps", 3)), int(opts.pop("bounces", 8))
spec = scenes.cornell_box()
sc = api.Scene(W, H)
if "bvh_policy" in opts:
    sc.set_option("bvh_policy", int(opts.pop("bvh_policy")))
sc.load(spec)
for k, v in opts.items():
    sc.set_option(k, int(v))
sc.iterations = bounces
for _ in range(reps):
    sc.render(spp)
sc.sync()
print("segments", sc.stat("segments"), "samples", sc.stat("samples"))
