"""Rate of the reference's own frame loop (generate_rays + trace_rays per sample, main.cpp:683-686) through the drop-in
API, next to the fused render(n) (development tool)."""
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

W, H, B = 1920, 1080, 8
for name, spec in (("cornell", scenes.cornell_box()), ("mesh100k", scenes.displaced_grid_mesh(100000))):
  for persistent in (0, 1):
    sc = api.Scene(W, H).load(spec)
    sc.set_option("persistent", persistent)
    sc.iterations = B
    print("persistent", persistent)
    for n in (4, 32):
        sc.render(2, fused=False)
        sc.sync()
        t = time.time()
        sc.render(n, fused=False)
        sc.sync()
        dt = time.time() - t
        print("%s: %d x (generate_rays + trace_rays): %.1f Msamples/s (%.2f ms per sample)" % (name, n, W * H * n / dt / 1e6, dt / n * 1e3), flush=True)
    sc.render(32); sc.sync()
    t = time.time(); sc.render(32); sc.sync(); dt = time.time() - t
    print("%s: render(32): %.1f Msamples/s" % (name, W * H * 32 / dt / 1e6), flush=True)
